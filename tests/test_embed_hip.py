"""GPU tests of layer 1 as an embedding sum (csrc/dca_embed.hip, `dca_l1_embed`): relu(b1 + sum_pos W1[:, pos*depth + s[pos]]).
The kernel adds the gathered fp32 weights in ascending position order, one fp32 add each — so it is checked BIT FOR BIT against
the same sequence of float32 additions on the host (numpy), for every instantiated geometry, ragged row counts and each output
form (fp32, bf16, fp16 planes, e4m3); then against the one-hot MFMA kernel it replaces and inside the network."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GEOS = [(54, 6), (16, 16), (25, 25), (36, 36), (49, 49), (49, 6)]


def _case(D, depth, m, n_pad, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(n_pad, D * depth, generator=g) * 0.3
    b = torch.randn(n_pad, generator=g)
    x = torch.randint(0, depth, (m, D), generator=g, dtype=torch.int64).to(torch.uint8)
    return w, b, x


def _host_sum(w, b, x, depth, relu=True):
    """The kernel's arithmetic: fp32 accumulator starts at the bias, one fp32 add per position, ascending."""
    wt = w.t().contiguous().numpy()  # [K, n]
    xs = x.numpy().astype(np.int64)
    acc = np.broadcast_to(b.numpy(), (x.shape[0], b.numel())).astype(np.float32).copy()
    for pos in range(x.shape[1]):
        acc += wt[pos * depth + xs[:, pos]]  # float32 + float32 -> float32, elementwise
    return np.maximum(acc, 0.0) if relu else acc


@pytest.mark.parametrize("D,depth", GEOS)
@pytest.mark.parametrize("m", [1, 130, 1537])
def test_embed_equals_the_same_fp32_additions_on_the_host(D, depth, m):
    from deepcubea_amd import _lib
    _lib.require_gpu()
    assert _lib.l1_embed_supported(D, depth)
    n_pad = 192
    w, b, x = _case(D, depth, m, n_pad, D * 1000 + depth * 10 + m)
    wt = w.t().contiguous().cuda()
    for relu in (True, False):
        want = _host_sum(w, b, x, depth, relu)
        got = _lib.l1_embed(x.cuda(), depth, wt, b.cuda(), relu).cpu().numpy()
        assert got.shape == (m, n_pad)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (D, depth, m, relu, float(np.abs(got - want).max()))
    # bf16 output = the fp32 value rounded to nearest-even; fp16 planes = (fp16(v), fp16(v - fp16(v)))
    want = torch.from_numpy(_host_sum(w, b, x, depth, True))
    got16 = _lib.l1_embed(x.cuda(), depth, wt, b.cuda(), True, torch.bfloat16).cpu()
    assert torch.equal(got16.view(torch.int16), want.to(torch.bfloat16).view(torch.int16))
    # e4m3 output (the fp8 mode's operand): the fp32 value scaled into range by the caller, rounded to nearest-even, saturating
    big = 100.0
    want8 = torch.from_numpy(_host_sum(w * big, b * big, x, depth, True)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    got8 = _lib.l1_embed(x.cuda(), depth, (w * big).t().contiguous().cuda(), (b * big).cuda(), True, torch.float8_e4m3fn).cpu()
    assert float(want8.float().max()) == 448.0 or m == 1  # (the saturating branch is exercised)
    assert torch.equal(got8.view(torch.uint8), want8.view(torch.uint8))
    ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
    pl = _lib.l1_embed(x.cuda(), depth, wt, b.cuda(), True, split="planes", overflow=ovf).cpu()
    hi = want.to(torch.float16)
    lo = (want - hi.float()).to(torch.float16)
    assert pl.shape == (2, m, n_pad) and int(ovf.item()) == 0
    assert torch.equal(pl[0].view(torch.int16), hi.view(torch.int16)) and torch.equal(pl[1].view(torch.int16), lo.view(torch.int16))


@pytest.mark.parametrize("D,depth", [(49, 49), (36, 36), (25, 25)])
def test_embed_large_batches_same_bits_as_the_host_sum(D, depth):
    """Engine-sized batches: every wave walks many steps, workgroups wrap around the row slices, the last step is ragged — the
    same additions in the same order as for one row, so the same bits: checked against the host sum at 20 003 rows, every output
    form, and against a separate launch on a slice of the same rows."""
    from deepcubea_amd import _lib
    _lib.require_gpu()
    m, n_pad = 20003, 128
    w, b, x = _case(D, depth, m, n_pad, 4242 + D)
    wt, bc, xc = w.t().contiguous().cuda(), b.cuda(), x.cuda()
    want = _host_sum(w, b, x, depth, True)
    got = _lib.l1_embed(xc, depth, wt, bc, True)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert torch.equal(_lib.l1_embed(xc[4096:4096 + 1000].contiguous(), depth, wt, bc, True), got[4096:4096 + 1000])
    wantt = torch.from_numpy(want)
    assert torch.equal(_lib.l1_embed(xc, depth, wt, bc, True, torch.bfloat16).cpu().view(torch.int16), wantt.to(torch.bfloat16).view(torch.int16))
    ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
    pl = _lib.l1_embed(xc, depth, wt, bc, True, split="planes", overflow=ovf).cpu()
    hi = wantt.to(torch.float16)
    assert int(ovf.item()) == 0 and torch.equal(pl[0].view(torch.int16), hi.view(torch.int16))
    assert torch.equal(pl[1].view(torch.int16), (wantt - hi.float()).to(torch.float16).view(torch.int16))
    nr = _lib.l1_embed(xc, depth, wt, bc, False).cpu().numpy()
    assert np.array_equal(nr.view(np.uint32), _host_sum(w, b, x, depth, False).view(np.uint32))


def test_embed_overflow_flag_and_wide_layer():
    """A value beyond fp16 raises the planes' overflow flag (FastResnet then redoes the batch in fp32); the network's real
    width (5120 columns: 80 column tiles, every workgroup walking several row chunks)."""
    from deepcubea_amd import _lib
    _lib.require_gpu()
    D = depth = 16
    w, b, x = _case(D, depth, 5000, 5120, 77)
    wt = w.t().contiguous().cuda()
    got = _lib.l1_embed(x.cuda(), depth, wt, b.cuda(), True).cpu().numpy()
    want = _host_sum(w, b, x, depth, True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.l1_embed(x.cuda(), depth, wt, b.cuda(), True, split="planes", overflow=ovf)
    assert int(ovf.item()) == 0
    b2 = b.clone()
    b2[4097] = 70000.0
    _lib.l1_embed(x.cuda(), depth, wt, b2.cuda(), True, split="planes", overflow=ovf)
    assert int(ovf.item()) == 1


def test_embed_rejects_what_it_cannot_do():
    from deepcubea_amd import _lib
    _lib.require_gpu()
    assert not _lib.l1_embed_supported(54, 7) and not _lib.l1_embed_supported(96, 6)
    w, b, x = _case(16, 16, 8, 64, 1)
    with pytest.raises(_lib.DcaError):
        _lib.l1_embed(torch.zeros(8, 17, dtype=torch.uint8).cuda(), 16, torch.zeros(17 * 16, 64).cuda(), b.cuda(), True)
    assert _lib.l1_embed(x[:0].cuda(), 16, w.t().contiguous().cuda(), b.cuda(), True).shape == (0, 64)
    # rows that do not start on a 16-byte boundary (a slice of a larger matrix): the wrapper hands the kernel an aligned copy
    w, b, x = _case(25, 25, 40, 64, 2)
    xc, wt, bc = x.cuda(), w.t().contiguous().cuda(), b.cuda()
    assert xc[3:].data_ptr() % 16 != 0
    assert torch.equal(_lib.l1_embed(xc[3:], 25, wt, bc, True), _lib.l1_embed(xc, 25, wt, bc, True)[3:])


@pytest.mark.parametrize("D,depth", [(16, 16), (49, 49), (54, 6)])
def test_embed_against_the_onehot_mfma_kernel_and_float64(D, depth):
    """Same layer, two kernels: both within a few fp32 roundings of the float64 value; the embedding sum (exact fp32 operands,
    D + 1 roundings) is the closer one or level."""
    from deepcubea_amd import _lib
    from deepcubea_amd.utils.pytorch_models import l1_weight_tiles
    _lib.require_gpu()
    m, n_pad = 3000, 256
    w, b, x = _case(D, depth, m, n_pad, 5)
    xs = x.numpy().astype(np.int64)
    want = np.broadcast_to(b.double().numpy(), (m, n_pad)).copy()
    mag = np.broadcast_to(b.abs().double().numpy(), (m, n_pad)).copy()
    wt64 = w.double().t().numpy()
    for pos in range(D):
        want += wt64[pos * depth + xs[:, pos]]
        mag += np.abs(wt64[pos * depth + xs[:, pos]])
    want = np.maximum(want, 0.0)
    emb = _lib.l1_embed(x.cuda(), depth, w.t().contiguous().cuda(), b.cuda(), True).cpu().double().numpy()
    tiles = l1_weight_tiles(w, 3, _lib.l1_kpad(D, depth)).cuda()
    mf = _lib.l1_onehot_gemm(x.cuda(), depth, tiles, 3, b.cuda(), True, torch.float32).cpu().double().numpy()
    bound = mag * (D + 1) * 2.0 ** -24
    assert np.all(np.abs(emb - want) <= bound)
    assert np.all(np.abs(mf - want) <= 4 * bound)
    assert np.abs(emb - want).max() <= np.abs(mf - want).max() * 1.5 + 1e-12


@torch.no_grad()
def test_network_with_the_embedding_layer_matches_the_mfma_layer_and_is_batch_independent():
    """FastResnet(l1='embed') vs FastResnet(l1='mfma') on a puzzle15 network, fp32 parity mode and bf16: the parity mode within
    1e-5 of each other; a state's value has the same bits whatever batch it sits in; 'auto' picks the embedding sum for the
    puzzles' fp32 mode and the MFMA kernel for cube3."""
    from deepcubea_amd import _lib
    from deepcubea_amd.utils.pytorch_models import FastResnet, ResnetModel, L1_EMBED_MIN_DEPTH
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    _lib.require_gpu()
    net = ResnetModel(16, 16, 5000, 1000, 4, 1, True)
    load_synthetic_weights(net, 2025)
    net = net.eval()
    g = torch.Generator().manual_seed(3)
    x = torch.stack([torch.randperm(16, generator=g) for _ in range(4099)]).to(torch.uint8).cuda()
    fe, fm = FastResnet(net, l1="embed").cuda(), FastResnet(net, l1="mfma").cuda()
    assert fe.l1_embed_w is not None and fm.l1_embed_w is None and fe.uses_l1_kernel
    ye, ym = fe(x)[:, 0], fm(x)[:, 0]
    assert fe.split_fallbacks == 0
    assert float((ye - ym).abs().max()) < 1e-5 * max(1.0, float(ym.abs().max()))
    sub = x[1000:1777].contiguous()
    assert torch.equal(fe(sub)[:, 0], ye[1000:1777])
    assert torch.equal(fe(x[4098:].contiguous())[:, 0], ye[4098:])
    be, bm = FastResnet(net, torch.bfloat16, l1="embed").cuda(), FastResnet(net, torch.bfloat16, l1="mfma").cuda()
    d = float((be(x)[:, 0] - bm(x)[:, 0]).abs().max())
    assert d < 0.05 * max(1.0, float(ym.abs().max())), d  # both bf16 networks; layer 1 differs only in the accumulation order
    auto = FastResnet(net).cuda()
    assert (auto.l1_embed_w is not None) == (16 >= L1_EMBED_MIN_DEPTH[torch.float32])
    cube = ResnetModel(54, 6, 5000, 1000, 4, 1, True)
    assert FastResnet(cube.eval()).l1_embed_w is None


@torch.no_grad()
def test_lightsout_geometry_has_only_the_embedding_kernel():
    """(49, 6) — lightsout7 — has no one-hot MFMA instantiation in the fp32 / bf16 modes: its first layer ran on materialised
    one-hot rows; in the fp32 parity mode `l1="auto"` now feeds the uint8 rows to dca_l1_embed instead (18 % off the forward).
    Same network, within 1e-5; and the engine's packed uint8 rows drive it to the same values as the kept children directly."""
    from deepcubea_amd import _lib
    from deepcubea_amd.utils.pytorch_models import FastResnet, ResnetModel
    from deepcubea_amd.utils.synthetic_weights import load_synthetic_weights
    _lib.require_gpu()
    assert not _lib.l1_supported(49, 6) and _lib.l1_embed_supported(49, 6)
    net = ResnetModel(49, 6, 5000, 1000, 4, 1, True)  # lights_out.py:80-83
    load_synthetic_weights(net, 7)
    net = net.eval()
    x = torch.randint(0, 2, (3001, 49), generator=torch.Generator().manual_seed(1)).to(torch.uint8).cuda()
    dflt, emb = FastResnet(net, l1="mfma").cuda(), FastResnet(net).cuda()  # ("mfma": no embedding sum -> layer 1 on one-hot rows)
    assert not dflt.uses_l1_kernel and emb.uses_l1_kernel and emb.l1_tiles is None
    assert FastResnet(net, torch.bfloat16).l1_embed_w is None  # bf16: the one-hot rows + library GEMM are faster there
    yd, ye = dflt(x)[:, 0], emb(x)[:, 0]
    assert emb.split_fallbacks == 0
    assert float((yd - ye).abs().max()) < 1e-5 * max(1.0, float(yd.abs().max()))
    assert torch.equal(emb(x[77:1500].contiguous())[:, 0], ye[77:1500])
    # the engine's dedup-first stepping hands the network its packed uint8 rows (lights_out.py state bytes)
    from deepcubea_amd.search_methods.engine import BwasEngine
    from deepcubea_amd.utils import nnet_utils
    hfn = nnet_utils.get_heuristic_fn_dev(emb, clip_zero=False, batch_size=1 << 17)
    eng = BwasEngine("lightsout7", 0.8, 256, max_nodes=1 << 20, packed=True)
    root = torch.zeros(49, dtype=torch.uint8)
    root[[3, 10, 11, 24, 30, 41]] = 1
    eng.reset(root.numpy())
    eng.root_commit(hfn(eng.root_nnet_in()))
    for it in range(4):
        nn, oh, src, rows = eng.pop_expand_packed()
        assert oh is None and rows > 0
        kept = eng.last_children()[src[:rows].long()].contiguous()
        n = (rows + 1023) // 1024 * 1024
        h = hfn(nn[:n])
        assert float((h[:rows] - emb(kept)[:, 0]).abs().max()) < 1e-5
        assert float((h[:rows] - dflt(kept)[:, 0]).abs().max()) < 1e-5 * max(1.0, float(h[:rows].abs().max()))
        eng.commit_packed(h.float().contiguous())
    eng.close()
