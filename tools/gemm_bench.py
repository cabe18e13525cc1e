#!/usr/bin/env python3
"""MFMA-pipe throughput of the hand-written f16x3 dense-layer kernel (both variants) next to round 1's arrangement (one
library f16 GEMM over the interleaved 3x operand + the dca_act_split glue kernel) on the cube3 network's layer shapes.
TFLOP/s are ISSUED f16 MFMA flops (3 products per useful one); `useful` = /3.  Candidates take turns over four rounds (the first
is a warm-up), medians reported.   python tools/gemm_bench.py [rows] [f16x3|16|e4m3]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcubea_amd import _lib  # noqa: E402
from deepcubea_amd.utils.pytorch_models import _pow2_scale, _split_f16  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 204800
only = sys.argv[2] if len(sys.argv) > 2 else ""  # "f16x3" | "16" | "e4m3": that section alone


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def interleaved(cands, rounds=4):
    """Median ms of each candidate over `rounds` rounds in which the candidates take turns (whatever is measured first in a
    process runs up to 5 % slower — clocks still ramping — so one pass in a fixed order mis-ranks kernels 3-6 % apart;
    round 0 is the warm-up and is dropped)."""
    ms = {name: [] for name, _ in cands}
    for r in range(rounds):
        for name, fn in cands:
            t = timed(fn)
            if r > 0:
                ms[name].append(t)
    return {name: sorted(v)[len(v) // 2] for name, v in ms.items()}


for n, k in ((1024, 1024), (1024, 5120)) if only in ("", "f16x3") else ():
    g = torch.Generator().manual_seed(n + k)
    x = torch.randn(m, k, generator=g).cuda()
    w = torch.randn(n, k, generator=g) / k ** 0.5
    sc = _pow2_scale(w)
    wh, wl = _split_f16(w, sc)
    inv = (1.0 / sc).cuda()
    b = torch.randn(n, generator=g).cuda()
    skip = torch.randn(m, n, generator=g).cuda()
    planes = _lib.split_planes(x)
    whc, wlc = wh.cuda().contiguous(), wl.cuda().contiguous()
    w3 = torch.stack([wh, wh, wl], dim=2).reshape(n, -1).contiguous().cuda()
    a3, _ = _lib.act_split(x, None, None, 1.0, False, False)
    flops = 2.0 * m * n * k * 3
    row = {"m": m, "n": n, "k": k}
    def hip_layer(v):
        def run():
            _lib.f16x3_gemm_variant(v)
            return _lib.f16x3_gemm(planes, whc, wlc, inv, 1.0, b, skip, True, True, True)
        return run

    def lib_layer():
        y = torch.mm(a3, w3.t(), out_dtype=torch.float32)
        return _lib.act_split(y, b, skip, inv, True, True)

    def hip_noskip(v, want_x):
        def run():
            _lib.f16x3_gemm_variant(v)
            return _lib.f16x3_gemm(planes, whc, wlc, inv, 1.0, b, None, True, True, want_x)
        return run

    res = interleaved([("hip_v3", hip_layer(3)), ("hip_v3_planes_only", hip_noskip(3, False)), ("hip_v2", hip_layer(2)),
                       ("library_gemm_plus_glue", lib_layer),
                       ("library_gemm_only", lambda: torch.mm(a3, w3.t(), out_dtype=torch.float32))])
    _lib.f16x3_gemm_variant(3)
    for name, ms in res.items():
        row[name + "_ms"] = round(ms, 4)
        row[name + "_mfma_tflops"] = round(flops / ms / 1e9, 1)
    print(json.dumps(row))
    del x, planes, a3, skip
    torch.cuda.empty_cache()

# ---- the 16-bit (non-parity) layer: dca_gemm16 with its tail in the epilogue vs the library's addmm_activation (+ the
# separate ReLU pass a residual layer needs there)
for dt, nm in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")) if only in ("", "16") else ():
    for n, k in ((1024, 1024), (1024, 5120)):
        g = torch.Generator().manual_seed(n + k)
        x = (torch.randn(m, k, generator=g) * 0.5).to(dt).cuda()
        w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dt).cuda()
        b32 = torch.randn(n, generator=g).cuda()
        bdt = b32.to(dt)
        skip = torch.randn(m, n, generator=g).to(dt).cuda()
        flops = 2.0 * m * n * k
        row = {"dtype": nm, "m": m, "n": n, "k": k}
        out, out2 = skip.clone(), skip.clone()

        def hip16(v, with_skip):
            def run():
                _lib.gemm16_variant(v)
                return _lib.gemm16(x, w, b32, out, True, out=out) if with_skip else _lib.gemm16(x, w, b32, None, True)
            return run

        res = interleaved([("hip_bias_relu", hip16(3, False)), ("hip_skip_relu", hip16(3, True)),
                           ("hip_general_tail_bias_relu", hip16(2, False)), ("hip_general_tail_skip_relu", hip16(2, True)),
                           ("library_bias_relu", lambda: torch._addmm_activation(bdt, x, w.t())),
                           ("library_skip_relu", lambda: out2.addmm_(x, w.t()).relu_())])
        _lib.gemm16_variant(3)
        for name, ms in res.items():
            row[name + "_ms"], row[name + "_tflops"] = round(ms, 4), round(flops / ms / 1e9, 1)
        print(json.dumps(row))
        del x, skip, out, out2
        torch.cuda.empty_cache()

# ---- the fp8 (e4m3) layer: dca_gemm8 (dequantise + tail + requantise in the epilogue) vs the library's scaled fp8 GEMM
# (torch._scaled_mm -> hipBLASLt, bf16 output, no tail)
E4M3 = torch.float8_e4m3fn
for n, k in ((1024, 1024), (1024, 5120)) if only in ("", "e4m3") else ():
    g = torch.Generator().manual_seed(n + k)
    x8 = torch.randn(m, k, generator=g).clamp(-448, 448).to(E4M3).cuda()
    w8 = torch.randn(n, k, generator=g).clamp(-448, 448).to(E4M3).cuda()
    sc = torch.full((n,), 1.0 / k ** 0.5).cuda()
    b32 = torch.randn(n, generator=g).cuda()
    skip = torch.randn(m, n, generator=g).to(torch.bfloat16).cuda()
    flops = 2.0 * m * n * k
    row = {"dtype": "e4m3", "m": m, "n": n, "k": k}
    out = skip.clone()
    cands = [("hip_bias_relu_to_e4m3", lambda: _lib.gemm8(x8, w8, sc, b32, None, True, False, 8.0)),
             ("hip_skip_relu_to_bf16_and_e4m3", lambda: _lib.gemm8(x8, w8, sc, b32, out, True, True, 8.0, out16=out))]
    # block-scaled form (dca_gemm8_mx): E8M0 scale per row and 64 elements in, e4m3 + scales out
    asc = torch.full((m, k // 64), 127, dtype=torch.uint8, device="cuda")
    out_b = skip.clone()
    cands += [("hip_mx_bias_relu_to_mx", lambda: _lib.gemm8_mx(x8, asc, w8, sc, b32, None, True, False, True)),
              ("hip_mx_skip_relu_to_bf16_and_mx", lambda: _lib.gemm8_mx(x8, asc, w8, sc, b32, out_b, True, True, True, out16=out_b)),
              ("hip_mx_bias_relu_to_bf16_only", lambda: _lib.gemm8_mx(x8, asc, w8, sc, b32, None, True, True, False))]
    one = torch.ones((), device="cuda")
    try:
        torch._scaled_mm(x8[:256], w8.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
        cands.append(("library_scaled_mm_to_bf16", lambda: torch._scaled_mm(x8, w8.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16)))
    except Exception as e:  # noqa: BLE001
        row["library_scaled_mm"] = "unavailable: %s" % str(e)[:80]
    for name, ms in interleaved(cands).items():
        row[name + "_ms"], row[name + "_tflops"] = round(ms, 4), round(flops / ms / 1e9, 1)
    print(json.dumps(row))
    del x8, w8, skip, out
    torch.cuda.empty_cache()

# ---- layer 1 of the fp8 mode (cube3, 324 -> 5120 from the uint8 rows, e4m3 out): the bf16-pipe kernel that only ROUNDS to
# e4m3 (round 5) vs the f8f6f4-pipe kernel with e4m3 weights (round 6)
if only in ("", "l1"):
    from deepcubea_amd.utils.pytorch_models import l1_weight_tiles, l1_weight_tiles8  # noqa: E402
    D, depth, n_pad = 54, 6, 5120
    g = torch.Generator().manual_seed(7)
    x = torch.randint(0, depth, (m, D), dtype=torch.uint8, generator=g).cuda()
    w1 = torch.randn(n_pad, D * depth, generator=g) * 8.0
    b1 = torch.randn(n_pad, generator=g).cuda()
    t16 = l1_weight_tiles(w1.to(torch.bfloat16).float(), 1, _lib.l1_kpad(D, depth)).cuda()
    sw = (w1.abs().amax(dim=1) / 448.0)
    t8 = l1_weight_tiles8((w1 / sw[:, None]).to(E4M3), _lib.l1_kpad8(D, depth)).cuda()
    swc = sw.float().cuda()
    res = interleaved([("bf16_pipe_e4m3_out", lambda: _lib.l1_onehot_gemm(x, depth, t16, 1, b1, True, _lib.E4M3)),
                       ("f8f6f4_pipe_e4m3_out", lambda: _lib.l1_onehot_gemm8(x, depth, t8, swc, b1, True)),
                       ("bf16_pipe_bf16_out", lambda: _lib.l1_onehot_gemm(x, depth, t16, 1, b1, True, torch.bfloat16))])
    row = {"layer": "l1 cube3 one-hot 324 -> 5120", "m": m}
    for name, ms in res.items():
        row[name + "_ms"] = round(ms, 4)
    print(json.dumps(row))
