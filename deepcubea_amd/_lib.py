"""ctypes binding of libdca_hip.so (the C ABI in include/dca.h).

There is NO CPU fallback: if the shared library is missing, or no HIP device is present when a
compute entry point is called, this module raises.  PyTorch-ROCm is used only for device memory,
streams and (elsewhere) the heuristic network.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdca_hip.so")

ENV_CUBE3, ENV_NPUZZLE, ENV_LIGHTSOUT, ENV_CUBE4 = 0, 1, 2, 3
DT_F32, DT_F16, DT_BF16, DT_F16X3, DT_F16_PLANES, DT_E4M3 = 0, 1, 2, 3, 4, 5
E4M3 = torch.float8_e4m3fn  # OCP e4m3: the fp8 format of gfx950's matrix pipes
SEM_PY, SEM_CPP = 0, 1
HEUR_MOD97, HEUR_KNUTH3, HEUR_HASHU01, HEUR_ZERO, HEUR_MANHATTAN = 0, 1, 2, 3, 4

_TORCH_DT = {torch.float32: DT_F32, torch.float16: DT_F16, torch.bfloat16: DT_BF16}

# every symbol include/dca.h (the product ABI) and include/dca_debug.h (test / tuning / profiling hooks) declare
# (tests check the library exports all of them)
ABI_SYMBOLS = [
    "dca_abi_version", "dca_last_error", "dca_cube3_perm_table", "dca_npuzzle_swap_table",
    "dca_cube3_next_state", "dca_cube3_prev_state", "dca_npuzzle_next_state", "dca_npuzzle_prev_state",
    "dca_cube3_expand_fused", "dca_npuzzle_expand_fused", "dca_is_solved", "dca_hash64", "dca_nnet_input",
    "dca_onehot", "dca_heuristic_builtin", "dca_generate_states", "dca_bellman_backup",
    "dca_engine_create", "dca_engine_create_multi", "dca_engine_num_instances", "dca_engine_destroy",
    "dca_engine_reset", "dca_engine_reset_instance", "dca_engine_root_commit", "dca_engine_root_commit_instance",
    "dca_engine_root_nnet_in", "dca_engine_root_nnet_in_instance", "dca_engine_status_instance",
    "dca_engine_solution_instance", "dca_engine_pop_expand", "dca_engine_commit", "dca_engine_run_builtin",
    "dca_engine_enable_packed", "dca_engine_pop_expand_packed", "dca_engine_commit_packed",
    "dca_engine_profile_builtin", "dca_engine_set_tiers", "dca_engine_debug", "dca_debug_tune", "dca_engine_status", "dca_engine_last_children", "dca_engine_solution",
    "dca_bn_workspace_bytes", "dca_bn_train_forward", "dca_bn_train_backward",
    "dca_l1_supported", "dca_l1_kpad", "dca_l1_onehot_gemm", "dca_act_split", "dca_f16x3_gemm", "dca_split_planes", "dca_f16x3_gemm_variant",
    "dca_absmax_bits", "dca_split_planes_scaled", "dca_split_rows_scaled", "dca_fill_inv_pow2", "dca_engine_plan_chunk",
    "dca_gemm16", "dca_gemm16_variant", "dca_gemm8", "dca_quant_e4m3", "dca_lightsout_next_state", "dca_lightsout_expand_fused",
    "dca_head_gemv", "dca_engine_packed_state", "dca_engine_info", "dca_gemm8_mx", "dca_l1_onehot_gemm_mx",
    "dca_cube4_perm_table", "dca_cube4_next_state", "dca_cube4_prev_state", "dca_cube4_expand_fused",
    "dca_engine_set_weight_instance", "dca_engine_set_weights", "dca_engine_park_instance", "dca_engine_last_popped",
    "dca_engine_reset_many", "dca_engine_root_commit_many", "dca_engine_set_weights_dev", "dca_debug_write_ceiling",
    "dca_l1_supported8", "dca_l1_kpad8", "dca_l1_onehot_gemm8", "dca_l1_embed_supported", "dca_l1_embed",
]


class DcaError(RuntimeError):
    pass


class DcaStatus(C.Structure):
    _fields_ = [("done", C.c_int32), ("failed", C.c_int32), ("iterations", C.c_int64),
                ("nodes_generated", C.c_int64), ("nodes_expanded", C.c_int64), ("open_size", C.c_int64),
                ("closed_size", C.c_int64), ("pool_size", C.c_int64), ("best_cost", C.c_double)]


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load libdca_hip.so; raise loudly if it was never built (python __graft_entry__.py builds it)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DcaError("HIP extension missing: %s — build it with `make -C deepcubea_amd/csrc` "
                           "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.dca_last_error.restype = C.c_char_p
        _lib.dca_cube3_perm_table.restype = C.POINTER(C.c_uint8)
        _lib.dca_cube4_perm_table.restype = C.POINTER(C.c_uint8)
        for name in ABI_SYMBOLS:
            fn = getattr(_lib, name)  # AttributeError here = stale build of libdca_hip.so
            if name not in ("dca_last_error", "dca_cube3_perm_table", "dca_cube4_perm_table", "dca_engine_destroy", "dca_bn_workspace_bytes",
                            "dca_l1_kpad", "dca_l1_kpad8"):
                fn.restype = C.c_int
        _lib.dca_l1_kpad.restype = C.c_int64
        _lib.dca_l1_kpad8.restype = C.c_int64
        _lib.dca_bn_workspace_bytes.restype = C.c_int64
        _lib.dca_bn_workspace_bytes.argtypes = [C.c_int64]
        _lib.dca_engine_destroy.restype = None
        if _lib.dca_abi_version() != 5:
            raise DcaError("libdca_hip.so ABI version mismatch")
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise DcaError("%s failed (rc=%d): %s" % (what or "dca call", rc, lib().dca_last_error().decode()))


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise DcaError("no HIP device visible: the deepcubea_amd hot path runs on an MI355X only (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return C.c_void_p(t.data_ptr())


def env_ids(env_name: str):
    """utils/env_utils.py:6-28 registry names -> (env id, dim, state_dim, num_moves, onehot_depth)."""
    import math
    import re
    name = env_name.lower()
    if name == "cube3":
        return ENV_CUBE3, 0, 54, 12, 6
    if name == "cube4":  # environment kernels only (no network / search driver: the reference's harness has none either)
        return ENV_CUBE4, 0, 96, 24, 6
    m = re.search(r"puzzle(\d+)", name)
    if m:
        dim = int(math.sqrt(int(m.group(1)) + 1))
        if 4 <= dim <= 7:
            return ENV_NPUZZLE, dim, dim * dim, 4, dim * dim
    m = re.search(r"lightsout(\d+)", name)
    if m and int(m.group(1)) == 7:
        return ENV_LIGHTSOUT, 7, 49, 49, 6
    raise ValueError("No known environment %s" % env_name)


def env_geometry(env: int, dim: int):
    """(state_dim, num_moves, onehot_depth) of an (env id, dim) pair."""
    if env == ENV_CUBE3:
        return 54, 12, 6
    if env == ENV_CUBE4:
        return 96, 24, 6
    if env == ENV_LIGHTSOUT:
        return dim * dim, dim * dim, 6
    return dim * dim, 4, dim * dim


# ------------------------------------------------------------------------------ tables (host)
def cube3_perm_table() -> np.ndarray:
    return np.ctypeslib.as_array(lib().dca_cube3_perm_table(), (12, 54)).copy()


def cube4_perm_table() -> np.ndarray:
    return np.ctypeslib.as_array(lib().dca_cube4_perm_table(), (24, 96)).copy()


def npuzzle_swap_table(dim: int) -> np.ndarray:
    out = np.zeros((dim * dim, 4), np.uint8)
    check(lib().dca_npuzzle_swap_table(dim, out.ctypes.data_as(C.c_void_p)), "dca_npuzzle_swap_table")
    return out


# ------------------------------------------------------------------------------ device ops
def _u8(t: torch.Tensor) -> torch.Tensor:
    assert t.dtype == torch.uint8 and t.is_cuda
    return t.contiguous()


def next_state(env: int, dim: int, states: torch.Tensor, action: int, prev: bool = False) -> torch.Tensor:
    states = _u8(states)
    out = torch.empty_like(states)
    n = states.shape[0]
    L = lib()
    if env == ENV_CUBE3:
        fn = L.dca_cube3_prev_state if prev else L.dca_cube3_next_state
        check(fn(ptr(states), C.c_int64(n), int(action), ptr(out), stream_ptr()), "dca_cube3_next_state")
    elif env == ENV_CUBE4:
        fn = L.dca_cube4_prev_state if prev else L.dca_cube4_next_state
        check(fn(ptr(states), C.c_int64(n), int(action), ptr(out), stream_ptr()), "dca_cube4_next_state")
    elif env == ENV_LIGHTSOUT:  # every move is its own inverse (lights_out.py:52-53)
        check(L.dca_lightsout_next_state(ptr(states), C.c_int64(n), dim, int(action), ptr(out), stream_ptr()),
              "dca_lightsout_next_state")
    else:
        fn = L.dca_npuzzle_prev_state if prev else L.dca_npuzzle_next_state
        check(fn(ptr(states), C.c_int64(n), dim, int(action), ptr(out), stream_ptr()), "dca_npuzzle_next_state")
    return out


def expand_fused(env: int, dim: int, parents: torch.Tensor, *, children: bool = True, nnet_in: bool = False,
                 onehot_dtype: Optional[torch.dtype] = None, solved: bool = True, hashes: bool = True,
                 out: Optional[dict] = None) -> dict:
    """One launch of the fused expansion.  Returns a dict of the requested device tensors.
    `out` may hold preallocated tensors (keys: children, nnet_in, onehot, solved, hash)."""
    parents = _u8(parents)
    n, D = parents.shape
    _, A, depth = env_geometry(env, dim)
    dev = parents.device
    out = dict(out) if out else {}
    if children and "children" not in out:
        out["children"] = torch.empty((n, A, D), dtype=torch.uint8, device=dev)
    if nnet_in and env == ENV_CUBE3 and "nnet_in" not in out:
        out["nnet_in"] = torch.empty((n * A, D), dtype=torch.uint8, device=dev)
    if onehot_dtype is not None and "onehot" not in out:
        out["onehot"] = torch.empty((n * A, D * depth), dtype=onehot_dtype, device=dev)
    if solved and "solved" not in out:
        out["solved"] = torch.empty((n * A,), dtype=torch.uint8, device=dev)
    if hashes and "hash" not in out:
        out["hash"] = torch.empty((n * A,), dtype=torch.int64, device=dev)  # bit pattern of the u64 hash
    oh = out.get("onehot")
    ohdt = _TORCH_DT[oh.dtype] if oh is not None else DT_F32
    L = lib()
    if env == ENV_CUBE4:
        assert oh is None and not nnet_in, "cube4 has no network input in the reference (environment kernels only)"
        check(L.dca_cube4_expand_fused(ptr(parents), C.c_int64(n), ptr(out.get("children")), ptr(out.get("solved")),
                                       ptr(out.get("hash")), stream_ptr()), "dca_cube4_expand_fused")
    elif env == ENV_CUBE3:
        check(L.dca_cube3_expand_fused(ptr(parents), C.c_int64(n), ptr(out.get("children")), ptr(out.get("nnet_in")),
                                       ptr(oh), ohdt, ptr(out.get("solved")), ptr(out.get("hash")), stream_ptr()),
              "dca_cube3_expand_fused")
    else:
        fn = L.dca_lightsout_expand_fused if env == ENV_LIGHTSOUT else L.dca_npuzzle_expand_fused
        check(fn(ptr(parents), C.c_int64(n), dim, ptr(out.get("children")), ptr(oh), ohdt,
                 ptr(out.get("solved")), ptr(out.get("hash")), stream_ptr()), "dca_npuzzle/lightsout_expand_fused")
        if nnet_in and out.get("children") is not None:
            out["nnet_in"] = out["children"].view(n * A, D)  # n_puzzle.py:84-89: the tiles themselves
    return out


def is_solved(env: int, dim: int, states: torch.Tensor) -> torch.Tensor:
    states = _u8(states)
    out = torch.empty((states.shape[0],), dtype=torch.uint8, device=states.device)
    check(lib().dca_is_solved(env, dim, ptr(states), C.c_int64(states.shape[0]), ptr(out), stream_ptr()),
          "dca_is_solved")
    return out


def hash64(states: torch.Tensor) -> torch.Tensor:
    states = _u8(states)
    out = torch.empty((states.shape[0],), dtype=torch.int64, device=states.device)
    check(lib().dca_hash64(ptr(states), C.c_int64(states.shape[0]), states.shape[1], ptr(out), stream_ptr()),
          "dca_hash64")
    return out


def nnet_input(env: int, dim: int, states: torch.Tensor) -> torch.Tensor:
    states = _u8(states)
    out = torch.empty_like(states)
    check(lib().dca_nnet_input(env, dim, ptr(states), C.c_int64(states.shape[0]), ptr(out), stream_ptr()),
          "dca_nnet_input")
    return out


def onehot(idx: torch.Tensor, depth: int, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    idx = _u8(idx)
    n, D = idx.shape
    out = torch.empty((n, D * depth), dtype=dtype, device=idx.device)
    check(lib().dca_onehot(ptr(idx), C.c_int64(n), D, depth, ptr(out), _TORCH_DT[dtype], stream_ptr()), "dca_onehot")
    return out


def heuristic_builtin(heur_id: int, states: torch.Tensor) -> torch.Tensor:
    states = _u8(states)
    out = torch.empty((states.shape[0],), dtype=torch.float32, device=states.device)
    check(lib().dca_heuristic_builtin(heur_id, ptr(states), C.c_int64(states.shape[0]), states.shape[1], ptr(out),
                                      stream_ptr()), "dca_heuristic_builtin")
    return out


def generate_states(env: int, dim: int, n: int, back_lo: int, back_hi: int, seed: int, index0: int = 0,
                    want_moves: bool = False):
    """Device random reverse walks from the goal -> (states [n,D] u8, num_back [n] i32, moves [n,back_hi] i8|None)."""
    dev = require_gpu()
    D = env_geometry(env, dim)[0]
    states = torch.empty((n, D), dtype=torch.uint8, device=dev)
    nb = torch.empty((n,), dtype=torch.int32, device=dev)
    mv = torch.full((n, max(back_hi, 1)), -1, dtype=torch.int8, device=dev) if want_moves else None
    check(lib().dca_generate_states(env, dim, C.c_int64(n), int(back_lo), int(back_hi), C.c_uint64(seed & (2**64 - 1)),
                                    C.c_int64(index0), ptr(states), ptr(nb), ptr(mv), max(back_hi, 1), stream_ptr()),
          "dca_generate_states")
    return states, nb, mv


def bellman_backup(h_children: torch.Tensor, solved_parent: Optional[torch.Tensor], num_moves: int,
                   clip_zero: bool = True):
    """-> (ctg_backup f32 [n], argmin i32 [n])  (search_utils.py:16-32, gbfs.py:108)."""
    h = h_children.to(torch.float32).contiguous()
    n = h.numel() // num_moves
    ctg = torch.empty((n,), dtype=torch.float32, device=h.device)
    am = torch.empty((n,), dtype=torch.int32, device=h.device)
    check(lib().dca_bellman_backup(ptr(h), ptr(solved_parent), C.c_int64(n), int(num_moves), int(clip_zero), ptr(ctg),
                                   ptr(am), stream_ptr()), "dca_bellman_backup")
    return ctg, am


# ------------------------------------------------------------------------------ training step
class _BnTrainFn(torch.autograd.Function):
    """BatchNorm1d (training statistics) [+ skip] [+ ReLU] through dca_bn_train_forward / backward."""

    @staticmethod
    def forward(ctx, x, skip, gamma, beta, eps, relu, stats_out):
        x = x.contiguous()
        n, c = x.shape
        ws = torch.empty(lib().dca_bn_workspace_bytes(c), dtype=torch.uint8, device=x.device)
        y = torch.empty_like(x)
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        var_u = torch.empty_like(mean)
        sk = skip.contiguous() if skip is not None else None
        check(lib().dca_bn_train_forward(ptr(x), ptr(sk), ptr(gamma.contiguous()), ptr(beta.contiguous()), C.c_int64(n),
                                         C.c_int64(c), C.c_double(eps), int(relu), ptr(y), ptr(mean), ptr(invstd), ptr(var_u),
                                         ptr(ws), C.c_int64(ws.numel()), stream_ptr()), "dca_bn_train_forward")
        stats_out.append((mean, var_u))
        ctx.save_for_backward(x, y, mean, invstd, gamma)
        ctx.relu, ctx.has_skip = bool(relu), skip is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, invstd, gamma = ctx.saved_tensors
        n, c = x.shape
        dy = dy.contiguous()
        ws = torch.empty(lib().dca_bn_workspace_bytes(c), dtype=torch.uint8, device=x.device)
        dx = torch.empty_like(x)
        dskip = torch.empty_like(x) if ctx.has_skip else None
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = torch.empty_like(dgamma)
        check(lib().dca_bn_train_backward(ptr(dy), ptr(x), ptr(y), ptr(mean), ptr(invstd), ptr(gamma.contiguous()),
                                          C.c_int64(n), C.c_int64(c), int(ctx.relu), ptr(dx), ptr(dskip), ptr(dgamma),
                                          ptr(dbeta), ptr(ws), C.c_int64(ws.numel()), stream_ptr()), "dca_bn_train_backward")
        return dx, dskip, dgamma, dbeta, None, None, None


def bn_train(x: torch.Tensor, bn: "torch.nn.BatchNorm1d", relu: bool, skip: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Training-mode `relu?(bn(x) (+ skip))` on the device, updating bn.running_mean / running_var /
    num_batches_tracked exactly like nn.BatchNorm1d (momentum average, unbiased variance)."""
    stats = []
    y = _BnTrainFn.apply(x, skip, bn.weight, bn.bias, float(bn.eps), bool(relu), stats)
    if bn.track_running_stats and bn.running_mean is not None:
        mean, var_u = stats[0]
        with torch.no_grad():
            bn.num_batches_tracked += 1
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1.0 - mom).add_(mean, alpha=mom)
            bn.running_var.mul_(1.0 - mom).add_(var_u, alpha=mom)
    return y


# ------------------------------------------------------------------------------ training step: dense layers
def _pad64(k: int) -> int:
    return (k + 63) // 64 * 64


def linear_f16x3(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], scale_a: bool,
                 amax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a [m, k] fp32 . w [n, k]^T (+ bias) -> [m, n] fp32 through dca_f16x3_gemm (fp32-accurate on the f16 matrix pipes),
    operands prepared on the spot: rows of w scaled by their own power of two (dca_split_rows_scaled), `a` by one power of two
    for the whole tensor when scale_a (gradients; dca_absmax_bits + dca_split_planes_scaled) — activations (O(1) after
    BatchNorm) are split as they are.  k, n % 4 == 0; K is zero-padded to a multiple of 64 inside the planes."""
    assert a.is_cuda and a.dtype == torch.float32 and w.dtype == torch.float32 and a.dim() == 2 and w.dim() == 2
    a, w = a.contiguous(), w.contiguous()
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k and k % 4 == 0 and n % 4 == 0
    kp = _pad64(k)
    dev = a.device
    if scale_a and amax is None:  # (the backward pass takes max|dy| once for both of its GEMMs)
        amax = torch.empty(1, dtype=torch.int32, device=dev)
        check(lib().dca_absmax_bits(ptr(a), C.c_int64(m), C.c_int64(k), C.c_int64(k), ptr(amax), stream_ptr()), "dca_absmax_bits")
    if not scale_a:
        amax = None
    ap = torch.empty((2, m, kp), dtype=torch.float16, device=dev)
    check(lib().dca_split_planes_scaled(ptr(a), C.c_int64(m), C.c_int64(k), C.c_int64(k), ptr(amax), ptr(ap[0]), ptr(ap[1]),
                                        C.c_int64(kp), C.c_int64(kp), stream_ptr()), "dca_split_planes_scaled")
    wp = torch.empty((2, n, kp), dtype=torch.float16, device=dev)
    cs = torch.empty(n, dtype=torch.float32, device=dev)
    check(lib().dca_split_rows_scaled(ptr(w), C.c_int64(n), C.c_int64(k), C.c_int64(k), ptr(wp[0]), ptr(wp[1]), C.c_int64(kp),
                                      C.c_int64(kp), ptr(cs), ptr(amax), stream_ptr()), "dca_split_rows_scaled")
    _, out = f16x3_gemm(ap, wp[0], wp[1], cs, 1.0, None if bias is None else bias.contiguous(), None, False, False, True)
    return out


def _absmax_bits(a: torch.Tensor) -> torch.Tensor:
    out = torch.empty(1, dtype=torch.int32, device=a.device)
    check(lib().dca_absmax_bits(ptr(a), C.c_int64(a.shape[0]), C.c_int64(a.shape[1]), C.c_int64(a.shape[1]), ptr(out), stream_ptr()),
          "dca_absmax_bits")
    return out


class _LinearTrainFn(torch.autograd.Function):
    """nn.Linear for the training step (reference nnet_utils.py:53-118 runs it as the library's fp32 GEMMs): forward and input
    gradient on dca_f16x3_gemm (2/3 of the layer's flops, ~3x the library's fp32 rate at fp32 accuracy).  The weight gradient
    dy^T . x contracts over the BATCH dimension and stays on the library's fp32 GEMM: a split-K form of the f16x3 kernel on
    operands transposed while they are split was built in round 4, exact to the same level and only a draw on time (faster
    GEMMs, but two transposes and the sum of the partials on top) — deleted in round 5."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        # scale_a: the activations get one power-of-two scale from their own magnitude, like the gradients (one extra pass
        # over x; without it a value beyond 65504 turns into inf in its fp16 plane and the loss into NaN without a word)
        return linear_f16x3(x, weight, bias, scale_a=True)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        amax = _absmax_bits(dy) if ctx.needs_input_grad[0] else None
        dx = linear_f16x3(dy, weight.t().contiguous(), None, scale_a=True, amax=amax) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dw = dy.t().mm(x)
        db = dy.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db


def linear_train(x: torch.Tensor, lin: "torch.nn.Linear") -> torch.Tensor:
    """`lin(x)` inside the training step.  The f16x3 path needs the GPU, fp32, 4-element-aligned widths and enough rows to
    fill a tile; anything else (the 1-wide output layer, the host) is F.linear."""
    if (x.is_cuda and x.dtype == torch.float32 and lin.weight.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 256
            and lin.in_features % 4 == 0 and lin.out_features % 4 == 0 and lin.in_features >= 64 and TRAIN_F16X3):
        return _LinearTrainFn.apply(x, lin.weight, lin.bias)
    return torch.nn.functional.linear(x, lin.weight, lin.bias)


TRAIN_F16X3 = os.environ.get("DCA_TRAIN_GEMM", "f16x3") != "library"  # A/B switches (bench.py --workload train, tools/train_grad_check.py)

# ------------------------------------------------------------------------------ heuristic network, layer 1
def l1_supported(state_dim: int, depth: int) -> bool:
    return bool(lib().dca_l1_supported(int(state_dim), int(depth)))


def l1_kpad(state_dim: int, depth: int) -> int:
    return int(lib().dca_l1_kpad(int(state_dim), int(depth)))


def l1_onehot_gemm(states_nnet: torch.Tensor, depth: int, w_tiles: torch.Tensor, planes: int, bias: torch.Tensor,
                   relu: bool, out_dtype, split=False, overflow: Optional[torch.Tensor] = None) -> torch.Tensor:
    """relu?(onehot(states_nnet) @ W1^T + b1) from the uint8 rows, [m, n_pad] in out_dtype (dca_l1_onehot_gemm);
    split=True: the library-GEMM f16x3 operand [m, 3*n_pad] fp16 of the next layer instead (DCA_DT_F16X3);
    split="planes": dca_f16x3_gemm's operand [2, m, n_pad] fp16 (high halves, low halves; DCA_DT_F16_PLANES)."""
    x = _u8(states_nnet)
    m, d = x.shape
    n_pad = bias.numel()
    if split == "planes":
        out = torch.empty((2, m, n_pad), dtype=torch.float16, device=x.device)
        code = DT_F16_PLANES
    elif split:
        out = torch.empty((m, 3 * n_pad), dtype=torch.float16, device=x.device)
        code = DT_F16X3
    elif out_dtype == E4M3:  # (planes == 1; the caller has folded the activation scale into the weights and bias)
        out = torch.empty((m, n_pad), dtype=E4M3, device=x.device)
        code = DT_E4M3
    else:
        out = torch.empty((m, n_pad), dtype=out_dtype, device=x.device)
        code = _TORCH_DT[out_dtype]
    check(lib().dca_l1_onehot_gemm(ptr(x), C.c_int64(m), int(d), int(depth), ptr(w_tiles), int(planes), C.c_int64(n_pad),
                                   ptr(bias), int(relu), ptr(out), code, ptr(overflow), stream_ptr()), "dca_l1_onehot_gemm")
    return out


def l1_supported8(state_dim: int, depth: int) -> bool:
    return bool(lib().dca_l1_supported8(int(state_dim), int(depth)))


def l1_kpad8(state_dim: int, depth: int) -> int:
    return int(lib().dca_l1_kpad8(int(state_dim), int(depth)))


def l1_onehot_gemm8(states_nnet: torch.Tensor, depth: int, w_tiles8: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor,
                    relu: bool) -> torch.Tensor:
    """Layer 1 of the fp8 mode on the f8f6f4 pipe (dca_l1_onehot_gemm8): e4m3(sat(relu?((onehot(s) . w8^T) * scale + bias))),
    [m, n_pad] e4m3, from the uint8 rows; w_tiles8 from pytorch_models.l1_weight_tiles8."""
    x = _u8(states_nnet)
    m, d = x.shape
    n_pad = bias.numel()
    assert scale.dtype == torch.float32 and bias.dtype == torch.float32 and scale.numel() == n_pad and n_pad % 128 == 0
    out = torch.empty((m, n_pad), dtype=E4M3, device=x.device)
    check(lib().dca_l1_onehot_gemm8(ptr(x), C.c_int64(m), int(d), int(depth), ptr(w_tiles8), C.c_int64(n_pad), ptr(scale),
                                    ptr(bias), int(relu), ptr(out), stream_ptr()), "dca_l1_onehot_gemm8")
    return out


def l1_embed_supported(state_dim: int, depth: int) -> bool:
    return bool(lib().dca_l1_embed_supported(int(state_dim), int(depth)))


def l1_embed(states_nnet: torch.Tensor, depth: int, w_t: torch.Tensor, bias: torch.Tensor, relu: bool, out_dtype=torch.float32,
             split=False, overflow: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Layer 1 as an embedding sum (dca_l1_embed): relu?(b1 + sum_pos w_t[pos * depth + s[pos]]) in exact fp32 arithmetic from
    the uint8 rows; w_t = W1^T fp32 [state_dim * depth, n_pad].  [m, n_pad] in out_dtype (fp32 / bf16 / e4m3 saturating), or split="planes":
    dca_f16x3_gemm's operand [2, m, n_pad] fp16."""
    x = _u8(states_nnet)
    if x.data_ptr() % 16:  # a row slice of a larger matrix: the kernel streams the rows in 16-byte pieces
        x = x.clone()
    m, d = x.shape
    n_pad = bias.numel()
    assert w_t.dtype == torch.float32 and bias.dtype == torch.float32 and w_t.is_contiguous() and tuple(w_t.shape) == (d * depth, n_pad)
    if split == "planes":
        out = torch.empty((2, m, n_pad), dtype=torch.float16, device=x.device)
        code = DT_F16_PLANES
    elif out_dtype == E4M3:  # (the caller has folded the activation scale into the weights and bias)
        out = torch.empty((m, n_pad), dtype=E4M3, device=x.device)
        code = DT_E4M3
    else:
        assert not split and out_dtype in (torch.float32, torch.bfloat16)
        out = torch.empty((m, n_pad), dtype=out_dtype, device=x.device)
        code = _TORCH_DT[out_dtype]
    if m == 0:
        return out
    check(lib().dca_l1_embed(ptr(x), C.c_int64(m), int(d), int(depth), ptr(w_t), C.c_int64(n_pad), ptr(bias), int(relu), ptr(out),
                             code, ptr(overflow), stream_ptr()), "dca_l1_embed")
    return out


def act_split(y: torch.Tensor, bias: Optional[torch.Tensor], skip: Optional[torch.Tensor], alpha, relu: bool,
              want_x: bool, want_a3=True, overflow: Optional[torch.Tensor] = None):
    """v = relu?(y*alpha + bias (+ skip)) -> (a3, v fp32 or None).  want_a3=True: the library-GEMM operand [m,3n] fp16,
    a3[3k..3k+2] = (vh, vl, vh); want_a3="planes": dca_f16x3_gemm's operand [2,m,n] fp16; False: None."""
    assert y.dtype == torch.float32 and y.is_contiguous() and (want_x or want_a3)
    m, n = y.shape
    planes = want_a3 == "planes"
    a3 = None
    if planes:
        a3 = torch.empty((2, m, n), dtype=torch.float16, device=y.device)
    elif want_a3:
        a3 = torch.empty((m, 3 * n), dtype=torch.float16, device=y.device)
    x_out = torch.empty_like(y) if want_x else None
    col_scale = alpha if isinstance(alpha, torch.Tensor) else None  # per-output-unit scale vector or one scalar
    check(lib().dca_act_split(ptr(y), ptr(bias), ptr(skip), ptr(col_scale), C.c_double(1.0 if col_scale is not None else alpha),
                              int(relu), C.c_int64(m), C.c_int64(n), ptr(x_out), ptr(a3), int(planes), ptr(overflow),
                              stream_ptr()), "dca_act_split")
    return a3, x_out


def split_planes(x: torch.Tensor, overflow: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 [m, n] -> its fp16 planes [2, m, n] (x = planes[0] + planes[1] to 22 bits): the operand of f16x3_gemm."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] % 4 == 0
    m, n = x.shape
    out = torch.empty((2, m, n), dtype=torch.float16, device=x.device)
    check(lib().dca_split_planes(ptr(x), C.c_int64(m), C.c_int64(n), C.c_int64(n), ptr(out[0]), ptr(out[1]), C.c_int64(n),
                                 ptr(overflow), stream_ptr()), "dca_split_planes")
    return out


def f16x3_gemm(a_planes: torch.Tensor, w_h: torch.Tensor, w_l: torch.Tensor, col_scale: Optional[torch.Tensor], alpha: float,
               bias: Optional[torch.Tensor], skip: Optional[torch.Tensor], relu: bool, want_planes: bool, want_x: bool,
               overflow: Optional[torch.Tensor] = None):
    """One dense layer of the cost-to-go network as the hand-written fp32-accurate MFMA kernel (dca_f16x3_gemm):
    v = relu?((a . w^T) * alpha * col_scale + bias (+ skip)).  a_planes [2, m, k] fp16, w_h / w_l [n, k] fp16.
    -> (planes of v [2, m, n] fp16 or None, v fp32 [m, n] or None)."""
    assert a_planes.dtype == torch.float16 and a_planes.dim() == 3 and a_planes.is_contiguous()
    assert w_h.dtype == torch.float16 and w_h.is_contiguous() and w_l.is_contiguous() and w_h.shape == w_l.shape
    _, m, k = a_planes.shape
    n = w_h.shape[0]
    assert w_h.shape[1] == k and (want_planes or want_x)
    planes = torch.empty((2, m, n), dtype=torch.float16, device=a_planes.device) if want_planes else None
    x_out = torch.empty((m, n), dtype=torch.float32, device=a_planes.device) if want_x else None
    check(lib().dca_f16x3_gemm(ptr(a_planes[0]), ptr(a_planes[1]), C.c_int64(m), int(k), C.c_int64(k), ptr(w_h), ptr(w_l),
                               int(n), C.c_int64(k), ptr(col_scale), C.c_double(alpha), ptr(bias), ptr(skip), int(relu),
                               ptr(planes[0]) if want_planes else C.c_void_p(0), ptr(planes[1]) if want_planes else C.c_void_p(0),
                               ptr(x_out), C.c_int64(n), ptr(overflow), stream_ptr()), "dca_f16x3_gemm")
    return planes, x_out


def gemm16(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], skip: Optional[torch.Tensor], relu: bool,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One dense layer in the non-parity 16-bit modes (dca_gemm16): relu?(a . w^T + bias (+ skip)), a [m, k] / w [n, k] /
    skip [m, n] bf16 or fp16, bias fp32, fp32 accumulation, result in the operands' type.  `out` may be `skip` (in place)."""
    assert a.dtype in (torch.bfloat16, torch.float16) and w.dtype == a.dtype and a.is_contiguous() and w.is_contiguous()
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k and (bias is None or (bias.dtype == torch.float32 and bias.numel() == n))
    assert skip is None or (skip.dtype == a.dtype and skip.shape == (m, n) and skip.is_contiguous())
    if out is None:
        out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    assert out.dtype == a.dtype and out.shape == (m, n) and out.is_contiguous() and out.data_ptr() != a.data_ptr()
    check(lib().dca_gemm16(ptr(a), C.c_int64(m), int(k), C.c_int64(k), ptr(w), int(n), C.c_int64(k), _TORCH_DT[a.dtype],
                           ptr(bias), ptr(skip), int(relu), ptr(out), C.c_int64(n), stream_ptr()), "dca_gemm16")
    return out


def gemm8(a: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, bias: Optional[torch.Tensor], skip: Optional[torch.Tensor],
          relu: bool, want16: bool, out8_scale: Optional[float], out16: Optional[torch.Tensor] = None):
    """One dense layer in the fp8 mode (dca_gemm8): v = relu?((a . w^T) * scale + bias (+ skip)), a [m, k] / w [n, k] e4m3, scale /
    bias [n] fp32, skip [m, n] bf16.  Returns (bf16 v or None, e4m3(sat(v * out8_scale)) or None); `out16` may be `skip`."""
    assert a.dtype == E4M3 and w.dtype == E4M3 and a.is_contiguous() and w.is_contiguous()
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k and scale.dtype == torch.float32 and scale.numel() == n
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() == n)
    assert skip is None or (skip.dtype == torch.bfloat16 and skip.shape == (m, n) and skip.is_contiguous())
    if want16 and out16 is None:
        out16 = torch.empty((m, n), dtype=torch.bfloat16, device=a.device)
    assert out16 is None or (out16.dtype == torch.bfloat16 and out16.shape == (m, n) and out16.is_contiguous())
    out8 = torch.empty((m, n), dtype=E4M3, device=a.device) if out8_scale is not None else None
    check(lib().dca_gemm8(ptr(a), C.c_int64(m), int(k), C.c_int64(k), ptr(w), int(n), C.c_int64(k), ptr(scale), ptr(bias), ptr(skip),
                          int(relu), ptr(out16), C.c_int64(n), ptr(out8), C.c_int64(n),
                          C.c_double(out8_scale if out8_scale is not None else 1.0), stream_ptr()), "dca_gemm8")
    return out16, out8


def gemm8_mx(a: torch.Tensor, a_scale: torch.Tensor, w: torch.Tensor, w_scale: torch.Tensor, bias: Optional[torch.Tensor],
             skip: Optional[torch.Tensor], relu: bool, want16: bool, want8: bool, out16: Optional[torch.Tensor] = None):
    """One dense layer in the block-scaled fp8 mode (dca_gemm8_mx): a [m, k] e4m3 with a_scale [m, k/64] E8M0 bytes, w [n, k] e4m3 with
    w_scale [n] fp32.  -> (bf16 v or None, e4m3 v blocks or None, their E8M0 scales [m, n/64] or None); `out16` may be `skip`."""
    assert a.dtype == E4M3 and w.dtype == E4M3 and a.is_contiguous() and w.is_contiguous()
    m, k = a.shape
    n = w.shape[0]
    assert a_scale.dtype == torch.uint8 and a_scale.shape == (m, k // 64) and a_scale.is_contiguous()
    assert w.shape[1] == k and w_scale.dtype == torch.float32 and w_scale.numel() == n
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() == n)
    assert skip is None or (skip.dtype == torch.bfloat16 and skip.shape == (m, n) and skip.is_contiguous())
    if want16 and out16 is None:
        out16 = torch.empty((m, n), dtype=torch.bfloat16, device=a.device)
    out8 = torch.empty((m, n), dtype=E4M3, device=a.device) if want8 else None
    out_sc = torch.empty((m, n // 64), dtype=torch.uint8, device=a.device) if want8 else None
    check(lib().dca_gemm8_mx(ptr(a), ptr(a_scale), C.c_int64(m), int(k), C.c_int64(k), C.c_int64(k // 64), ptr(w), int(n), C.c_int64(k),
                             ptr(w_scale), ptr(bias), ptr(skip), int(relu), ptr(out16), C.c_int64(n), ptr(out8), C.c_int64(n),
                             ptr(out_sc), C.c_int64(n // 64), stream_ptr()), "dca_gemm8_mx")
    return out16, out8, out_sc


def l1_onehot_gemm_mx(states_nnet: torch.Tensor, depth: int, w_tiles: torch.Tensor, bias: torch.Tensor, relu: bool):
    """Layer 1 (one bf16 weight plane) leaving as e4m3 + one E8M0 block scale per row and 64 units (dca_l1_onehot_gemm_mx)
    -> (e4m3 [m, n_pad], uint8 [m, n_pad / 64])."""
    x = _u8(states_nnet)
    m, d = x.shape
    n_pad = bias.numel()
    out = torch.empty((m, n_pad), dtype=E4M3, device=x.device)
    sc = torch.empty((m, n_pad // 64), dtype=torch.uint8, device=x.device)
    check(lib().dca_l1_onehot_gemm_mx(ptr(x), C.c_int64(m), int(d), int(depth), ptr(w_tiles), C.c_int64(n_pad), ptr(bias), int(relu),
                                      ptr(out), ptr(sc), C.c_int64(n_pad // 64), stream_ptr()), "dca_l1_onehot_gemm_mx")
    return out, sc


def quant_e4m3(x: torch.Tensor, scale: float) -> torch.Tensor:
    """fp32 / bf16 [m, n] -> e4m3(sat(x * scale)) (dca_quant_e4m3)."""
    assert x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous() and x.dim() == 2 and x.shape[1] % 4 == 0
    m, n = x.shape
    out = torch.empty((m, n), dtype=E4M3, device=x.device)
    check(lib().dca_quant_e4m3(ptr(x), _TORCH_DT[x.dtype], C.c_int64(m), C.c_int64(n), C.c_int64(n), C.c_double(scale), ptr(out),
                               C.c_int64(n), stream_ptr()), "dca_quant_e4m3")
    return out


def head_gemv(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """Output layer (dca_head_gemv): x [m, k] fp32 / fp16 / bf16 (rows may be strided), w [n_out, k] fp32, bias [n_out] fp32
    -> [m, n_out] fp32, summed in a fixed order (a row's bits do not depend on m or on the row's position)."""
    assert x.is_cuda and x.dim() == 2 and x.stride(1) == 1 and x.dtype in _TORCH_DT
    assert w.dtype == torch.float32 and w.is_contiguous() and w.shape[1] == x.shape[1]
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() == w.shape[0])
    m, k = x.shape
    out = torch.empty((m, w.shape[0]), dtype=torch.float32, device=x.device)
    if m == 0:
        return out
    check(lib().dca_head_gemv(C.c_void_p(x.data_ptr()), _TORCH_DT[x.dtype], C.c_int64(m), int(k), C.c_int64(x.stride(0)), ptr(w),
                              ptr(bias), int(w.shape[0]), ptr(out), stream_ptr()), "dca_head_gemv")
    return out


def gemm16_variant(v: int) -> None:
    """Test hook: 3 (default) = ping-pong schedule, swapped operand roles, lean tail per layer form; 2 = the same schedule with
    the general tail; 1 = two-stage loop (one drain + barrier per K-step).  Bit-identical results."""
    check(lib().dca_gemm16_variant(int(v)), "dca_gemm16_variant")


def f16x3_gemm_variant(v: int) -> None:
    """Test hook: 3 (default) = LDS-DMA 256x256 kernel on the ping-pong schedule, 2 = the same tile with two whole-K-step
    stages (bit-identical)."""
    check(lib().dca_f16x3_gemm_variant(int(v)), "dca_f16x3_gemm_variant")
