"""ctypes binding of oracle/liboracle.so and oracle/_ref/libref_env.so — TEST INFRASTRUCTURE ONLY.

liboracle.so   = this build's C++ restatement of the reference hot path (oracle/dca_oracle.cpp).
libref_env.so  = the REFERENCE's cpp/environments.cpp compiled in place (oracle/Makefile `ref`).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Callable, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libref_env.so")

ENV_IDS = {"cube3": (0, 0), "puzzle15": (1, 4), "puzzle24": (1, 5), "puzzle35": (1, 6), "puzzle48": (1, 7),
           "lightsout7": (2, 7), "cube4": (3, 0)}


def num_moves(env: str) -> int:
    e, d = ENV_IDS[env]
    return 12 if e == 0 else 24 if e == 3 else d * d if e == 2 else 4
SEM_PY, SEM_CPP = 0, 1


def build(ref: bool = True) -> None:
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.isdir("/root/reference/cpp"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


class OracleResult(C.Structure):
    _fields_ = [("solved", C.c_int32), ("num_moves", C.c_int32), ("path_cost", C.c_double),
                ("nodes_generated", C.c_int64), ("iterations", C.c_int64), ("nodes_expanded", C.c_int64),
                ("seconds", C.c_double), ("open_size", C.c_int64), ("closed_size", C.c_int64)]


HEUR_CB = C.CFUNCTYPE(None, C.POINTER(C.c_uint8), C.c_int64, C.c_int, C.POINTER(C.c_float), C.c_void_p)

_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build(ref=False)
        _lib = C.CDLL(_LIB)
        _lib.oracle_cube3_perm_table.restype = C.POINTER(C.c_uint8)
        _lib.oracle_astar.restype = C.c_int
    return _lib


def ref_lib():
    """The reference's own C++ environments, or None when oracle/_ref was never built."""
    global _ref
    if _ref is None:
        if not os.path.exists(_REF):
            if os.path.isdir("/root/reference/cpp"):
                build(ref=True)
            else:
                return None
        _ref = C.CDLL(_REF)
    return _ref


def _p8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def cube3_perm_table() -> np.ndarray:
    return np.ctypeslib.as_array(lib().oracle_cube3_perm_table(), (12, 54)).copy()


def npuzzle_swap_table(dim: int) -> np.ndarray:
    out = np.zeros((dim * dim, 4), np.uint8)
    lib().oracle_npuzzle_swap_table(dim, _p8(out))
    return out


def next_state(env: str, states: np.ndarray, action: int) -> np.ndarray:
    e, d = ENV_IDS[env]
    s = np.ascontiguousarray(states, np.uint8)
    out = np.empty_like(s)
    lib().oracle_next_state(e, d, _p8(s), C.c_int64(s.shape[0]), action, _p8(out))
    return out


def expand(env: str, states: np.ndarray, threads: int = 0):
    """-> children [n,A,D], solved [n*A] bool, hash [n*A] u64."""
    e, d = ENV_IDS[env]
    s = np.ascontiguousarray(states, np.uint8)
    n, D = s.shape
    A = num_moves(env)
    ch = np.empty((n, A, D), np.uint8)
    sv = np.empty(n * A, np.uint8)
    hs = np.empty(n * A, np.uint64)
    lib().oracle_expand(e, d, _p8(s), C.c_int64(n), _p8(ch), _p8(sv), hs.ctypes.data_as(C.c_void_p), threads)
    return ch, sv.astype(bool), hs


def is_solved(env: str, states: np.ndarray) -> np.ndarray:
    e, d = ENV_IDS[env]
    s = np.ascontiguousarray(states, np.uint8)
    out = np.empty(s.shape[0], np.uint8)
    lib().oracle_is_solved(e, d, _p8(s), C.c_int64(s.shape[0]), _p8(out))
    return out.astype(bool)


def hash64(states: np.ndarray) -> np.ndarray:
    s = np.ascontiguousarray(states, np.uint8)
    out = np.empty(s.shape[0], np.uint64)
    lib().oracle_hash64(_p8(s), C.c_int64(s.shape[0]), s.shape[1], out.ctypes.data_as(C.c_void_p))
    return out


def heur_builtin(heur_id: int, states: np.ndarray) -> np.ndarray:
    s = np.ascontiguousarray(states, np.uint8)
    out = np.empty(s.shape[0], np.float32)
    lib().oracle_heur_builtin(heur_id, _p8(s), C.c_int64(s.shape[0]), s.shape[1], out.ctypes.data_as(C.c_void_p))
    return out


def nnet_input(env: str, states: np.ndarray) -> np.ndarray:
    e, _ = ENV_IDS[env]
    s = np.ascontiguousarray(states, np.uint8)
    out = np.empty_like(s)
    lib().oracle_nnet_input(e, _p8(s), C.c_int64(s.shape[0]), s.shape[1], _p8(out))
    return out


def onehot_f32(idx: np.ndarray, depth: int) -> np.ndarray:
    s = np.ascontiguousarray(idx, np.uint8)
    out = np.empty((s.shape[0], s.shape[1] * depth), np.float32)
    lib().oracle_onehot_f32(_p8(s), C.c_int64(s.shape[0]), s.shape[1], depth, out.ctypes.data_as(C.c_void_p))
    return out


def astar(env: str, root: np.ndarray, weight: float, batch: int, semantics: int = SEM_PY,
          heur_builtin_id: int = 0, heur_fn: Optional[Callable[[np.ndarray], np.ndarray]] = None,
          max_iters: int = 1 << 40, threads: int = 0, trace_cap: int = 0, stop_on_goal: bool = True):
    """Run the C++ restatement of BWAS.  heur_fn(states u8 [n,D]) -> f32 [n] overrides the builtin."""
    e, d = ENV_IDS[env]
    r = np.ascontiguousarray(root, np.uint8)
    res = OracleResult()
    moves = np.zeros(4096, np.int32)
    trace = np.zeros((max(trace_cap, 1), 3), np.int64)
    cb = HEUR_CB(0)
    if heur_fn is not None:
        def _cb(ptr, n, D, out, _user):
            st = np.ctypeslib.as_array(ptr, (n, D))
            hv = np.asarray(heur_fn(st), np.float32)
            np.ctypeslib.as_array(out, (n,))[:] = hv
        cb = HEUR_CB(_cb)
    rc = lib().oracle_astar(e, d, semantics, _p8(r), heur_builtin_id, cb, None, C.c_double(weight), batch,
                            C.c_int64(max_iters), threads, C.byref(res), moves.ctypes.data_as(C.c_void_p),
                            moves.size, trace.ctypes.data_as(C.c_void_p), C.c_int64(trace_cap), int(stop_on_goal))
    assert rc == 0
    out = {k: getattr(res, k) for k, _ in OracleResult._fields_}
    out["moves"] = moves[:res.num_moves].tolist() if res.solved else None
    out["trace"] = trace[:min(trace_cap, res.iterations)]
    return out


def num_threads() -> int:
    return int(lib().oracle_num_threads())


# ----------------------------------------------------------------- reference's own C++ envs
def ref_expand(env: str, states: np.ndarray, threads: int = 0):
    r = ref_lib()
    assert r is not None, "oracle/_ref/libref_env.so not built"
    e, d = ENV_IDS[env]
    s = np.ascontiguousarray(states, np.uint8)
    n, D = s.shape
    A = num_moves(env)
    ch = np.empty((n, A, D), np.uint8)
    sv = np.empty(n * A, np.uint8)
    r.ref_expand(e, d, _p8(s), C.c_int64(n), _p8(ch), _p8(sv), threads)
    return ch, sv.astype(bool)


def ref_next_state(env: str, states: np.ndarray, action: int) -> np.ndarray:
    r = ref_lib()
    assert r is not None
    e, d = ENV_IDS[env]
    s = np.ascontiguousarray(states, np.uint8)
    out = np.empty_like(s)
    r.ref_next_state(e, d, _p8(s), C.c_int64(s.shape[0]), action, _p8(out))
    return out
