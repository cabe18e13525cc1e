"""Host handle of the device-resident BWAS engine (libdca_hip.so, `dca_engine_*`).

Mirrors the surface of the reference's `AStar` (search_methods/astar.py:232-340) for ONE instance:
`step(heuristic)`, `has_found_goal()`, solution / node-count accessors — but all state (OPEN, CLOSED,
node pool) stays in HBM and a step enqueues kernels without reading anything back.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

from .. import _lib

_OH = {None: -1, torch.float32: _lib.DT_F32, torch.float16: _lib.DT_F16, torch.bfloat16: _lib.DT_BF16}


class BwasEngine:
    """One search instance on the current HIP device.

    semantics: _lib.SEM_PY reproduces search_methods/astar.py node for node; _lib.SEM_CPP reproduces
    cpp/parallel_weighted_astar.cpp (ties between equal float32 costs are broken by push order, where the
    reference's std::priority_queue order is unspecified — SURVEY §3.3)."""

    def __init__(self, env_name: str, weight: float, batch_size: int, max_nodes: int = 1 << 24,
                 semantics: int = _lib.SEM_PY, onehot_dtype: Optional[torch.dtype] = None, num_instances: int = 1,
                 packed: bool = False, onehot_stride: Optional[int] = None):
        """packed=True: dedup-first stepping — the CLOSED check runs before the heuristic and only the surviving
        children are handed out (`pop_expand_packed` / `commit_packed`; `step` picks the mode).  Same search as the
        reference's order (astar.py:272-282), ~15-40 % fewer network rows.  onehot_stride: elements per packed
        one-hot row (>= state_dim*depth, tail zero), e.g. FastResnet.in_pad."""
        _lib.require_gpu()
        self.env_id, self.dim, self.state_dim, self.num_moves, self.depth = _lib.env_ids(env_name)
        self.batch_size = int(batch_size)
        self.weight = float(weight)
        self.semantics = semantics
        self.onehot_dtype = onehot_dtype
        self.num_instances = int(num_instances)
        # K instances share every launch (grid.y = instance); their batch buffers are contiguous, instance-major
        self.m_capacity = self.batch_size * self.num_moves * self.num_instances
        self._h = C.c_void_p(0)
        self.packed = bool(packed)
        _lib.check(_lib.lib().dca_engine_create_multi(C.byref(self._h), self.env_id, self.dim, C.c_double(self.weight),
                                                      self.batch_size, C.c_int64(int(max_nodes)), semantics,
                                                      -1 if packed else _OH[onehot_dtype], self.num_instances),
                   "dca_engine_create_multi")
        self._zero_h = torch.zeros(1, dtype=torch.float32, device="cuda")
        self.rows_evaluated = 0  # network rows handed to heuristic closures by step()
        self.last_rows = -1      # packed stepping: rows of the last iteration (0 = nothing left to expand: every instance done)
        self.onehot_stride = self.state_dim * self.depth
        if packed:
            if onehot_stride is not None:
                self.onehot_stride = int(onehot_stride)
            elif onehot_dtype is not None:
                self.onehot_stride = (self.onehot_stride + 7) // 8 * 8
            _lib.check(_lib.lib().dca_engine_enable_packed(self._h, _OH[onehot_dtype], C.c_int64(self.onehot_stride)),
                       "dca_engine_enable_packed")
            self.packed_capacity = (self.m_capacity + 1023) // 1024 * 1024

    @staticmethod
    def bytes_per_node(env_name: str) -> int:
        """Device bytes one node id costs (dca_engine_create_multi's allocations): state row + g/parent/move/solved, the
        CLOSED table (16-byte slots, power of two >= 2 ids per id: up to 4 slots), four OPEN buffers of (key, id), the
        pop's scratch arrays (tmp key/id/bin/idx, ord key/id)."""
        D = _lib.env_ids(env_name)[2]
        return D + 10 + 4 * 16 + 4 + 4 * 12 + 18 + 12  # (+4: the list of CLOSED slots in use, what a reset clears)

    @staticmethod
    def auto_max_nodes(env_name: str, batch_size: int, num_instances: int = 1, fraction: float = 0.8,
                       sharers: int = 1) -> int:
        """Largest node pool that fits `fraction` of this process's share of the device's HBM (288 GB on an MI355X), split
        between the engine's instances; capped by the 31-bit node id.  `sharers`: processes that use this device (ranks
        mapped onto the same GPU, sharding.ranks_on_my_device()): each takes 1/sharers of the TOTAL memory at most — the
        free amount alone is a race between them — and never more than what is free right now."""
        _lib.require_gpu()
        free, total = torch.cuda.mem_get_info()
        budget = min(free, total // max(1, int(sharers)))
        n = int(budget * fraction) // (BwasEngine.bytes_per_node(env_name) * max(1, int(num_instances)))
        floor = int(batch_size) * _lib.env_ids(env_name)[3] * 4 + 64
        return max(floor, min(n, 0x7FFFFF00 // 2))  # (the CLOSED table's slot index is 32-bit: 2 slots per id)

    def close(self):
        if self._h:
            _lib.lib().dca_engine_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- search control ---------------------------------------------------------------------
    def reset(self, root: np.ndarray, instance: int = 0) -> None:
        r = np.ascontiguousarray(root, dtype=np.uint8)
        assert r.shape == (self.state_dim,)
        _lib.check(_lib.lib().dca_engine_reset_instance(self._h, int(instance), r.ctypes.data_as(C.c_void_p),
                                                        _lib.stream_ptr()), "dca_engine_reset_instance")

    def root_nnet_in(self, instance: int = 0) -> torch.Tensor:
        p = C.c_void_p(0)
        _lib.check(_lib.lib().dca_engine_root_nnet_in_instance(self._h, int(instance), C.byref(p)),
                   "dca_engine_root_nnet_in_instance")
        return _wrap_u8(p.value, (1, self.state_dim))

    def root_commit(self, h_root: torch.Tensor, instance: int = 0) -> None:
        h_root = h_root.to(torch.float32).contiguous()
        _lib.check(_lib.lib().dca_engine_root_commit_instance(self._h, int(instance), _lib.ptr(h_root),
                                                              _lib.stream_ptr()), "dca_engine_root_commit_instance")

    def pop_expand(self) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """-> (nnet_in [M,D] uint8 view, onehot [M, D*depth] view or None), M = batch*num_moves fixed."""
        pn, po, m = C.c_void_p(0), C.c_void_p(0), C.c_int64(0)
        _lib.check(_lib.lib().dca_engine_pop_expand(self._h, C.byref(pn), C.byref(po), C.byref(m),
                                                    _lib.stream_ptr()), "dca_engine_pop_expand")
        nn = _wrap_u8(pn.value, (m.value, self.state_dim))
        oh = None
        if po.value:
            oh = _wrap(po.value, (m.value, self.state_dim * self.depth), self.onehot_dtype)
        return nn, oh

    def commit(self, h: torch.Tensor) -> None:
        assert h.is_cuda and h.dtype == torch.float32 and h.numel() == self.m_capacity and h.is_contiguous()
        _lib.check(_lib.lib().dca_engine_commit(self._h, _lib.ptr(h), _lib.stream_ptr()), "dca_engine_commit")

    def pop_expand_packed(self) -> Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor, int]:
        """-> (nnet_in [cap,D], onehot [cap,stride] or None, src [cap], rows): the first `rows` rows are this
        iteration's kept children (all instances); cap = K*batch*num_moves rounded up to 1024."""
        pn, po, ps, m = C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), C.c_int64(0)
        _lib.check(_lib.lib().dca_engine_pop_expand_packed(self._h, C.byref(pn), C.byref(po), C.byref(ps), C.byref(m),
                                                           _lib.stream_ptr()), "dca_engine_pop_expand_packed")
        cap = self.packed_capacity
        nn = _wrap_u8(pn.value, (cap, self.state_dim))
        oh = _wrap(po.value, (cap, self.onehot_stride), self.onehot_dtype) if po.value else None
        src = _wrap(ps.value, (cap,), torch.int32)
        return nn, oh, src, int(m.value)

    def commit_packed(self, h: torch.Tensor) -> None:
        assert h.is_cuda and h.dtype == torch.float32 and h.is_contiguous()
        _lib.check(_lib.lib().dca_engine_commit_packed(self._h, _lib.ptr(h), _lib.stream_ptr()), "dca_engine_commit_packed")

    def set_weight(self, weight: float, instance: int = 0) -> None:
        """Weight of path cost of ONE instance (astar.py:196 `weights`: AStar takes a list, one per instance)."""
        _lib.check(_lib.lib().dca_engine_set_weight_instance(self._h, int(instance), C.c_double(float(weight))),
                   "dca_engine_set_weight_instance")

    def set_weights(self, weights) -> None:
        """Weights of path cost of instances 0 .. len(weights)-1 at once (between iterations)."""
        w = np.ascontiguousarray(weights, dtype=np.float64)
        _lib.check(_lib.lib().dca_engine_set_weights(self._h, w.ctypes.data_as(C.c_void_p), int(w.size)), "dca_engine_set_weights")

    def reset_many(self, roots_dev: torch.Tensor) -> None:
        """Instances 0..n-1 restart from the rows of a DEVICE tensor [n, D] uint8, the others are parked; nothing synchronises."""
        assert roots_dev.is_cuda and roots_dev.dtype == torch.uint8 and roots_dev.is_contiguous()
        assert roots_dev.dim() == 2 and roots_dev.shape[1] == self.state_dim and roots_dev.shape[0] <= self.num_instances
        _lib.check(_lib.lib().dca_engine_reset_many(self._h, _lib.ptr(roots_dev), int(roots_dev.shape[0]), _lib.stream_ptr()),
                   "dca_engine_reset_many")

    def root_commit_many(self, h_roots: torch.Tensor) -> None:
        h = h_roots.to(torch.float32).contiguous().view(-1)
        _lib.check(_lib.lib().dca_engine_root_commit_many(self._h, _lib.ptr(h), int(h.numel()), _lib.stream_ptr()),
                   "dca_engine_root_commit_many")

    def set_weights_dev(self, weights_dev: torch.Tensor) -> None:
        """Weights of path cost of instances 0..n-1 from a device float64 tensor (stream-ordered, no host sync)."""
        w = weights_dev.to(torch.float64).contiguous().view(-1)
        assert w.is_cuda and 1 <= w.numel() <= self.num_instances
        _lib.check(_lib.lib().dca_engine_set_weights_dev(self._h, _lib.ptr(w), int(w.numel()), _lib.stream_ptr()),
                   "dca_engine_set_weights_dev")

    def park(self, instance: int) -> None:
        """Mark an instance finished (its launches are no-ops) until it is reset again."""
        _lib.check(_lib.lib().dca_engine_park_instance(self._h, int(instance), _lib.stream_ptr()), "dca_engine_park_instance")

    def last_popped(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Between pop_expand and commit: (states u8 [K*batch, D], flags u8 [K*batch]) of the parents just popped,
        instance-major; flags 0 = none, 1 = popped, 2 = popped and solved."""
        n = self.num_instances * self.batch_size
        st = torch.empty((n, self.state_dim), dtype=torch.uint8, device="cuda")
        fl = torch.empty((n,), dtype=torch.uint8, device="cuda")
        _lib.check(_lib.lib().dca_engine_last_popped(self._h, _lib.ptr(st), _lib.ptr(fl), _lib.stream_ptr()),
                   "dca_engine_last_popped")
        return st, fl

    def packed_state(self) -> Tuple[int, int]:
        """(instances finished, instances failed) as the last pop_expand_packed found them — no extra host sync."""
        nd, nf = C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().dca_engine_packed_state(self._h, C.byref(nd), C.byref(nf)), "dca_engine_packed_state")
        return int(nd.value), int(nf.value)

    def info(self) -> dict:
        out = (C.c_int64 * 8)()
        _lib.check(_lib.lib().dca_engine_info(self._h, out, _lib.stream_ptr()), "dca_engine_info")
        return dict(zip(["collect_blocks", "collect_resident", "grid_refinement", "closed_table_bytes", "coop_off"], list(out)))

    def step(self, heuristic_fn_dev: Callable[[torch.Tensor], torch.Tensor]) -> None:
        """One BWAS iteration (AStar.step, astar.py:256-317) with a device heuristic closure."""
        if self.packed:
            nn, oh, _, rows = self.pop_expand_packed()
            self.rows_evaluated += rows
            self.last_rows = rows
            if rows == 0:
                self.commit_packed(self._zero_h)
                return
            n = min((rows + 1023) // 1024 * 1024, self.packed_capacity)  # few distinct GEMM shapes
            try:  # rows past `rows` are padding (stale or zero): a model that calibrates on its input must not look at them
                heuristic_fn_dev.valid_rows = rows
            except AttributeError:
                pass
            h = heuristic_fn_dev(oh[:n], True) if oh is not None else heuristic_fn_dev(nn[:n])
            assert h.shape[0] >= rows
            self.commit_packed(h.to(torch.float32).contiguous())
            return
        self.rows_evaluated += self.m_capacity
        nn, oh = self.pop_expand()
        h = heuristic_fn_dev(oh, True) if oh is not None else heuristic_fn_dev(nn)
        self.commit(h.to(torch.float32).contiguous())

    def run_builtin(self, heur_id: int, iters: int, use_graph: bool = False) -> None:
        _lib.check(_lib.lib().dca_engine_run_builtin(self._h, heur_id, int(iters), int(use_graph),
                                                     _lib.stream_ptr()), "dca_engine_run_builtin")

    PROF_SLOTS = ["refill_hist", "refill_scan", "refill_move", "sel_hist", "sel_scan", "sel_collect", "rank", "expand",
                  "probe", "decide", "pack", "commit", "rank_small", "rank_big", "rank_big_load", "rank_big_count",
                  "rank_big_scatter", "rank_big_order"]

    def profile_builtin(self, heur_id: int, iters: int, use_graph: bool = True) -> dict:
        """Device-side profile of `iters` built-in iterations (graph replays by default): per launch the busy span
        (max workgroup end - min workgroup start on the device wall clock) and the idle gap in front of it, in ms per
        iteration -> {"span_ms": {launch: ms}, "gap_ms": {launch: ms}}.  Launches that did not run are omitted."""
        n = len(self.PROF_SLOTS)
        span, gap = (C.c_float * n)(), (C.c_float * n)()
        _lib.check(_lib.lib().dca_engine_profile_builtin(self._h, heur_id, int(iters), int(use_graph), span, gap,
                                                         _lib.stream_ptr()), "dca_engine_profile_builtin")
        it = max(iters, 1)
        return {"span_ms": {nm: span[k] / it for k, nm in enumerate(self.PROF_SLOTS) if span[k] > 0},
                "gap_ms": {nm: gap[k] / it for k, nm in enumerate(self.PROF_SLOTS) if span[k] > 0}}

    def set_tiers(self, front_keep: int, front_max: int) -> None:
        """Override the FRONT-tier hysteresis (tests use tiny values to force constant refills / spills)."""
        _lib.check(_lib.lib().dca_engine_set_tiers(self._h, C.c_int64(front_keep), C.c_int64(front_max)),
                   "dca_engine_set_tiers")

    def debug(self) -> dict:
        out = (C.c_double * 16)()
        _lib.check(_lib.lib().dca_engine_debug(self._h, out, _lib.stream_ptr()), "dca_engine_debug")
        names = ["front_n", "back_n", "front_cmin", "front_cmax", "back_cmin", "back_cmax", "T", "want", "bstar",
                 "n_ord", "max_bin", "giant_bins_seen", "max_sub", "spill_bin", "npop", "m"]
        return dict(zip(names, list(out)))

    def status(self, instance: int = 0) -> dict:
        st = _lib.DcaStatus()
        _lib.check(_lib.lib().dca_engine_status_instance(self._h, int(instance), C.byref(st), _lib.stream_ptr()),
                   "dca_engine_status_instance")
        return {k: getattr(st, k) for k, _ in _lib.DcaStatus._fields_}

    def last_children(self) -> torch.Tensor:
        p, m = C.c_void_p(0), C.c_int64(0)
        _lib.check(_lib.lib().dca_engine_last_children(self._h, C.byref(p), C.byref(m), _lib.stream_ptr()),
                   "dca_engine_last_children")
        return _wrap_u8(p.value, (m.value, self.state_dim))

    def solution(self, instance: int = 0) -> Tuple[List[int], float]:
        moves = np.zeros(4096, np.int32)
        n, pc = C.c_int(0), C.c_double(0)
        _lib.check(_lib.lib().dca_engine_solution_instance(self._h, int(instance), moves.ctypes.data_as(C.c_void_p),
                                                           moves.size, C.byref(n), C.byref(pc), _lib.stream_ptr()),
                   "dca_engine_solution_instance")
        return moves[:n.value].tolist(), float(pc.value)

    # ---- convenience drivers -----------------------------------------------------------------
    def solve_builtin(self, root: np.ndarray, heur_id: int, max_iters: int = 1 << 30, chunk: int = 16,
                      use_graph: bool = False) -> dict:
        """Full search with a built-in heuristic; the host only looks at the done flag every `chunk`
        iterations (iterations after `done` are device-side no-ops, so node counts stay exact)."""
        self.reset(root)
        if self.semantics == _lib.SEM_PY:
            h0 = _lib.heuristic_builtin(heur_id, torch.from_numpy(np.ascontiguousarray(root, np.uint8)[None]).cuda())
            self.root_commit(h0)
        it = 0
        while it < max_iters:
            n = min(chunk, max_iters - it)
            self.run_builtin(heur_id, n, use_graph)
            it += n
            st = self.status()
            if st["done"]:
                break
        return self._result()

    def solve(self, root: np.ndarray, heuristic_fn_dev, max_iters: int = 1 << 30) -> dict:
        """Full search with a device heuristic closure (e.g. nnet_utils.get_heuristic_fn_dev)."""
        self.reset(root)
        if self.semantics == _lib.SEM_PY:
            self.root_commit(heuristic_fn_dev(self.root_nnet_in()).to(torch.float32))
        for it in range(max_iters):
            self.step(heuristic_fn_dev)
            # Packed stepping: the done / failed flags come back with the packed row count every iteration (one host sync,
            # which the row count needs anyway), so a search that finishes or fails is noticed at the next iteration's pop.
            if self.packed:
                if self.packed_state()[0] >= self.num_instances:
                    break
                continue
            if self.status()["done"]:
                break
        return self._result()

    def _result(self, instance: int = 0) -> dict:
        st = self.status(instance)
        res = dict(st)
        res["solved"] = bool(st["done"] and not st["failed"])
        if res["solved"]:
            res["moves"], res["path_cost"] = self.solution(instance)
        else:
            res["moves"], res["path_cost"] = None, float("nan")
        return res

    # ---- K instances at once (per-instance sharding inside one GPU) ---------------------------
    def solve_many_builtin(self, roots, heur_id: int, max_iters: int = 1 << 30, chunk: int = 16,
                           use_graph: bool = False, on_done=None) -> List[dict]:
        """len(roots) <= num_instances searches stepped together; returns one result dict per root.  on_done(i) is
        called once, at the first poll (every `chunk` iterations) that finds instance i finished."""
        k = len(roots)
        assert 1 <= k <= self.num_instances
        for i, root in enumerate(roots):
            self.reset(root, i)
            if self.semantics == _lib.SEM_PY:
                h0 = _lib.heuristic_builtin(heur_id, torch.from_numpy(np.ascontiguousarray(root, np.uint8)[None]).cuda())
                self.root_commit(h0, i)
        it = 0
        finished = [False] * k
        while it < max_iters:
            n = min(chunk, max_iters - it)
            self.run_builtin(heur_id, n, use_graph)
            it += n
            for i in range(k):
                if not finished[i] and self.status(i)["done"]:
                    finished[i] = True
                    if on_done is not None:
                        on_done(i)
            if all(finished):
                break
        return [self._result(i) for i in range(k)]

    def solve_many(self, roots, heuristic_fn_dev, max_iters: int = 1 << 30, on_done=None) -> List[dict]:
        """Same with a device heuristic closure: ONE network call per iteration evaluates the children of all
        instances (their batch buffers are contiguous).  on_done(i) is called once when instance i finishes."""
        k = len(roots)
        assert 1 <= k <= self.num_instances
        for i, root in enumerate(roots):
            self.reset(root, i)
            if self.semantics == _lib.SEM_PY:
                self.root_commit(heuristic_fn_dev(self.root_nnet_in(i)).to(torch.float32), i)
        finished = [False] * k
        for _ in range(max_iters):
            self.step(heuristic_fn_dev)
            for i in range(k):
                if not finished[i] and self.status(i)["done"]:
                    finished[i] = True
                    if on_done is not None:
                        on_done(i)
            if all(finished):
                break
        return [self._result(i) for i in range(k)]


def _wrap(ptr: int, shape, dtype: torch.dtype) -> torch.Tensor:
    """Zero-copy torch view of library-owned device memory (valid until the next pop_expand)."""
    esz = torch.empty(0, dtype=dtype).element_size()
    n = int(np.prod(shape))

    class _Cai:
        __cuda_array_interface__ = {"shape": (n * esz,), "typestr": "|u1", "data": (ptr, False), "version": 2}
    t = torch.as_tensor(_Cai(), device="cuda")
    return t.view(dtype).view(*shape)


def _wrap_u8(ptr: int, shape) -> torch.Tensor:
    return _wrap(ptr, shape, torch.uint8)
