"""Environment / State API of the BWAS path, mirroring the reference seam
`environments/environment_abstract.py:8-163` (same method names, argument meaning and errors), with
array-native companions (`*_np`, `*_dev`) that the HIP search engine uses directly.

All batched work runs on the GPU through libdca_hip.so; there is no CPU implementation here.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import List, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import _lib


class State(ABC):
    """environment_abstract.py:8-15."""

    @abstractmethod
    def __hash__(self):
        pass

    @abstractmethod
    def __eq__(self, other):
        pass


class Environment(ABC):
    """environment_abstract.py:18-163.  Subclasses provide `_env_id`, `_dim`, `state_dim`,
    `_state_cls`, `_get_arr`."""

    def __init__(self):
        self.dtype = float  # reference: np.float (environment_abstract.py:20)
        self.fixed_actions: bool = True

    # ---- array-native core ------------------------------------------------------------------
    def to_device(self, states_np: np.ndarray) -> torch.Tensor:
        dev = _lib.require_gpu()
        return torch.from_numpy(np.ascontiguousarray(states_np, dtype=np.uint8)).to(dev)

    def states_to_np(self, states: List[State]) -> np.ndarray:
        if len(states) == 0:
            return np.zeros((0, self.state_dim), np.uint8)
        return np.stack([self._get_arr(s) for s in states], axis=0).astype(np.uint8, copy=False)

    def np_to_states(self, arr: np.ndarray) -> List[State]:
        return [self._state_cls(x) for x in arr]

    def next_state_dev(self, states: torch.Tensor, action: int) -> torch.Tensor:
        if not (0 <= int(action) < self.get_num_moves()):
            raise IndexError("action %s out of range" % action)
        return _lib.next_state(self._env_id, self._dim, states, action)

    def prev_state_dev(self, states: torch.Tensor, action: int) -> torch.Tensor:
        if not (0 <= int(action) < self.get_num_moves()):
            raise IndexError("action %s out of range" % action)
        return _lib.next_state(self._env_id, self._dim, states, action, prev=True)

    def expand_dev(self, states: torch.Tensor, **kw) -> dict:
        """Fused expansion on device (children / nnet_in / onehot / solved / hash)."""
        return _lib.expand_fused(self._env_id, self._dim, states, **kw)

    def is_solved_dev(self, states: torch.Tensor) -> torch.Tensor:
        return _lib.is_solved(self._env_id, self._dim, states)

    # ---- reference API (lists of State objects) ---------------------------------------------
    def next_state(self, states: List[State], action: int) -> Tuple[List[State], List[float]]:
        """environment_abstract.py:23-31; cube3.py:48-54; n_puzzle.py:46-61."""
        out = self.next_state_dev(self.to_device(self.states_to_np(states)), action).cpu().numpy()
        return self.np_to_states(out), [1.0 for _ in range(len(states))]

    def prev_state(self, states: List[State], action: int) -> List[State]:
        """environment_abstract.py:33-41; moves_rev pairs a with a^1 (cube3.py:28-29, n_puzzle.py:28-29)."""
        out = self.prev_state_dev(self.to_device(self.states_to_np(states)), action).cpu().numpy()
        return self.np_to_states(out)

    @abstractmethod
    def generate_goal_states(self, num_states: int, np_format: bool = False):
        pass

    def is_solved(self, states: List[State]) -> np.ndarray:
        """environment_abstract.py:52-60."""
        if len(states) == 0:
            return np.zeros(0, dtype=bool)
        return self.is_solved_dev(self.to_device(self.states_to_np(states))).cpu().numpy().astype(bool)

    @abstractmethod
    def state_to_nnet_input(self, states: List[State]) -> List[np.ndarray]:
        pass

    @abstractmethod
    def get_num_moves(self) -> int:
        pass

    @abstractmethod
    def get_nnet_model(self) -> nn.Module:
        pass

    def expand(self, states: List[State]) -> Tuple[List[List[State]], List[np.ndarray]]:
        """environment_abstract.py:127-163 / cube3.py:129-161: children of every state + unit costs."""
        n = len(states)
        A = self.get_num_moves()
        if n == 0:
            return [], []
        ch = self.expand_dev(self.to_device(self.states_to_np(states)), children=True, solved=False,
                             hashes=False)["children"].cpu().numpy()
        states_exp = [[self._state_cls(ch[i, a]) for a in range(A)] for i in range(n)]
        tc = np.ones((n, A))
        return states_exp, [tc[i] for i in range(n)]

    def generate_states(self, num_states: int, backwards_range: Tuple[int, int], seed: int = None) -> Tuple[List[State], List[int]]:
        """environment_abstract.py:88-125 (cube3.py:96-127, n_puzzle.py:100-134): start from the goal and take
        `k_i ~ U{lo..hi}` reverse moves.  One launch of the device random-walk kernel (`dca_generate_states`): every
        state takes its own independent uniformly random reverse moves (the reference draws one move per randomly
        chosen subset; the per-state marginal is the same uniform walk)."""
        assert num_states > 0
        assert backwards_range[0] >= 0
        assert self.fixed_actions, "Environments without fixed actions must implement their own method"
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 31 - 1))
        st, nb, _ = _lib.generate_states(self._env_id, self._dim, num_states, backwards_range[0], backwards_range[1], seed)
        return self.np_to_states(st.cpu().numpy()), nb.cpu().numpy().tolist()
