"""Cube3 — mirror of the reference `environments/cube3.py` behind the same Environment API, with every
batched operation running on the MI355X (libdca_hip.so).  Move tables come from the library
(`dca_cube3_perm_table`), not from Python."""
from __future__ import annotations

from typing import List, Union

import numpy as np
from torch import nn

from .. import _lib
from .environment_abstract import Environment, State


class Cube3State(State):
    """cube3.py:10-24: key = the raw sticker bytes."""
    __slots__ = ['colors', 'hash']

    def __init__(self, colors: np.ndarray):
        self.colors: np.ndarray = colors
        self.hash = None

    def __hash__(self):
        if self.hash is None:
            self.hash = hash(np.asarray(self.colors, dtype=np.uint8).tobytes())
        return self.hash

    def __eq__(self, other):
        return np.array_equal(self.colors, other.colors)

    def __setstate__(self, state):
        # reference pickles (data/cube3/test/data_0.pkl) store int64 colors: normalise to uint8
        slots = state[1] if isinstance(state, tuple) else state
        self.colors = np.asarray(slots['colors']).astype(np.uint8)
        self.hash = None


class Cube3(Environment):
    moves: List[str] = ["%s%i" % (f, n) for f in ['U', 'D', 'L', 'R', 'B', 'F'] for n in [-1, 1]]
    moves_rev: List[str] = ["%s%i" % (f, n) for f in ['U', 'D', 'L', 'R', 'B', 'F'] for n in [1, -1]]

    _env_id = _lib.ENV_CUBE3
    _dim = 0
    state_dim = 54
    env_name = "cube3"  # the registry name (utils/env_utils.py), what the engine is created with
    _state_cls = Cube3State

    def __init__(self):
        super().__init__()
        self.dtype = np.uint8
        self.cube_len = 3
        self.goal_colors: np.ndarray = np.arange(0, 54, 1, dtype=np.uint8)

    @staticmethod
    def _get_arr(state: Cube3State) -> np.ndarray:
        return state.colors

    def generate_goal_states(self, num_states: int, np_format: bool = False) -> Union[List[Cube3State], np.ndarray]:
        """cube3.py:62-69."""
        if np_format:
            return np.repeat(self.goal_colors[None].copy(), num_states, axis=0)
        return [Cube3State(self.goal_colors.copy()) for _ in range(num_states)]

    def state_to_nnet_input(self, states: List[Cube3State]) -> List[np.ndarray]:
        """cube3.py:77-85: colour index = sticker // 9."""
        if len(states) == 0:
            return [np.zeros((0, 54), np.uint8)]
        dev = self.to_device(self.states_to_np(states))
        return [_lib.nnet_input(self._env_id, self._dim, dev).cpu().numpy()]

    def get_num_moves(self) -> int:
        return len(self.moves)

    def get_nnet_model(self) -> nn.Module:
        """cube3.py:90-94."""
        from ..utils.pytorch_models import ResnetModel
        return ResnetModel(54, 6, 5000, 1000, 4, 1, True)
