"""LightsOut — mirror of the reference `environments/lights_out.py` (dim x dim board, press a cell: it and its in-board
4-neighbours flip) behind the same Environment API; batched operations run on the MI355X (libdca_hip.so).  The
reference's C++ core (`cpp/environments.cpp:133-208`, `parallel_weighted_astar.cpp:388-389`) and `train.sh:65-71` use
dim 7 ("lightsout7"): that is the size instantiated on the device."""
from __future__ import annotations

from typing import List, Union

import numpy as np
import torch.nn as nn

from .. import _lib
from .environment_abstract import Environment, State


class LOState(State):
    """lights_out.py:9-24."""
    __slots__ = ['tiles', 'hash']

    def __init__(self, tiles: np.ndarray):
        self.tiles: np.ndarray = tiles
        self.hash = None

    def __hash__(self):
        if self.hash is None:
            self.hash = hash(np.asarray(self.tiles, dtype=np.uint8).tobytes())
        return self.hash

    def __eq__(self, other):
        return np.array_equal(self.tiles, other.tiles)

    def __setstate__(self, state):
        slots = state[1] if isinstance(state, tuple) else state
        self.tiles = np.asarray(slots['tiles']).astype(np.uint8)
        self.hash = None


class LightsOut(Environment):
    _env_id = _lib.ENV_LIGHTSOUT
    _state_cls = LOState

    def __init__(self, dim: int):
        super().__init__()
        if dim != 7:
            raise ValueError("LightsOut is built for dim 7 (lightsout7: the size of the reference's C++ core), got %d" % dim)
        self.dtype = np.uint8  # lights_out.py:29
        self.dim: int = dim
        self._dim = dim
        self.env_name = "lightsout%d" % dim  # the registry name (utils/env_utils.py)
        self.num_tiles: int = dim ** 2
        self.state_dim = self.num_tiles
        # lights_out.py:33-44 — kept for callers that read it; the device kernels compute the same mask arithmetically
        self.move_matrix = np.zeros((self.num_tiles, 5), dtype=np.int64)
        for move in range(self.num_tiles):
            x_pos, y_pos = move // dim, move % dim
            right = move + dim if x_pos < (dim - 1) else move
            left = move - dim if x_pos > 0 else move
            up = move + 1 if y_pos < (dim - 1) else move
            down = move - 1 if y_pos > 0 else move
            self.move_matrix[move] = [move, right, left, up, down]

    @staticmethod
    def _get_arr(state: LOState) -> np.ndarray:
        return state.tiles

    def generate_goal_states(self, num_states: int, np_format: bool = False) -> Union[List[LOState], np.ndarray]:
        """lights_out.py:55-63: all lights off."""
        if np_format:
            return np.zeros((num_states, self.num_tiles), dtype=self.dtype)
        return [LOState(np.zeros(self.num_tiles, dtype=self.dtype)) for _ in range(num_states)]

    def state_to_nnet_input(self, states: List[LOState]) -> List[np.ndarray]:
        """lights_out.py:70-75: the cells themselves."""
        return [self.states_to_np(states)]

    def get_num_moves(self) -> int:
        return self.num_tiles

    def get_nnet_model(self) -> nn.Module:
        """lights_out.py:80-83."""
        from ..utils.pytorch_models import ResnetModel
        return ResnetModel(self.num_tiles, 6, 5000, 1000, 4, 1, True)
