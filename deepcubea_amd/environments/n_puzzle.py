"""NPuzzle — mirror of the reference `environments/n_puzzle.py` ((n^2-1)-puzzle, n = 4..7) behind the
same Environment API; batched operations run on the MI355X (libdca_hip.so)."""
from __future__ import annotations

from typing import List, Union

import numpy as np
import torch.nn as nn

from .. import _lib
from .environment_abstract import Environment, State


class NPuzzleState(State):
    """n_puzzle.py:10-24."""
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


class NPuzzle(Environment):
    moves: List[str] = ['U', 'D', 'L', 'R']
    moves_rev: List[str] = ['D', 'U', 'R', 'L']

    _env_id = _lib.ENV_NPUZZLE
    _state_cls = NPuzzleState

    def __init__(self, dim: int):
        super().__init__()
        if not (4 <= dim <= 7):
            raise ValueError("NPuzzle dim must be 4..7 (puzzle15/24/35/48), got %d" % dim)
        self.dim: int = dim
        self._dim = dim
        self.state_dim = dim * dim
        self.env_name = "puzzle%d" % (dim * dim - 1)  # the registry name (utils/env_utils.py)
        self.dtype = np.uint8  # reference: uint8 for dim<=15 (n_puzzle.py:35-38); tiles < 49 always fit
        self.goal_tiles: np.ndarray = np.concatenate((np.arange(1, dim * dim), [0])).astype(np.uint8)

    @staticmethod
    def _get_arr(state: NPuzzleState) -> np.ndarray:
        return state.tiles

    def generate_goal_states(self, num_states: int, np_format: bool = False) -> Union[List[NPuzzleState], np.ndarray]:
        """n_puzzle.py:69-76."""
        if np_format:
            return np.repeat(self.goal_tiles[None].copy(), num_states, axis=0)
        return [NPuzzleState(self.goal_tiles.copy()) for _ in range(num_states)]

    def state_to_nnet_input(self, states: List[NPuzzleState]) -> List[np.ndarray]:
        """n_puzzle.py:84-89: the tiles themselves."""
        return [self.states_to_np(states)]

    def get_num_moves(self) -> int:
        return len(self.moves)

    def get_nnet_model(self) -> nn.Module:
        """n_puzzle.py:94-98."""
        from ..utils.pytorch_models import ResnetModel
        return ResnetModel(self.state_dim, self.dim ** 2, 5000, 1000, 4, 1, True)
