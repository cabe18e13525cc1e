"""Environment registry of the path (names as in the reference's `utils/env_utils.py:6-28`): `cube3` and
`puzzle<N>` for N in {15, 24, 35, 48}, and `lightsout7` (the remaining environment of the reference's C++ core
that its Python harness can drive).  Sokoban is outside the hot-path scope and not built."""
import re

_PUZZLE_DIMS = {15: 4, 24: 5, 35: 6, 48: 7}


def get_environment(env_name: str):
    key = env_name.strip().lower()
    if key == "cube3":
        from ..environments.cube3 import Cube3
        return Cube3()
    found = re.fullmatch(r".*puzzle(\d+).*", key)
    if found and int(found.group(1)) in _PUZZLE_DIMS:
        from ..environments.n_puzzle import NPuzzle
        return NPuzzle(_PUZZLE_DIMS[int(found.group(1))])
    found = re.fullmatch(r".*lightsout(\d+).*", key)
    if found:
        from ..environments.lights_out import LightsOut
        return LightsOut(int(found.group(1)))
    raise ValueError('No known environment %s' % env_name)
