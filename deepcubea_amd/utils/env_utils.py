"""Environment registry for the path (reference `utils/env_utils.py:6-28`): cube3 and puzzle15/24/35/48.
Environments outside the hot-path scope (lightsout, sokoban) are not provided."""
import math
import re


def get_environment(env_name: str):
    name = env_name.lower()
    m = re.search(r"puzzle(\d+)", name)
    if name == 'cube3':
        from ..environments.cube3 import Cube3
        return Cube3()
    if m is not None:
        from ..environments.n_puzzle import NPuzzle
        return NPuzzle(int(math.sqrt(int(m.group(1)) + 1)))
    raise ValueError('No known environment %s' % env_name)
