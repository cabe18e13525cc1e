"""`utils/search_utils.py:7-13` — the solution check every BWAS driver asserts (astar.py:443,556)."""
from typing import List


def is_valid_soln(state, soln: List[int], env) -> bool:
    cur = state
    for move in soln:
        cur = env.next_state([cur], move)[0][0]
    return bool(env.is_solved([cur])[0])
