#!/usr/bin/env python3
"""Golden fixture of the reference's command-line surfaces, made by IMPORTING the reference (build container only):

    python tests/golden/make_golden_cli.py      -> tests/golden/cli_flags.json

For `search_methods/astar.py` and `ctg_approx/avi.py` the script intercepts `ArgumentParser.parse_args` while the
reference's own `main()` / `parse_arguments()` builds its parser and records every option: flag strings, default, type,
required, action.  Data only (no reference source)."""
import json
import os
import sys
from argparse import ArgumentParser

import numpy as np

np.float = float  # noqa
np.int = int  # noqa
sys.path.insert(0, "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


class _Captured(Exception):
    pass


def capture(fn):
    rec = {}
    orig = ArgumentParser.parse_args

    def fake(self, *a, **k):
        for act in self._actions:
            if not act.option_strings or act.dest == "help":
                continue
            rec[act.dest] = {"flags": sorted(act.option_strings), "default": act.default, "required": bool(act.required),
                             "type": getattr(act.type, "__name__", None), "action": type(act).__name__}
        raise _Captured()

    ArgumentParser.parse_args = fake
    try:
        fn()
    except _Captured:
        pass
    finally:
        ArgumentParser.parse_args = orig
    return rec


from search_methods import astar as ref_astar  # noqa: E402
from ctg_approx import avi as ref_avi  # noqa: E402

out = {"astar": capture(ref_astar.main), "avi": capture(lambda: ref_avi.parse_arguments(ArgumentParser()))}
json.dump(out, open(os.path.join(OUT, "cli_flags.json"), "w"), indent=1, sort_keys=True)
print({k: len(v) for k, v in out.items()})
