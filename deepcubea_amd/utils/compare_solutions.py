"""Parity report between two BWAS result sets — same statistics and wording as the reference's
`scripts/compare_solutions.py:16-55` (times, lengths, nodes generated, nodes/sec; length differences and % equal),
plus a reader for the published per-state log lines of `results/*/output.txt` (astar.py:449-452 format), because
`results/cube3/results.pkl` itself is not redistributed.

    python -m deepcubea_amd.utils.compare_solutions --soln1 data/cube3/test/data_0.pkl --soln2 results/cube3/results.pkl
    python -m deepcubea_amd.utils.compare_solutions --soln1 results/cube3/output.txt --soln2 my_results/results.pkl
"""
from __future__ import annotations

import re
from argparse import ArgumentParser
from typing import Dict, List

import numpy as np

from . import data_utils

_LINE = re.compile(r"State: (\d+), SolnCost: ([\d.]+), # Moves: (\d+), # Nodes Gen: ([\d,]+), Time: ([\d.]+)")


def parse_output_txt(path: str) -> Dict[str, np.ndarray]:
    """Per-state (length, nodes generated, time) from a BWAS log (astar.py:449-452 / 559-562 line format)."""
    lens, nodes, times = [], [], []
    for line in open(path):
        m = _LINE.search(line)
        if m:
            lens.append(int(m.group(3)))
            nodes.append(int(m.group(4).replace(",", "")))
            times.append(float(m.group(5)))
    return {"lens": np.array(lens), "num_nodes_generated": np.array(nodes, np.float64), "times": np.array(times)}


def load_results(path: str) -> Dict[str, np.ndarray]:
    if path.endswith(".txt"):
        return parse_output_txt(path)
    r = data_utils.load_pickle(path)
    return {"lens": np.array([len(x) for x in r["solutions"]]), "times": np.array(r["times"], np.float64),
            "num_nodes_generated": np.array(r["num_nodes_generated"], np.float64)}


def stats(data) -> Dict[str, float]:
    d = np.asarray(data, np.float64)
    return {"min": float(d.min()), "max": float(d.max()), "median": float(np.median(d)), "mean": float(d.mean()),
            "std": float(d.std())}


def summarize(res: Dict[str, np.ndarray]) -> Dict[str, Dict[str, float]]:
    with np.errstate(divide="ignore", invalid="ignore"):
        nps = res["num_nodes_generated"] / res["times"]
    return {"Times": stats(res["times"]), "Lengths": stats(res["lens"]),
            "Nodes Generated": stats(res["num_nodes_generated"]), "Nodes/Sec": stats(nps[np.isfinite(nps)])}


def compare(res1, res2, offset: int = 0) -> Dict[str, object]:
    """offset: state index of soln1 that soln2's first entry corresponds to (a `--start_idx` run).  Result sets of
    different length without an explicit offset are refused — the reference script fails on the shape mismatch too,
    and silently comparing misaligned states would be worse."""
    n1, n2 = len(res1["lens"]), len(res2["lens"])
    if offset == 0 and n1 != n2:
        raise ValueError("result sets differ in length (%d vs %d states): pass --offset for a --start_idx run" % (n1, n2))
    if offset:
        if offset < 0 or offset + n2 > n1:
            raise ValueError("offset %d + %d states does not fit the %d states of soln1" % (offset, n2, n1))
        res1 = {k: v[offset:offset + n2] for k, v in res1.items()}
    n = min(len(res1["lens"]), len(res2["lens"]))
    diff = res2["lens"][:n] - res1["lens"][:n]
    return {"num_states": n, "soln1": summarize(res1), "soln2": summarize(res2), "length_diff": stats(diff),
            "pct_equal": float(100.0 * np.mean(diff == 0))}


def _fmt(s: Dict[str, float]) -> str:
    return "Min/Max/Median/Mean(Std) %f/%f/%f/%f(%f)" % (s["min"], s["max"], s["median"], s["mean"], s["std"])


def format_report(cmp: Dict[str, object]) -> str:
    out: List[str] = ["%i states" % cmp["num_states"]]
    for title, key in (("\n--SOLUTION 1---", "soln1"), ("\n--SOLUTION 2---", "soln2")):
        out.append(title)
        for name in ("Times", "Lengths", "Nodes Generated", "Nodes/Sec"):
            out.append("-%s-" % name)
            out.append(_fmt(cmp[key][name]))
    out.append("\n\n------Solution 2 - Solution 1 Lengths-----")
    out.append(_fmt(cmp["length_diff"]))
    out.append("%.2f%% soln2 equal to soln1" % cmp["pct_equal"])
    return "\n".join(out)


def main(argv=None):
    parser = ArgumentParser()
    parser.add_argument('--soln1', type=str, required=True, help="results.pkl / data_0.pkl / output.txt")
    parser.add_argument('--soln2', type=str, required=True, help="results.pkl / output.txt")
    parser.add_argument('--offset', type=int, default=0, help="soln2's first state is soln1's state number OFFSET")
    args = parser.parse_args(argv)
    print(format_report(compare(load_results(args.soln1), load_results(args.soln2), args.offset)))


if __name__ == "__main__":
    main()
