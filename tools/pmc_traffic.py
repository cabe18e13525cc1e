#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — they do not fit one pass on gfx950):
per-kernel mean over the dispatches, FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads,
MI355X_MICROARCH.md §HBM).  python tools/pmc_traffic.py FETCH_DIR WRITE_DIR env batch out.txt out.json"""
import collections
import csv
import glob
import json
import re
import sys

fd, wd, env, batch, out_txt, out_json = sys.argv[1:7]
shape = sys.argv[7] if len(sys.argv) > 7 else "rocprofv3 PMC passes of bench.py at 100-step episodes"


def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
rows = []
for k in sorted(set(fe) | set(wr)):
    f = sum(fe[k]) / len(fe[k]) if fe.get(k) else 0.0
    w = sum(wr[k]) / len(wr[k]) if wr.get(k) else 0.0
    rows.append((k, max(len(fe.get(k, [])), len(wr.get(k, []))), f, w, (2 * f + w) * 1024))
rows.sort(key=lambda r: -r[4] * r[1])
with open(out_txt, "w") as o:
    o.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); KB per dispatch averaged over all dispatches;\n"
            "# HBM_bytes = (2 * FETCH + WRITE) * 1024 (gfx950: FETCH_SIZE counts half the bytes of wide reads)\n")
    o.write("%-70s %6s %14s %14s %16s\n" % ("kernel", "calls", "FETCH_KB(raw)", "WRITE_KB", "HBM_bytes"))
    for k, n, f, w, b in rows:
        o.write("%-70s %6d %14.1f %14.1f %16.0f\n" % (k[:70], n, f, w, b))
kern = {}
for k, n, f, w, b in rows:
    m = re.search(r"(k_[a-z_0-9]+|expand_fused_kernel)", k)
    if m and m.group(1) not in kern:
        kern[m.group(1)] = b
json.dump({"env": env, "batch_size": int(batch), "kernels": kern, "shape": shape,
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `python bench.py` (tools/profile_round.sh)"},
          open(out_json, "w"), indent=1)
