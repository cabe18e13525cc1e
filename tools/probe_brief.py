#!/usr/bin/env python3
"""stdin: engine_probe.py rows -> one short line per row (launch spans without the k_rank phase marks)."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    s = d["span_us"]
    print(d["it"], d["front_n"], d["max_bin"], {k: v for k, v in s.items() if not k.startswith("rank_b")}, flush=True)
