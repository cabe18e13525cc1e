#!/bin/bash
# diagnostic sweeps of tools/engine_probe.py with dca_debug_tune knobs:  bash tools/whatif.sh <outdir> "<args>" ...
o=$1; shift
mkdir -p "$o"
for a in "$@"; do
  echo "== $a"
  python tools/engine_probe.py cube3 20000 $a 2>/dev/null | python tools/probe_brief.py
done 2>&1 | tee "$o/whatif.txt"
