#!/bin/bash
# Round 6, GPU visit T: update -> train -> search on CUBE3 (the north star's own configuration: train.sh:4,9) inside one GPU-box visit:
# 95 minutes of avi.py (10 M states per update, 3 epochs = 3000 Adam steps per update, fixed seeds), then BWAS (w 0.6, batch 10 000)
# on the first 100 shipped test states against the shipped optimal lengths and the published results.
out=gpurun_out/r06t
mkdir -p $out
export TMPDIR=/tmp
DCA_E2E_MAX_NODES=100000000 DCA_E2E_EXPORT=$out/cube3_avi_fp16.pt timeout -s KILL 7500 python tools/avi_e2e.py 5700 100 10000000 - 3 cube3 > $out/avi_e2e_cube3.log 2>&1
echo "rc=$?" >> $out/avi_e2e_cube3.log
grep -v "^Itr: " $out/avi_e2e_cube3.log | grep -v "Back Steps: \([1-9]\|1[0-9]\|2[1-9]\)," | tail -150 | cut -c1-220
