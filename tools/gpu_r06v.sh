#!/bin/bash
# Round 6, GPU visit V: update -> train -> search on CUBE3 inside ONE GPU-box visit (gpurun caps a visit at 3600 s; visit T ran into
# that cap 60 minutes into training and lost its network): 2800 s of avi.py (10 M states per update, 3 epochs = 3000 Adam steps per
# update, fixed seeds), export of the network, then BWAS (w 0.6, batch 10 000: train.sh:9) on the first 100 shipped test states, in
# chunks of 20, until the deadline.
out=gpurun_out/r06v
mkdir -p $out
export TMPDIR=/tmp
DCA_E2E_MAX_NODES=130000000 DCA_E2E_DEADLINE=3440 DCA_E2E_EXPORT_FP32=1 DCA_E2E_EXPORT=$out/cube3_avi.pt \
  timeout -s KILL 3540 python tools/avi_e2e.py 2800 100 10000000 - 3 cube3 > $out/avi_e2e_cube3.log 2>&1
echo "rc=$?" >> $out/avi_e2e_cube3.log
grep -v "^Itr: " $out/avi_e2e_cube3.log | grep -v "Back Steps: \([1-9]\|1[0-9]\|2[1-9]\)," | tail -60 | cut -c1-220
