#!/bin/bash
# Round 6, GPU visit Y: the reference's second training stage on cube3 (train.sh:5-6: copy current -> target by hand, train on) with
# the network of visit V: 1500 s more of avi.py from iteration 270 000 with the imported network as the new target (hand-over 15),
# then BWAS (w 0.6, batch 10 000) on the first 100 shipped test states again.
out=gpurun_out/r06y
mkdir -p $out
export TMPDIR=/tmp
DCA_E2E_IMPORT=tools/bin/cube3_avi.pt DCA_E2E_MAX_NODES=130000000 DCA_E2E_CHUNK=50 DCA_E2E_DEADLINE=1750 DCA_E2E_EXPORT_FP32=1 DCA_E2E_EXPORT=$out/cube3_avi_stage2.pt \
  timeout -s KILL 1850 python tools/avi_e2e.py 1500 100 10000000 - 3 cube3 > $out/avi_e2e_cube3_stage2.log 2>&1
echo "rc=$?" >> $out/avi_e2e_cube3_stage2.log
grep -v "^Itr: \|^State: " $out/avi_e2e_cube3_stage2.log | grep -v "Back Steps: \([1-9]\|1[0-9]\|2[1-9]\)," | tail -40 | cut -c1-200
