#!/bin/bash
# Round 6, GPU visit R: records for the final dca_l1_embed — PMC passes (tools/gpu_r06k.sh), the vector-instruction probe with long
# loops, the tolerance tests with their printed deviations.
out=gpurun_out/r06r
mkdir -p $out
export TMPDIR=/tmp
bash tools/gpu_r06k.sh > $out/k.log 2>&1
cp gpurun_out/r06k/pmc_summary.txt $out/l1_embed_pmc.txt
timeout 200 tools/bin/valu_rate_probe | tee $out/valu_rate_probe.txt
timeout -s KILL 900 python -m pytest tests/test_parity_configs_hip.py -q -s -k "tolerance or puzzle_network" 2>&1 | grep -E "seed 20|puzzle48 \|h\||deviation|passed|failed" | cut -c1-700 | tee $out/tolerance.txt
