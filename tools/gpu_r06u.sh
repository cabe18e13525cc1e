#!/bin/bash
# Round 6, GPU visit U (the switches below existed only for this visit; what won is the default now, the rest was removed):
# dca_l1_embed experiments — row addresses by SDWA byte moves (NT = 64 geometries) and puzzle24 on 64-column
# tiles (12 waves) — tests with the switches on, then the per-geometry timing three ways.
out=gpurun_out/r06u
mkdir -p $out
DCA_EMBED_SDWA=1 DCA_EMBED_P24_NT64=1 timeout -s KILL 600 python -m pytest tests/test_embed_hip.py -x -q 2>&1 | tail -5 | tee $out/tests_sdwa.txt
echo "== default" | tee $out/l1_embed_bench.txt
timeout -s KILL 300 python tools/l1_embed_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $out/l1_embed_bench.txt
echo "== DCA_EMBED_P24_NT64" | tee -a $out/l1_embed_bench.txt
DCA_EMBED_P24_NT64=1 timeout -s KILL 300 python tools/l1_embed_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $out/l1_embed_bench.txt
echo "== DCA_EMBED_SDWA DCA_EMBED_P24_NT64" | tee -a $out/l1_embed_bench.txt
DCA_EMBED_SDWA=1 DCA_EMBED_P24_NT64=1 timeout -s KILL 300 python tools/l1_embed_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $out/l1_embed_bench.txt
for v in 1 2 3 4; do
  echo "== DCA_EMBED_PG=$v (position groups: 1 = 8 waves x 32 states, 2 = 12 x 20, 3 = 16 x 12, 4 = 16 x 10)" | tee -a $out/l1_embed_bench.txt
  DCA_EMBED_PG=$v timeout -s KILL 300 python -m pytest tests/test_embed_hip.py -x -q -k "large_batches" 2>&1 | tail -2 | tee -a $out/tests_sdwa.txt
  DCA_EMBED_PG=$v timeout -s KILL 300 python tools/l1_embed_bench.py 2>&1 | grep "puzzle35\|puzzle48" | tee -a $out/l1_embed_bench.txt
done
