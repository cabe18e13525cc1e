#!/bin/bash
# Round 6, GPU visit K: what bounds the embedding-sum kernel — PMC counters (LDS array cycles, bank conflicts, VALU busy) of the
# per-geometry bench, (tests of the alignment fix ran in the first attempt of this visit: 30 passed).
out=gpurun_out/r06k
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -s KILL 300 rocprofv3 --pmc $set -d $GRAFT_REPO_ROOT/$out/pmc_$tag -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/l1_embed_bench.py 40960 > $GRAFT_REPO_ROOT/$out/pmc_$tag.log 2>&1
  echo "pmc $set rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out 2>&1 | grep -E "embed|counter|kernel" | head -80 | tee $out/pmc_summary.txt
