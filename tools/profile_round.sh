#!/bin/bash
# rocprofv3 records of one round (run on the GPU box from the repo root): per-kernel stats of the default astar bench
# (engine legs), HBM traffic of its kernels from separate FETCH_SIZE / WRITE_SIZE passes (never combined with trace
# domains), kernel stats + MFMA-busy counters of the heuristic network.  Raw output under gpurun_out/<tag>_prof/, the
# summaries to copy into profiles/ under gpurun_out/<tag>_prof/summary/.
tag=${1:-r05}
steps=${2:-20}   # episode shape of the PMC / stats passes: the driver's flags (--steps 20 --warmup 5) by default
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/${tag}_prof
mkdir -p $out/summary
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --nnet-steps 0 --no-cpu-baseline --concurrent 0 --queue-states 0 --no-expand-block --profile-iters 0 --steps $steps --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/astar_stats -o astar -- $BENCH > $out/astar_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/astar_fetch -o astar -- $BENCH --no-onehot-leg > $out/astar_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/astar_write -o astar -- $BENCH --no-onehot-leg > $out/astar_write.log 2>&1
NN="python $R/tools/profile_nnet.py fp32 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/nnet_stats -o nnet -- $NN > $out/nnet_stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/nnet_mfma -o nnet -- $NN > $out/nnet_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/nnet_fetch -o nnet -- $NN > $out/nnet_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/nnet_write -o nnet -- $NN > $out/nnet_write.log 2>&1
for dt in bf16 fp8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/nnet_${dt}_stats -o nnet -- python $R/tools/profile_nnet.py $dt 3 > $out/nnet_${dt}_stats.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/nnet_fp8_mfma -o nnet -- python $R/tools/profile_nnet.py fp8 3 > $out/nnet_fp8_mfma.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/train_stats -o train -- python $R/bench.py --workload train --steps 10 --warmup 3 > $out/train_stats.log 2>&1
cd $R
f=$(find $out/astar_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/summary/${tag}_astar_kernel_stats.csv
f=$(find $out/nnet_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/summary/${tag}_nnet_fp32_kernel_stats.csv
for dt in bf16 fp8; do f=$(find $out/nnet_${dt}_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/summary/${tag}_nnet_${dt}_kernel_stats.csv; done
f=$(find $out/train_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/summary/${tag}_train_kernel_stats.csv
python tools/pmc_summary.py $out/nnet_fp8_mfma > $out/summary/${tag}_nnet_fp8_pmc_mfma.txt
python tools/pmc_traffic.py $out/astar_fetch $out/astar_write cube3 20000 $out/summary/${tag}_astar_pmc_traffic.txt $out/summary/${tag}_pmc_traffic.json "rocprofv3 PMC passes of \`bench.py --steps $steps --warmup 5\`: the timed shape"
python tools/pmc_traffic.py $out/nnet_fetch $out/nnet_write cube3 0 $out/summary/${tag}_nnet_fp32_pmc_traffic.txt /dev/null
python tools/pmc_summary.py $out/nnet_mfma > $out/summary/${tag}_nnet_fp32_pmc_mfma.txt
ls -la $out/summary
