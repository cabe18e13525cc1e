#!/bin/bash
# the heuristic-network half of tools/profile_round.sh on its own (kernel stats, MFMA-busy counters, HBM traffic of the
# fp32 parity mode: layer-1 one-hot MFMA kernel + f16x3 GEMMs).  Summaries under gpurun_out/<tag>_prof/summary/.
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/${tag}_prof
mkdir -p $out/summary
cd /tmp; export TMPDIR=/tmp
NN="python $R/tools/profile_nnet.py fp32 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/nnet_stats -o nnet -- $NN > $out/nnet_stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/nnet_mfma -o nnet -- $NN > $out/nnet_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/nnet_fetch -o nnet -- $NN > $out/nnet_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/nnet_write -o nnet -- $NN > $out/nnet_write.log 2>&1
cd $R
f=$(find $out/nnet_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/summary/${tag}_nnet_fp32_kernel_stats.csv
python tools/pmc_traffic.py $out/nnet_fetch $out/nnet_write cube3 0 $out/summary/${tag}_nnet_fp32_pmc_traffic.txt /dev/null
python tools/pmc_summary.py $out/nnet_mfma > $out/summary/${tag}_nnet_fp32_pmc_mfma.txt
rm -rf $out/nnet_stats $out/nnet_mfma $out/nnet_fetch $out/nnet_write
cat $out/summary/${tag}_nnet_fp32_kernel_stats.csv | cut -c1-180 | head -n 12
cat $out/summary/${tag}_nnet_fp32_pmc_mfma.txt $out/summary/${tag}_nnet_fp32_pmc_traffic.txt | cut -c1-200
