#!/bin/bash
# Round 6, GPU visit Z: rocprofv3 kernel stats of the puzzle48 network (fp32 parity mode and bf16; layer 1 = k_l1_embed), 204 800 rows x 3 forwards.
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r06z
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for m in fp32 bf16; do
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$m -o nnet -- python $R/tools/profile_nnet.py $m 3 puzzle48 > $out/st_$m.log 2>&1
  f=$(find $out/st_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/r06_nnet_puzzle48_${m}_kernel_stats.csv
  rm -rf $out/st_$m
  head -8 $out/r06_nnet_puzzle48_${m}_kernel_stats.csv | cut -c1-170
done
