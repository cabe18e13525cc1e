#!/bin/bash
# Round 6, GPU visit ZA: HBM traffic of dca_l1_embed (rocprofv3 FETCH_SIZE / WRITE_SIZE in separate passes) on tools/l1_embed_bench.py 40960.
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r06za
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o e -- python $R/tools/l1_embed_bench.py 40960 > $out/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o e -- python $R/tools/l1_embed_bench.py 40960 > $out/write.log 2>&1
cd $R
python tools/pmc_traffic.py $out/fetch $out/write puzzle 40960 $out/r06_l1_embed_pmc_traffic.txt /dev/null "tools/l1_embed_bench.py 40960"
rm -rf $out/fetch $out/write
grep "k_l1_embed" $out/r06_l1_embed_pmc_traffic.txt | cut -c1-140
