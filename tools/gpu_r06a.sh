#!/bin/bash
# Round 6, GPU visit A: smoke + GPU test suite, bench at the driver's flags, layer-1 / gemm16 probes, one-hot overlap A/B.
out=gpurun_out/r06a
mkdir -p $out
export TMPDIR=/tmp
bash tools/gpu_round.sh r06a nobench
timeout -s KILL 900 python bench.py --steps 20 --warmup 5 > $out/bench_contract.json 2> $out/bench_contract.err; echo "bench rc=$?" | tee -a $out/summary.txt
timeout -s KILL 300 python tools/gemm_bench.py 204800 l1 > $out/gemm_l1.txt 2>&1
timeout -s KILL 600 python tools/gemm16_probe.py 204800 3,4,5 1024,5120 > $out/gemm16_probe.txt 2>&1
for mode in 0 1 2; do
  timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --nnet-steps 0 --no-cpu-baseline --concurrent 0 --queue-states 0 --no-expand-block --tune 14=$mode > $out/onehot_mode$mode.json 2> $out/onehot_mode$mode.err
done
tail -c 1500 $out/gemm_l1.txt; tail -c 3000 $out/gemm16_probe.txt
python - <<'PY'
import json
for m in (0,1,2):
    try:
        j=json.loads(open('gpurun_out/r06a/onehot_mode%d.json'%m).read().strip().splitlines()[-1])
        o=j["engine_onehot_f32"]; print("mode",m,"value %.3e"%j["value"],"onehot %.3e"%o["value"], o.get("roofline_expand",{}).get("kernel_ms"), o.get("roofline_expand",{}).get("frac"))
    except Exception as e: print("mode",m,"failed",e)
PY
