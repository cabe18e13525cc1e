#!/bin/bash
# Round 6, GPU visit F: everything after the move to the 16x16x32 matrix instruction (gemm16 + f16x3): full GPU suite, bench, gemm bench.
out=gpurun_out/r06f
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 600 python tools/gemm_bench.py 204800 f16x3 2>&1 | grep "^{" | tee $out/gemm_bench_f16x3.txt | cut -c1-700
bash tools/gpu_round.sh r06f nobench
timeout -s KILL 900 python bench.py --steps 20 --warmup 5 > $out/bench_contract.json 2> $out/bench_contract.err; echo "bench rc=$?" | tee -a $out/summary.txt
python - <<'PY' | tee -a $out/summary.txt
import json
j=json.loads(open('gpurun_out/r06f/bench_contract.json').read().strip().splitlines()[-1])
print("value %.4e ms %.5f"%(j["value"], j["ms_per_step"]), "onehot %.4e" % j["engine_onehot_f32"]["value"])
for k,v in j["end_to_end_nnet"].items(): print(k, "%.4e"%v["value"], "%.3f ms" % v["ms_per_step"])
for k,v in j["expand_1M"].items(): print(k, "%.4e" % v["value"], v["ms_per_launch"], v["roofline"]["frac"], v["roofline"]["write_ceiling_GBs"])
PY
