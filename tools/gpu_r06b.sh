#!/bin/bash
# Round 6, GPU visit B: the tests that failed in visit A (fixed), one-hot overlap A/B (paced), rocprofv3 records, gemm bench.
out=gpurun_out/r06b
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 1200 python -m pytest tests/test_gemm8_hip.py tests/test_parity_configs_hip.py tests/test_engine_hip.py tests/test_astar_cli_hip.py -m gpu -q -s --timeout 600 -p no:cacheprovider > $out/pytest_fixed.log 2>&1; echo "fixed rc=$?" | tee $out/summary.txt
grep -E "seed 20|puzzle48 \|h\||deviation / max|passed|failed" $out/pytest_fixed.log | cut -c1-600 | tee -a $out/summary.txt
B="python bench.py --steps 20 --warmup 5 --nnet-steps 0 --no-cpu-baseline --concurrent 0 --queue-states 0 --no-expand-block"
for cfg in "14=0" "14=2,15=1" "14=2,15=2" "14=2,15=4" "14=2,15=8" "14=0"; do
  timeout -s KILL 300 $B --tune $cfg > $out/onehot_$cfg.json 2> $out/onehot_$cfg.err
  python - "$out/onehot_$cfg.json" "$cfg" <<'PY' | tee -a $out/summary.txt
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); o = j["engine_onehot_f32"]
    print("knobs", sys.argv[2], "value %.4e" % j["value"], "onehot %.4e" % o["value"], "k_expand_oh ms", o.get("roofline_expand", {}).get("kernel_ms"), "frac", o.get("roofline_expand", {}).get("frac"))
except Exception as e:
    print("knobs", sys.argv[2], "failed", e)
PY
done
timeout -s KILL 900 python tools/gemm_bench.py 204800 > $out/gemm_bench.txt 2>&1
bash tools/profile_expand.sh r06 > $out/profile_expand.log 2>&1
bash tools/profile_round.sh r06 20 > $out/profile_round.log 2>&1
tail -5 $out/gemm_bench.txt | cut -c1-700
ls gpurun_out/r06_prof/summary gpurun_out/r06_expand_prof/summary
