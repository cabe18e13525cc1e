#!/bin/bash
# Round 6, GPU visit C: new / fixed tests, fp8 layer-1 diagnostic + timing, one-hot tile-size A/B, gather kernel with the f32 chunk form.
out=gpurun_out/r06c
mkdir -p $out
export TMPDIR=/tmp
for g in "49 6 128 1000" "54 6 256 1500" "25 25 256 64"; do timeout -s KILL 300 python tools/l1_fp8_check.py $g; done > $out/l1_fp8_check.txt 2>&1
timeout -s KILL 1500 python -m pytest tests/test_gemm8_hip.py tests/test_engine_hip.py tests/test_env_hip.py tests/test_astar_cli_hip.py -m gpu -q --timeout 600 -p no:cacheprovider -k "layer1 or fp8 or device_set_weights or onehot or external_heuristic or golden or expand or config2 or nnet_fp8" > $out/pytest_sel.log 2>&1; echo "sel rc=$?" | tee $out/summary.txt
tail -15 $out/pytest_sel.log | cut -c1-300 | tee -a $out/summary.txt
timeout -s KILL 300 python tools/gemm_bench.py 204800 l1 2>&1 | grep "^{" | tee -a $out/summary.txt
B="python bench.py --steps 20 --warmup 5 --nnet-steps 0 --no-cpu-baseline --concurrent 0 --queue-states 0 --no-expand-block"
for cfg in "14=0" "8=32" "8=64" "8=64,14=2" "14=0"; do
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
for oh in f32 bf16; do timeout -s KILL 300 python bench.py --workload expand --onehot $oh --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$oh', 'value %.4e'%j['value'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'ceiling', r['write_ceiling_GBs'], 'of ceiling', r['frac_of_write_ceiling'])" | tee -a $out/summary.txt; done
cat $out/l1_fp8_check.txt | cut -c1-300
