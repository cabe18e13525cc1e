#!/bin/bash
# Round 6, GPU visit H: the record on the final build — smoke + full GPU suite, bench at the driver's flags and at the defaults (timed).
out=gpurun_out/r06h
mkdir -p $out
export TMPDIR=/tmp
bash tools/gpu_round.sh r06h nobench
t0=$(date +%s); timeout -s KILL 900 python bench.py --steps 20 --warmup 5 > $out/bench_contract.json 2> $out/bench_contract.err; echo "bench contract rc=$? $(( $(date +%s) - t0 )) s" | tee -a $out/summary.txt
t0=$(date +%s); timeout -s KILL 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$? $(( $(date +%s) - t0 )) s" | tee -a $out/summary.txt
python - <<'PY' | tee -a $out/summary.txt
import json
for f in ("bench_contract", "bench_default"):
    j=json.loads(open('gpurun_out/r06h/%s.json' % f).read().strip().splitlines()[-1])
    print(f, "value %.4e ms %.5f"%(j["value"], j["ms_per_step"]), "onehot %.4e" % j["engine_onehot_f32"]["value"], "K16 %.4e" % j["concurrent_instances"]["value"])
    print("  nnet", {k: "%.3e" % v["value"] for k,v in j["end_to_end_nnet"].items()})
    print("  expand", {k: (round(v["roofline"]["frac"],3), round(v["roofline"]["frac_of_write_ceiling"],3)) for k,v in j["expand_1M"].items()})
PY
