#!/bin/bash
# Round 6, GPU visit W: the record on the final build — smoke + full GPU suite, bench at the driver's flags and at the defaults, the
# layer-1 / network probes and the puzzle benches, PMC passes of dca_l1_embed, then BWAS on ALL 1000 shipped cube3 test states with the
# network GPU visit V trained (tools/bin/cube3_avi.pt, search only).
out=gpurun_out/r06w
mkdir -p $out
export TMPDIR=/tmp
bash tools/gpu_round.sh r06w nobench
t0=$(date +%s); timeout -s KILL 900 python bench.py --steps 20 --warmup 5 > $out/bench_contract.json 2> $out/bench_contract.err; echo "bench contract rc=$? $(( $(date +%s) - t0 )) s" | tee -a $out/summary.txt
t0=$(date +%s); timeout -s KILL 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$? $(( $(date +%s) - t0 )) s" | tee -a $out/summary.txt
python - <<'PY' | tee -a $out/summary.txt
import json
for f in ("bench_contract", "bench_default"):
    j=json.loads(open('gpurun_out/r06w/%s.json' % f).read().strip().splitlines()[-1])
    print(f, "value %.4e ms %.5f"%(j["value"], j["ms_per_step"]), "onehot %.4e" % j["engine_onehot_f32"]["value"], "K16 %.4e" % j["concurrent_instances"]["value"])
    print("  nnet", {k: "%.3e" % v["value"] for k,v in j["end_to_end_nnet"].items()})
    print("  expand", {k: (round(v["roofline"]["frac"],3), round(v["roofline"]["frac_of_write_ceiling"],3)) for k,v in j["expand_1M"].items()})
PY
timeout -s KILL 300 python tools/l1_embed_bench.py 2>&1 | grep -v amdgpu.ids | tee $out/l1_embed_bench.txt
timeout -s KILL 600 python tools/nnet_forward_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/nnet_forward_probe.txt
for args in "--env puzzle48" "--workload avi" "--workload avi --nnet_dtype bf16" "--workload avi --env puzzle48" "--workload avi --env puzzle48 --nnet_dtype bf16" "--workload avi --env puzzle15" "--workload train --steps 20 --warmup 5"; do
  f=$out/bench_$(echo $args | tr -d '-' | tr ' ' '_').json
  timeout -s KILL 900 python bench.py $args > $f 2> ${f%.json}.err; echo "bench $args rc=$?" | tee -a $out/summary.txt
  python - "$f" <<'PY' | tee -a $out/summary.txt
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  value %.4e %s ms %.4f" % (j["value"], j["unit"], j["ms_per_step"]), {k: "%.3e" % v["value"] for k, v in j.get("end_to_end_nnet", {}).items()})
PY
done
bash tools/gpu_r06k.sh > $out/k.log 2>&1; cp gpurun_out/r06k/pmc_summary.txt $out/l1_embed_pmc.txt
DCA_E2E_IMPORT=tools/bin/cube3_avi.pt DCA_E2E_MAX_NODES=130000000 DCA_E2E_CHUNK=50 DCA_E2E_DEADLINE=1250 timeout -s KILL 1450 python tools/avi_e2e.py 0 1000 10000000 - 3 cube3 > $out/cube3_search_1000.log 2>&1
echo "search rc=$?" | tee -a $out/summary.txt
grep -v "^State: " $out/cube3_search_1000.log | tail -75 | cut -c1-200
