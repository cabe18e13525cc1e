#!/bin/bash
# Round 6, GPU visit J: the build with the embedding-sum layer 1 selected for the sliding puzzles — smoke + full GPU suite, then the
# puzzle48 benches (search with the network in the loop; AVI update step) it changes.
out=gpurun_out/${1:-r06j}
mkdir -p $out
export TMPDIR=/tmp
bash tools/gpu_round.sh ${1:-r06j} nobench
for args in "--env puzzle48" "--workload avi --env puzzle48" "--workload avi --env puzzle48 --nnet_dtype bf16" "--workload avi --env puzzle15"; do
  f=$out/bench_$(echo $args | tr -d '-' | tr ' ' '_').json
  t0=$(date +%s); timeout -s KILL 900 python bench.py $args > $f 2> ${f%.json}.err; echo "bench $args rc=$? $(( $(date +%s) - t0 )) s" | tee -a $out/summary.txt
  python - "$f" <<'PY' | tee -a $out/summary.txt
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  value %.4e %s ms %.4f" % (j["value"], j["unit"], j["ms_per_step"]), {k: "%.3e" % v["value"] for k, v in j.get("end_to_end_nnet", {}).items()})
PY
done
timeout -s KILL 300 python tools/l1_embed_bench.py 2>&1 | tee $out/l1_embed_bench.txt
timeout -s KILL 600 python tools/nnet_forward_probe.py 2>&1 | tee $out/nnet_forward_probe.txt
