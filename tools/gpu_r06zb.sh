#!/bin/bash
# Round 6, GPU visit ZB: bench.py --env puzzle15 / puzzle24 / puzzle35 on the final build (engine-only value + the network-in-the-loop legs).
out=gpurun_out/r06zb
mkdir -p $out
for e in puzzle15 puzzle24 puzzle35; do
  timeout -s KILL 200 python bench.py --env $e > $out/bench_$e.json 2> $out/bench_$e.err; echo "bench --env $e rc=$?"
  python - "$out/bench_$e.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  value %.4e %s ms %.4f" % (j["value"], j["unit"], j["ms_per_step"]), {k: "%.3e" % v["value"] for k, v in j.get("end_to_end_nnet", {}).items()})
PY
done
