#!/bin/bash
# One GPU-box visit: smoke, the GPU test suite (engine tests first, each stage in its own process under a hard
# timeout so a hung kernel cannot eat the box), then the default bench.  Everything lands under gpurun_out/$1/.
out=gpurun_out/${1:-run}
mkdir -p "$out"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
echo "== smoke" | tee "$out/summary.txt"
timeout -s KILL 600 python __graft_entry__.py smoke > "$out/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$out/summary.txt"
echo "== engine tests" | tee -a "$out/summary.txt"
timeout -s KILL 1500 python -m pytest tests/test_engine_hip.py tests/test_engine_fallback_hip.py tests/test_puzzle_optimal_hip.py \
    -m gpu -q --timeout 600 -p no:cacheprovider > "$out/pytest_engine.log" 2>&1; echo "engine rc=$?" | tee -a "$out/summary.txt"
tail -n 40 "$out/pytest_engine.log" | tee -a "$out/summary.txt"
echo "== other tests" | tee -a "$out/summary.txt"
timeout -s KILL 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider \
    --deselect tests/test_engine_hip.py --deselect tests/test_engine_fallback_hip.py --deselect tests/test_puzzle_optimal_hip.py \
    > "$out/pytest_rest.log" 2>&1; echo "rest rc=$?" | tee -a "$out/summary.txt"
tail -n 60 "$out/pytest_rest.log" | tee -a "$out/summary.txt"
if [ "${2:-bench}" = "bench" ]; then
  echo "== bench" | tee -a "$out/summary.txt"
  timeout -s KILL 900 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc=$?" | tee -a "$out/summary.txt"
  tail -c 6000 "$out/bench_default.json" | tee -a "$out/summary.txt"
  tail -n 15 "$out/bench_default.err" | tee -a "$out/summary.txt"
fi
