#!/bin/bash
# Round 6, GPU visit G: records on the (near-)final build — fp8 MFMA shapes probe, tolerance tests with their printed values,
# rocprofv3 records of the engine / gather kernel / network, gemm bench.
out=gpurun_out/r06g
mkdir -p $out
export TMPDIR=/tmp
tools/bin/mfma_power_probe 4000 > $out/mfma_power_probe.txt 2>&1
timeout -s KILL 900 python -m pytest tests/test_parity_configs_hip.py tests/test_gemm8_hip.py -m gpu -q -s --timeout 600 -p no:cacheprovider -k "tolerance or deviation or fp8_layer1" > $out/pytest_tol.log 2>&1; echo "tol rc=$?" | tee $out/summary.txt
grep -E "seed 20|puzzle48 \|h\||deviation / max|passed|failed" $out/pytest_tol.log | cut -c1-700 | tee -a $out/summary.txt
timeout -s KILL 900 python tools/gemm_bench.py 204800 > $out/gemm_bench.txt 2>&1
bash tools/profile_expand.sh r06 > $out/profile_expand.log 2>&1
bash tools/profile_round.sh r06 20 > $out/profile_round.log 2>&1
cat $out/mfma_power_probe.txt
