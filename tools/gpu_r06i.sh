#!/bin/bash
# Round 6, GPU visit I: layer 1 as an embedding sum — its tests, per-geometry timing against the MFMA kernel, whole-network forward.
out=gpurun_out/${1:-r06i}
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_embed_hip.py -x -q 2>&1 | tail -15 | tee $out/tests.txt
timeout -s KILL 300 python tools/l1_embed_bench.py 2>&1 | tee $out/l1_embed_bench.txt
timeout -s KILL 600 python tools/nnet_forward_probe.py 2>&1 | tee $out/nnet_forward_probe.txt
