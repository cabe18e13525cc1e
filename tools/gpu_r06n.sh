#!/bin/bash
# embedding-sum kernel iteration: tests + per-geometry timing only
out=gpurun_out/${1:-r06n}
mkdir -p $out
timeout -s KILL 600 python -m pytest tests/test_embed_hip.py -x -q 2>&1 | tail -5 | tee $out/tests.txt
timeout -s KILL 300 python tools/l1_embed_bench.py 2>&1 | tee $out/l1_embed_bench.txt
