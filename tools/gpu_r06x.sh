#!/bin/bash
# Round 6, GPU visit X (as it was run): the one cube3 test state visit W left unsolved inside 1.3e8 node ids, on its own with a 1e9-id pool.
mkdir -p gpurun_out/r06x
DCA_E2E_STATES=564 DCA_E2E_IMPORT=tools/bin/cube3_avi.pt DCA_E2E_MAX_NODES=1000000000 timeout -s KILL 500 python tools/avi_e2e.py 0 1 10000000 - 3 cube3 > gpurun_out/r06x/state564.log 2>&1
grep -v "^device\|amdgpu" gpurun_out/r06x/state564.log | tail -40
