#!/bin/bash
# round 5 session am: AERO_POISON_ALLOC=1 (every pool block filled with 0xA5 when handed out): does any proof depend on a word nobody wrote?
mkdir -p gpurun_out/r5am
AERO_POISON_ALLOC=1 timeout 1500 python -m pytest tests/test_gpu_full_configs.py tests/test_gpu_sharded_local.py tests/test_gpu_parity.py tests/test_gpu_stages.py tests/test_gpu_aux.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r5am/poison.log 2>&1; echo "poison rc=$? $(grep -E 'passed|failed' gpurun_out/r5am/poison.log | tail -1)" | tee gpurun_out/r5am/summary.txt
grep -n "^E  \|FAILED" gpurun_out/r5am/poison.log | head -10 | cut -c1-300 | tee -a gpurun_out/r5am/summary.txt
AERO_POISON_ALLOC=1 timeout 600 python3 tools/fuzz_sharded.py 6 91 2>&1 | tail -2 | tee -a gpurun_out/r5am/summary.txt
AERO_POISON_ALLOC=1 timeout 600 python3 tools/fuzz_configs.py 40 92 15 2>&1 | tail -1 | tee -a gpurun_out/r5am/summary.txt
