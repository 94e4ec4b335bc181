#!/bin/bash
# round 5 session a: two-wave first pass - parity, then A/B against the round-4 kernel on the same box
set -x
mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r5a/parity.txt
cat gpurun_out/r5a/parity.txt
bash tools/ntt_ab.sh r5a "AERO_NTT_F8X2=0" "" "AERO_NTT_F8X2=0" ""
