#!/bin/bash
# round 5 session r: the two-wave first pass on one-column launches only (one wave per SIMD or less in the one-wave form)
mkdir -p gpurun_out/r5r
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py tests/test_gpu_switches.py -x -q -m gpu -k "not general and not pool" 2>&1 | tail -3 | tee gpurun_out/r5r/parity.txt
python3 tools/ntt_ab.py 20x1 20x2 19x1 | tee gpurun_out/r5r/ab.txt
for i in 1 2 3; do python3 tools/single_latency.py 20 2 300; done | tee gpurun_out/r5r/single.txt
