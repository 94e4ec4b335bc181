#!/bin/bash
# round 5 session m: review-item tests (zero denominators, long sequences, pinned views / placement), then the whole GPU suite
mkdir -p gpurun_out/r5m
timeout 1800 python -m pytest tests/test_gpu_air.py tests/test_gpu_host_handover.py -q -m gpu -x -k "zero_denominator or million or pinned_views" 2>&1 | tail -15 | tee gpurun_out/r5m/new.txt
timeout 3400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r5m/suite.txt
