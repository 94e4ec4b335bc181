#!/bin/bash
# round 5 session i: the new switch / plan-coverage tests
mkdir -p gpurun_out/r5i
timeout 2400 python -m pytest tests/test_gpu_switches.py -q -m gpu 2>&1 | tail -40 | tee gpurun_out/r5i/switches.txt
