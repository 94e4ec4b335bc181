#!/bin/bash
# round 5 session al: the prefix of the suite that met the one wrong sharded proof (parity, stages, full configurations in one process), in a loop
mkdir -p gpurun_out/r5al
for i in $(seq 1 8); do
  AERO_TEST_STOP_AFTER=test_gpu_full_configs timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r5al/run_$i.log 2>&1; echo "run $i rc=$? $(grep -E 'passed|failed' gpurun_out/r5al/run_$i.log | tail -1)"
done | tee gpurun_out/r5al/summary.txt
