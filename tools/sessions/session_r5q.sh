#!/bin/bash
# round 5 session q: -DGL_ZMUL (mul / 160-bit sums / 2^24 shift with carries kept in VCC) against the default build on every bench workload, one box
mkdir -p gpurun_out/r5q
Z=$PWD/aero_amd/libaero_stark_zmul.so
AERO_LIB_PATH=$Z timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py tests/test_gpu_aux.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r5q/parity_zmul.txt
for rep in 1 2; do
  echo "== default" | tee -a gpurun_out/r5q/all.txt; bash tools/all_workloads.sh 2>&1 | cut -c1-170 | tee -a gpurun_out/r5q/all.txt
  echo "== zmul" | tee -a gpurun_out/r5q/all.txt; AERO_LIB_PATH=$Z bash tools/all_workloads.sh 2>&1 | cut -c1-170 | tee -a gpurun_out/r5q/all.txt
done
