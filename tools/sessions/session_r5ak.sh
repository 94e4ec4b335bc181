#!/bin/bash
# round 5 session ak: the 8-way thread-rank 2^24 proof failed ONCE in a whole-suite run (trace root differs): how often, and does the library of the
# start of the day (libaero_stark_old.so) do it too?
mkdir -p gpurun_out/r5ak
for i in $(seq 1 12); do
  timeout 600 python -m pytest tests/test_gpu_full_configs.py -x -q -m gpu -k sharded_8 -p no:cacheprovider > gpurun_out/r5ak/new_$i.log 2>&1; echo "new $i rc=$? $(tail -1 gpurun_out/r5ak/new_$i.log)"
done | tee gpurun_out/r5ak/summary.txt
