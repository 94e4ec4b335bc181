#!/bin/bash
# the code added after the soak (general aux recurrences, sequence tables by sort-merge, aero_rccl_available): its tests, the same under
# the guard-page allocator, then the whole suite twice in the driver's command line
OUT=gpurun_out/r4i; mkdir -p $OUT
export AERO_CRASH_TRACE=1 AERO_CRASH_LOG=$PWD/$OUT/crash.log
timeout 900 python3 -m pytest tests/test_gpu_air.py tests/test_gpu_air_fuzz.py tests/test_gpu_rccl.py -q -m gpu -p no:cacheprovider -k "version2 or rccl or ranks or exchanges or world_of_one or bad_arguments or pairwise" --tb=short > $OUT/new.log 2>&1
echo "new rc=$?" | tee -a $OUT/summary.txt; tail -8 $OUT/new.log | cut -c1-300
AERO_POOL_GUARD=1 timeout 900 python3 -m pytest tests/test_gpu_air.py tests/test_gpu_air_fuzz.py -q -m gpu -p no:cacheprovider -k "version2" --tb=short > $OUT/guard.log 2>&1
echo "guard rc=$?" | tee -a $OUT/summary.txt; tail -3 $OUT/guard.log | cut -c1-300
tools/hunt_abort.sh 0 2 0
rm -rf gpurun_out/r4i_hunt; mv gpurun_out/hunt gpurun_out/r4i_hunt
cat $OUT/summary.txt
