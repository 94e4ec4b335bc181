#!/bin/bash
# the GPU suite under the guard-page allocator (AERO_POOL_GUARD=1): an access past a pool block faults at the access
OUT=gpurun_out/r4d; mkdir -p $OUT
export AERO_CRASH_TRACE=1 AERO_CRASH_LOG=$PWD/$OUT/crash.log AERO_POOL_GUARD=1
for mod in test_gpu_parity test_gpu_stages test_gpu_aux test_gpu_host_handover test_gpu_worker_messages test_trace_file test_gpu_fallback_paths test_gpu_sharded_local test_gpu_air test_gpu_random_configs test_gpu_air_fuzz test_gpu_full_configs; do
  t0=$(date +%s)
  timeout 1200 python3 -m pytest tests/$mod.py -q -m gpu -p no:cacheprovider --tb=short > $OUT/$mod.log 2>&1
  echo "$mod rc=$? secs=$(( $(date +%s) - t0 ))" | tee -a $OUT/summary.txt
  tail -4 $OUT/$mod.log | cut -c1-300
done
cat $OUT/summary.txt
