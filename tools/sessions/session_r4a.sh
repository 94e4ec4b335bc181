#!/bin/bash
# GPU session: (1) the tests added this round, (2) NTT variant A/B, (3) the suite's core modules with the guard-page allocator,
# (4) NTT counter passes. Everything under gpurun_out/r4a/.
OUT=gpurun_out/r4a; mkdir -p $OUT
export AERO_CRASH_TRACE=1 AERO_CRASH_LOG=$PWD/$OUT/crash.log
t0=$(date +%s)
timeout 1500 python3 -m pytest tests/test_gpu_air.py tests/test_gpu_air_fuzz.py tests/test_gpu_sharded_local.py tests/test_gpu_rccl.py tests/test_gpu_host_handover.py -q -m gpu -p no:cacheprovider \
   -k "version2 or lengths or local_group or refused or two_ranks or host" > $OUT/new_tests.log 2>&1
echo "new tests rc=$? secs=$(( $(date +%s) - t0 ))" | tee -a $OUT/summary.txt; tail -15 $OUT/new_tests.log
t0=$(date +%s)
timeout 900 python3 -m pytest tests/test_gpu_bench_flow.py -q -m gpu -p no:cacheprovider -k "refused" > $OUT/bench_flow_refused.log 2>&1
echo "bench flow refused rc=$? secs=$(( $(date +%s) - t0 ))" | tee -a $OUT/summary.txt; tail -5 $OUT/bench_flow_refused.log
t0=$(date +%s)
tools/ntt_ab.sh r4a "" "AERO_NTT_R6=1" "AERO_NTT_R6=2" "AERO_NTT_R6=3" "AERO_NTT_R6=1 AERO_NTT_BUF=1" "AERO_INV_2PHASE_MIN=21" "AERO_INV_2PHASE_MIN=21 AERO_NTT_R6=1 AERO_NTT_BUF=1" > $OUT/ab_stdout.txt 2>&1
echo "ntt ab secs=$(( $(date +%s) - t0 ))" | tee -a $OUT/summary.txt; cat $OUT/ab.txt
t0=$(date +%s)
AERO_POOL_GUARD=1 timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py tests/test_gpu_aux.py tests/test_gpu_host_handover.py tests/test_gpu_sharded_local.py tests/test_gpu_air.py -q -m gpu -p no:cacheprovider \
   -k "not large_properties and not full_size and not 2p2 and not vm_72_9" > $OUT/guard.log 2>&1
echo "guard rc=$? secs=$(( $(date +%s) - t0 ))" | tee -a $OUT/summary.txt; tail -30 $OUT/guard.log
t0=$(date +%s)
tools/ntt_gap.sh r4a_ntt_gap 20x2 20x72 > $OUT/ntt_gap_stdout.txt 2>&1
echo "ntt gap secs=$(( $(date +%s) - t0 ))" | tee -a $OUT/summary.txt
cat $OUT/summary.txt
