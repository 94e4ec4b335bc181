#!/bin/bash
OUT=gpurun_out/r4j; mkdir -p $OUT
export AERO_CRASH_TRACE=1
timeout 1200 python3 -m pytest tests/test_gpu_host_handover.py tests/test_gpu_full_configs.py tests/test_gpu_sharded_local.py tests/test_gpu_bench_flow.py tests/test_gpu_rccl.py -q -m gpu -p no:cacheprovider --tb=short > $OUT/tests.log 2>&1
echo "tests rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/tests.log | cut -c1-300
python3 bench.py --steps 20 --warmup 3 --no-air-program 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['single_proof_ms'], d['single_proof_ms_hbm_resident'], d['cpu_baseline']['value'])" | tee -a $OUT/summary.txt
