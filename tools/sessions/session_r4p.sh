#!/bin/bash
# Round 4, session p: small inverse transforms as 12 bits + ONE strided LDS pass (AERO_INV_LDS_PLAN=1) against 12 + two radix-16 register passes
OUT=gpurun_out/r4p; mkdir -p $OUT
for setting in "" "AERO_INV_LDS_PLAN=1" "AERO_INV_LDS_PLAN=1 AERO_INV_2PHASE_MIN=22"; do
  echo "=== [$setting]" | tee -a $OUT/ab.txt
  env $setting python3 tools/ntt_ab.py 20x1 20x2 19x2 18x2 16x2 14x2 2>&1 | tail -1 | tee -a $OUT/ab.txt
  for i in 1 2; do env $setting python3 tools/single_latency.py 2>&1 | tail -1 | tee -a $OUT/ab.txt; done
done
AERO_INV_LDS_PLAN=1 AERO_INV_2PHASE_MIN=22 timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/parity.txt
