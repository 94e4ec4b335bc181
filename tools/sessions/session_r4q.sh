#!/bin/bash
# Round 4, session q: the headline (8 proofs in flight) with and without AERO_INV_LDS_PLAN
OUT=gpurun_out/r4q; mkdir -p $OUT
for setting in "" "AERO_INV_LDS_PLAN=1" "" "AERO_INV_LDS_PLAN=1"; do
  echo "=== [$setting]" | tee -a $OUT/ab.txt
  env $setting python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-air-program 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d.get('single_proof_ms'), d.get('single_proof_ms_hbm_resident'), d.get('hbm_resident_value'))" | tee -a $OUT/ab.txt
done
