#!/bin/bash
# Round 4, session l: where should the inverse transform switch from the LDS rounds to the two-phase last pass + register passes?
# (the DEEP stage of a 2^20-row proof inverts ONE column of 2^20 points: 2^20 elements, below the 2^21 threshold of the time)
OUT=gpurun_out/r4l; mkdir -p $OUT
for setting in "" "AERO_INV_2PHASE_MIN=20" "AERO_INV_2PHASE_MIN=19" "AERO_INV_2PHASE_MIN=18" "AERO_INV_2PHASE_MIN=16"; do
  echo "=== [$setting]" | tee -a $OUT/ab.txt
  env $setting python3 tools/ntt_ab.py 20x1 19x2 19x1 18x2 18x1 16x2 20x2 2>&1 | tail -1 | tee -a $OUT/ab.txt
  env $setting python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-air-program 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d.get('single_proof_ms'), d.get('single_proof_ms_hbm_resident'))" | tee -a $OUT/ab.txt
done
