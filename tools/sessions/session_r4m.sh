#!/bin/bash
# Round 4, session m: LDS-round passes with 512 threads per tile on small launches (AERO_NTT_LDS512_MAX), alone and with the
# inverse transform's two-phase threshold moved up (so that the 2-column 2^20 interpolation takes the LDS last pass again).
OUT=gpurun_out/r4m; mkdir -p $OUT
for setting in "" "AERO_NTT_LDS512_MAX=21" "AERO_NTT_LDS512_MAX=21 AERO_INV_2PHASE_MIN=22" "AERO_NTT_LDS512_MAX=22 AERO_INV_2PHASE_MIN=23" "AERO_NTT_LDS512_MAX=20"; do
  echo "=== [$setting]" | tee -a $OUT/ab.txt
  env $setting python3 tools/ntt_ab.py 20x1 20x2 21x1 19x2 16x2 2>&1 | tail -1 | tee -a $OUT/ab.txt
  env $setting python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-air-program 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d.get('single_proof_ms'), d.get('single_proof_ms_hbm_resident'))" | tee -a $OUT/ab.txt
done
AERO_NTT_LDS512_MAX=40 timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -x -q -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/parity_lds512.txt
