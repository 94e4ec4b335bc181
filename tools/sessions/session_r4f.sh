#!/bin/bash
OUT=gpurun_out/r4f; mkdir -p $OUT
tools/ntt_ab.sh r4f "" "AERO_NTT_F8_CHAINS=2" "AERO_NTT_F8_CHAINS=4" > $OUT/ab_stdout.txt 2>&1
cat $OUT/ab.txt
for setting in "" "AERO_NTT_BUF=1" "AERO_NTT_F8_CHAINS=2 AERO_NTT_BUF=1"; do
  echo "=== config-5 stand-in [$setting]" | tee -a $OUT/c5.txt
  env $setting python3 bench.py --workload 'standin_miden_shape_2^22x(72+9aux)_deg8_fold4' --no-cpu-baseline --no-air-program --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['hbm_resident_value'], d['single_proof_ms'], d['single_proof_ms_hbm_resident'])" | tee -a $OUT/c5.txt
done
