#!/bin/bash
# round 5 session f: column-group copies in pool mode for wide traces (A/B on one box)
mkdir -p gpurun_out/r5f
for setting in "AERO_POOL_PIPELINE_MIN_W=0" "AERO_POOL_PIPELINE_MIN_W=16" "AERO_POOL_PIPELINE_MIN_W=0" "AERO_POOL_PIPELINE_MIN_W=16"; do
for w in "fib_2^20x72_blowup8_blake2s_base" "standin_miden_shape_2^22x(72+9aux)_deg8_fold4"; do
  env $setting python bench.py --workload "$w" --no-cpu-baseline --no-air-program --steps 6 --warmup 1 2>gpurun_out/r5f/err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$setting', d['config']['workload'], 'h2d incl', round(d['value']/1e9,3), 'resident', round(d['hbm_resident_value']/1e9,3), 'link frac', round(d['pcie']['frac'],3), 'single', round(d['single_proof_ms'],2), round(d['single_proof_ms_hbm_resident'],2))" | tee -a gpurun_out/r5f/ab.txt
done; done
tail -3 gpurun_out/r5f/err.txt
