#!/bin/bash
# round 5 session e: the host link (tools/ubench_h2d) and the H2D-inclusive / resident figures of the wide workloads
mkdir -p gpurun_out/r5e
tools/ubench_h2d gpurun_out/r5e/h2d.json | tee gpurun_out/r5e/h2d.txt
for w in "fib_2^20x72_blowup8_blake2s_base" "standin_miden_shape_2^22x(72+9aux)_deg8_fold4" "fib_2^20x2_blowup8_blake2s_base"; do
  python bench.py --workload "$w" --no-cpu-baseline --no-air-program --steps 4 --warmup 1 2>gpurun_out/r5e/err.txt | tee -a gpurun_out/r5e/bench.jsonl | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['config']['workload'], 'h2d incl', d['value']/1e9, 'resident', d['hbm_resident_value']/1e9, 'pcie', d['pcie'], 'single', d['single_proof_ms'], d['single_proof_ms_hbm_resident'])"
done
tail -3 gpurun_out/r5e/err.txt
