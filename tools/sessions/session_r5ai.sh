#!/bin/bash
# round 5 session ai: every domain-only value of the constraint kernel from the per-shape table (5 words per row) - parity, then against the 2-word table's numbers
mkdir -p gpurun_out/r5ai
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py tests/test_gpu_aux.py tests/test_gpu_random_configs.py tests/test_gpu_switches.py -x -q -m gpu -k "not general and not pool" 2>&1 | tail -3 | tee gpurun_out/r5ai/parity.txt
for e in "AERO_CONS_INV_TABLE=1" "AERO_CONS_INV_TABLE=0" "AERO_CONS_INV_TABLE=1" "AERO_CONS_INV_TABLE=0"; do
  echo "$e"; env $e python3 tools/single_latency.py 20 2 300; env $e python3 bench.py --steps 10 --no-cpu-baseline --no-air-program --stages 2>&1 | grep -E "fib_constraints|^\{" | cut -c1-130
done | tee gpurun_out/r5ai/ab.txt
for w in "fib_2^20x72_blowup8_blake2s_base" "fib_2^20x2_blowup8_blake2s_quadratic"; do python3 bench.py --workload "$w" --steps 4 --no-cpu-baseline --no-air-program --stages 2>&1 | grep -E "fib_constraints|^\{" | cut -c1-130; done | tee -a gpurun_out/r5ai/ab.txt
