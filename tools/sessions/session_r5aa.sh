#!/bin/bash
# round 5 session aa: boundary-divisor inverses of the FibAir constraint domain from a per-shape table instead of per-thread batch inversions
mkdir -p gpurun_out/r5aa
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py tests/test_gpu_aux.py tests/test_gpu_random_configs.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r5aa/parity.txt
for e in "AERO_CONS_INV_TABLE=0" "AERO_CONS_K=4" "AERO_CONS_K=2" "AERO_CONS_K=1" "AERO_CONS_INV_TABLE=0" "AERO_CONS_K=4"; do
  echo "$e"; env $e python3 tools/single_latency.py 20 2 200; env $e python3 bench.py --steps 10 --no-cpu-baseline --no-air-program --stages 2>&1 | grep -E "fib_constraints|^\{" | cut -c1-200
done | tee gpurun_out/r5aa/ab.txt
