#!/bin/bash
# round 5 session ab: divisor-inverse table (one row per thread) + the shorter inversion chain: parity, switch test, A/B against the previous commit's library
mkdir -p gpurun_out/r5ab
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py tests/test_gpu_aux.py tests/test_gpu_random_configs.py tests/test_gpu_air.py tests/test_gpu_switches.py -x -q -m gpu -k "not general and not pool" 2>&1 | tail -3 | tee gpurun_out/r5ab/parity.txt
for lib in old new old new; do
  if [ $lib = old ]; then export AERO_LIB_PATH=$PWD/aero_amd/libaero_stark_old.so; else unset AERO_LIB_PATH; fi
  echo "lib=$lib"; python3 tools/single_latency.py 20 2 300; python3 bench.py --steps 10 --no-cpu-baseline --no-air-program --stages 2>&1 | grep -E "fib_constraints|deep_kernel|^\{" | cut -c1-160
done | tee gpurun_out/r5ab/ab.txt
