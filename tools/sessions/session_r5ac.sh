#!/bin/bash
# round 5 session ac: DEEP composition in coefficient form - parity, switch test, A/B by the switch
mkdir -p gpurun_out/r5ac
timeout 1500 python -m pytest tests/test_gpu_switches.py -x -q -m gpu -k "deep_composition" 2>&1 | tail -12 | cut -c1-400 | tee gpurun_out/r5ac/parity.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py tests/test_gpu_aux.py tests/test_gpu_random_configs.py tests/test_gpu_air.py -x -q -m gpu 2>&1 | tail -3 | tee -a gpurun_out/r5ac/parity.txt
for f in 0 1 0 1; do
  echo "AERO_DEEP_COEFF=$f"; AERO_DEEP_COEFF=$f python3 tools/single_latency.py 20 2 300; AERO_DEEP_COEFF=$f python3 bench.py --steps 10 --no-cpu-baseline --no-air-program --stages 2>&1 | grep -E "deep|ntt_inv|^\{" | cut -c1-160
done | tee gpurun_out/r5ac/ab.txt
