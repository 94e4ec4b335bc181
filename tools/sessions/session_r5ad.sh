#!/bin/bash
# round 5 session ad: coefficient-form DEEP: size of the one-workgroup level (2^14 / 2^12 / 2^10 values per chain)
mkdir -p gpurun_out/r5ad
timeout 600 python -m pytest tests/test_gpu_switches.py -x -q -m gpu -k "deep_composition" 2>&1 | tail -2 | tee gpurun_out/r5ad/parity.txt
for f in 14 12 10 13 11; do
  echo "AERO_DEEP_MID_BITS=$f"; AERO_DEEP_MID_BITS=$f python3 tools/single_latency.py 20 2 300; AERO_DEEP_MID_BITS=$f python3 bench.py --steps 6 --no-cpu-baseline --no-air-program --stages 2>&1 | grep -E "deep|^\{" | cut -c1-130
done | tee gpurun_out/r5ad/ab.txt
