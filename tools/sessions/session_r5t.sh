#!/bin/bash
# round 5 session t: the contiguous inverse last pass with two wavefronts per tile (ntt_inv_last_pass_11x2): parity, per-kernel A/B, one proof alone
mkdir -p gpurun_out/r5t
timeout 1200 python -m pytest tests/test_gpu_switches.py -x -q -m gpu -k "every_ntt_plan or two_wave" 2>&1 | tail -5 | tee gpurun_out/r5t/parity.txt
for t in 0 2048 8192; do AERO_INV_X2_TILES=$t python3 tools/ntt_ab.py 21x1 21x2 22x1 22x2; done | tee gpurun_out/r5t/ab.txt
for t in 0 2048 0 2048; do echo "AERO_INV_X2_TILES=$t"; AERO_INV_X2_TILES=$t python3 tools/single_latency.py 20 2 300; done | tee gpurun_out/r5t/single.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -x -q -m gpu 2>&1 | tail -3 | tee -a gpurun_out/r5t/parity.txt
