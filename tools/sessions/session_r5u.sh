#!/bin/bash
# round 5 session u: two-wave inverse last pass - parity on every plan, and the A/B on wide launches (is there a launch the one-wave form still wins?)
mkdir -p gpurun_out/r5u
timeout 1500 python -m pytest tests/test_gpu_switches.py -x -q -m gpu -k "every_ntt_plan or two_wave" 2>&1 | tail -5 | tee gpurun_out/r5u/parity.txt
for t in 0 100000000; do AERO_INV_X2_TILES=$t python3 tools/ntt_ab.py 20x72 20x16 24x2 18x72 16x255; done | tee gpurun_out/r5u/ab_wide.txt
