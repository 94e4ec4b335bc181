#!/bin/bash
# round 5 session p: multiplication with the carries kept in VCC (opaque zero registers, -DGL_ZMUL) against the default build, one box
set -x
mkdir -p gpurun_out/r5p
Z=$PWD/aero_amd/libaero_stark_zmul.so
AERO_LIB_PATH=$Z timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r5p/parity_zmul.txt
bash tools/ntt_ab.sh r5p "" "AERO_LIB_PATH=$Z" "" "AERO_LIB_PATH=$Z"
