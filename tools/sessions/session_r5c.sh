#!/bin/bash
# round 5 session c: the library built with -DGL_ASM against the default build, with both first-pass kernels, on one box
set -x
mkdir -p gpurun_out/r5c
ASM=$PWD/aero_amd/libaero_stark_asm.so
AERO_LIB_PATH=$ASM timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r5c/parity_asm.txt
bash tools/ntt_ab.sh r5c "AERO_NTT_F8X2=0" "AERO_NTT_F8X2=1" "AERO_NTT_F8X2=0 AERO_LIB_PATH=$ASM" "AERO_NTT_F8X2=1 AERO_LIB_PATH=$ASM" "AERO_NTT_F8X2=0"
