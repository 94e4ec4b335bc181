#!/bin/bash
# round 5 session l: chunked exchange parity (thread ranks sharing the GPU) + the sharded bench flow with chunks
mkdir -p gpurun_out/r5l
timeout 2400 python -m pytest tests/test_gpu_sharded_local.py -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/r5l/tests.txt
timeout 900 python -m pytest tests/test_gpu_bench_flow.py tests/test_gpu_sharded.py -q -m gpu -x 2>&1 | tail -5 | tee -a gpurun_out/r5l/tests.txt
