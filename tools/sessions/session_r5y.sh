#!/bin/bash
# round 5 session y: fuzzers / stress / soak and the big proofs on the rewritten inverse passes
mkdir -p gpurun_out/r5y
timeout 900 python3 tools/fuzz_configs.py 60 61 15 2>&1 | tail -2 | tee -a gpurun_out/r5y/summary.txt
timeout 900 python3 tools/fuzz_sharded.py 6 62 2>&1 | tail -2 | tee -a gpurun_out/r5y/summary.txt
timeout 900 python3 tools/stress_handover.py 40 63 2>&1 | tail -2 | tee -a gpurun_out/r5y/summary.txt
timeout 900 python3 tools/stress_programs.py 3 2>&1 | tail -2 | tee -a gpurun_out/r5y/summary.txt
timeout 600 python3 tools/soak.py 2>&1 | tail -2 | tee -a gpurun_out/r5y/summary.txt
