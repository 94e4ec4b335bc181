#!/bin/bash
# round 5 session v: three-phase contiguous inverse passes (2048- and 4096-point tiles) against the previous commit's library (libaero_stark_old.so)
mkdir -p gpurun_out/r5v
timeout 1500 python -m pytest tests/test_gpu_switches.py tests/test_gpu_parity.py tests/test_gpu_stages.py tests/test_gpu_fallback_paths.py -x -q -m gpu -k "not general and not pool" 2>&1 | tail -5 | tee gpurun_out/r5v/parity.txt
for lib in old new old new; do
  if [ $lib = old ]; then export AERO_LIB_PATH=$PWD/aero_amd/libaero_stark_old.so; else unset AERO_LIB_PATH; fi
  python3 tools/ntt_ab.py 20x2 21x1 20x1 19x2 18x2 16x2 14x2 20x72
done | tee gpurun_out/r5v/ab.txt
for lib in old new old new; do
  if [ $lib = old ]; then export AERO_LIB_PATH=$PWD/aero_amd/libaero_stark_old.so; else unset AERO_LIB_PATH; fi
  echo "lib=$lib"; python3 tools/single_latency.py 20 2 300
done | tee gpurun_out/r5v/single.txt
unset AERO_LIB_PATH
python3 bench.py 2>/dev/null | tail -1 | tee gpurun_out/r5v/bench_new.json
AERO_LIB_PATH=$PWD/aero_amd/libaero_stark_old.so python3 bench.py 2>/dev/null | tail -1 | tee gpurun_out/r5v/bench_old.json
