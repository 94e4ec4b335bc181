#!/bin/bash
# round 5 session x: whole suite + smoke + bench on the three-phase inverse passes (final plan: 12-bit where it saves a strided pass, 9-bit strided LDS pass for 1 x 2^21)
mkdir -p gpurun_out/r5x
t0=$(date +%s); timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r5x/suite.log 2>&1; echo "suite rc=$? secs=$(( $(date +%s) - t0 )) $(tail -1 gpurun_out/r5x/suite.log)" | tee gpurun_out/r5x/summary.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r5x/summary.txt
python3 bench.py 2>/dev/null | tail -1 > gpurun_out/r5x/bench.json
AERO_LIB_PATH=$PWD/aero_amd/libaero_stark_old.so python3 bench.py 2>/dev/null | tail -1 > gpurun_out/r5x/bench_old.json
python3 bench.py 2>/dev/null | tail -1 > gpurun_out/r5x/bench2.json
for i in 1 2; do python3 tools/single_latency.py 20 2 300; AERO_LIB_PATH=$PWD/aero_amd/libaero_stark_old.so python3 tools/single_latency.py 20 2 300; done | tee gpurun_out/r5x/single.txt
python3 tools/ntt_ab.py 20x2 21x1 20x1 18x72 24x2 | tee gpurun_out/r5x/ab.txt
