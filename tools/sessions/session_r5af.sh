#!/bin/bash
# round 5 session af: whole suite + smoke + fuzzers on the code with the divisor table, the inversion chain and the coefficient-form DEEP; A/B lines
mkdir -p gpurun_out/r5af
t0=$(date +%s); timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r5af/suite.log 2>&1; echo "suite rc=$? secs=$(( $(date +%s) - t0 )) $(grep -E 'passed|failed' gpurun_out/r5af/suite.log | tail -1)" | tee gpurun_out/r5af/summary.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r5af/summary.txt
timeout 900 python3 tools/fuzz_configs.py 60 71 15 2>&1 | tail -1 | tee -a gpurun_out/r5af/summary.txt
timeout 900 python3 tools/fuzz_sharded.py 6 72 2>&1 | tail -2 | tee -a gpurun_out/r5af/summary.txt
timeout 900 python3 tools/stress_handover.py 40 73 2>&1 | tail -1 | tee -a gpurun_out/r5af/summary.txt
timeout 600 python3 tools/soak.py 2>&1 | tail -2 | tee -a gpurun_out/r5af/summary.txt
for f in 1 0 1 0; do echo "AERO_DEEP_COEFF=$f"; AERO_DEEP_COEFF=$f python3 tools/single_latency.py 20 2 300; AERO_DEEP_COEFF=$f python3 bench.py --steps 10 --no-cpu-baseline --no-air-program --stages 2>&1 | grep -E "deep|ntt_inv|^\{" | cut -c1-130; done | tee gpurun_out/r5af/deep_ab.txt
