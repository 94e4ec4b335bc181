#!/bin/bash
# round 5 session s: robustness on the final code - the whole suite three times (driver's command line), seeded fuzz over the option space,
# sharded fuzz, hand-over stress, program stress, soak
mkdir -p gpurun_out/r5s
for i in 1 2 3; do
  t0=$(date +%s); timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r5s/suite_$i.log 2>&1; echo "suite run $i rc=$? secs=$(( $(date +%s) - t0 )) $(tail -1 gpurun_out/r5s/suite_$i.log)" | tee -a gpurun_out/r5s/summary.txt
done
timeout 900 python3 tools/fuzz_configs.py 40 51 15 2>&1 | tail -2 | tee -a gpurun_out/r5s/summary.txt
timeout 900 python3 tools/fuzz_sharded.py 6 52 2>&1 | tail -2 | tee -a gpurun_out/r5s/summary.txt
timeout 900 python3 tools/stress_handover.py 40 53 2>&1 | tail -2 | tee -a gpurun_out/r5s/summary.txt
timeout 900 python3 tools/stress_programs.py 3 2>&1 | tail -2 | tee -a gpurun_out/r5s/summary.txt
timeout 600 python3 tools/soak.py 2>&1 | tail -2 | tee -a gpurun_out/r5s/summary.txt
