#!/bin/bash
# validation of the final code of round 4: four whole-suite runs in the driver's command line, the round's profile set, every workload
tools/hunt_abort.sh 0 4 0
rm -rf gpurun_out/r4g_hunt; mv gpurun_out/hunt gpurun_out/r4g_hunt
tools/profile_round.sh r4 > gpurun_out/r4_profile_stdout.txt 2>&1
bash tools/all_workloads.sh > gpurun_out/r4_all_workloads.txt 2>&1
cat gpurun_out/r4_all_workloads.txt
python3 tools/single_latency.py 20 2 30 > gpurun_out/r4_single_latency.txt 2>&1; tail -3 gpurun_out/r4_single_latency.txt
