#!/bin/bash
# final validation of the round: the whole -m gpu suite six times in the driver's command line, then the round's profile set
tools/hunt_abort.sh 0 6 200
mv gpurun_out/hunt gpurun_out/r4e_hunt
tools/profile_round.sh r4 > gpurun_out/r4_profile_stdout.txt 2>&1
tail -c 1500 gpurun_out/r4/bench.json
