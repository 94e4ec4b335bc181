#!/bin/bash
# round 5 session aj: whole suite + smoke on the round's last code, then the evidence set (profile_round)
mkdir -p gpurun_out/r5aj
t0=$(date +%s); timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r5aj/suite.log 2>&1; echo "suite rc=$? secs=$(( $(date +%s) - t0 )) $(grep -E 'passed|failed' gpurun_out/r5aj/suite.log | tail -1)" | tee gpurun_out/r5aj/summary.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r5aj/summary.txt
bash tools/profile_round.sh r5zzz > gpurun_out/r5aj/profile_round.log 2>&1
tail -c 200 gpurun_out/r5aj/profile_round.log
