#!/bin/bash
# round 5 session an: two whole-suite runs (driver's command line) + smoke on the round's last code, then the evidence set
mkdir -p gpurun_out/r5an
for i in 1 2; do
  t0=$(date +%s); timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r5an/suite_$i.log 2>&1; echo "suite run $i rc=$? secs=$(( $(date +%s) - t0 )) $(grep -E 'passed|failed' gpurun_out/r5an/suite_$i.log | tail -1)" | tee -a gpurun_out/r5an/summary.txt
done
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r5an/summary.txt
bash tools/profile_round.sh r5final > gpurun_out/r5an/profile_round.log 2>&1
tail -c 200 gpurun_out/r5an/profile_round.log
