#!/bin/bash
# Round 4, session o: the whole GPU suite as the driver runs it, smoke(), and the default bench line on the final code
OUT=gpurun_out/r4o; mkdir -p $OUT
export AERO_CRASH_LOG=$PWD/$OUT/crash.log
timeout 1200 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/suite.log 2>&1; echo "suite rc=$?" | tee $OUT/summary.txt
tail -3 $OUT/suite.log | tee -a $OUT/summary.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $OUT/summary.txt
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
head -c 600 $OUT/bench.json
