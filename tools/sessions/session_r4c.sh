#!/bin/bash
OUT=gpurun_out/r4c; mkdir -p $OUT
export AERO_CRASH_TRACE=1
echo "== guard" > $OUT/guard_debug.txt
AERO_POOL_GUARD=1 timeout 300 python3 tools/guard_debug.py >> $OUT/guard_debug.txt 2>&1
echo "== normal" >> $OUT/guard_debug.txt
timeout 300 python3 tools/guard_debug.py >> $OUT/guard_debug.txt 2>&1
cat $OUT/guard_debug.txt
