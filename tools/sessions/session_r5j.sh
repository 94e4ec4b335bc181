#!/bin/bash
# round 5 session j: one proof alone - wall clock, and the kernel timeline of one proof (gaps = host round trips)
R=$PWD; O=$R/gpurun_out/r5j; mkdir -p $O
python3 tools/single_latency.py 20 2 300 | tee $O/single.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/single_latency.py 20 2 20 > $O/kt.log 2>&1
cd $R
f=$(find $O/kt -name "*kernel_trace.csv" | head -1); cp $f $O/kernel_trace.csv; rm -rf $O/kt
python3 tools/timeline_gaps.py $O/kernel_trace.csv 3 | tee $O/gaps.txt | tail -120
