#!/bin/bash
# round 5 session ag: kernel timeline of one proof alone on the round's last code (rocprofv3 --kernel-trace), single-proof latency, host gaps
R=$PWD; O=$R/gpurun_out/r5ag; mkdir -p $O
for i in 1 2 3; do python3 tools/single_latency.py 20 2 300; done | tee $O/single.txt
AERO_HOST_GAPS=1 python3 tools/single_latency.py 20 2 6 2>&1 | tail -3 | tee -a $O/single.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/single_latency.py 20 2 20 > $O/kt.log 2>&1
cd $R
f=$(find $O/kt -name "*kernel_trace.csv" | head -1); cp $f $O/kernel_trace.csv; rm -rf $O/kt
python3 tools/timeline_gaps.py $O/kernel_trace.csv 3 > $O/gaps.txt; tail -70 $O/gaps.txt
