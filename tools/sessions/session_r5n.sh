#!/bin/bash
# round 5 session n: results stored in mapped pinned memory by the producing kernels, one opening launch - parity, then one proof alone
R=$PWD; O=$R/gpurun_out/r5n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py tests/test_gpu_aux.py tests/test_gpu_random_configs.py -x -q -m gpu 2>&1 | tail -4 | tee $O/parity.txt
for i in 1 2 3; do python3 tools/single_latency.py 20 2 300; done | tee $O/single.txt
AERO_QUERY_TIMING=1 python3 tools/single_latency.py 20 2 6 2>&1 | tail -7 | tee -a $O/single.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/single_latency.py 20 2 20 > $O/kt.log 2>&1
cd $R
f=$(find $O/kt -name "*kernel_trace.csv" | head -1); cp $f $O/kernel_trace.csv; rm -rf $O/kt
python3 tools/timeline_gaps.py $O/kernel_trace.csv 3 > $O/gaps.txt; grep "gap\b.*<--\|kernels " $O/gaps.txt | tail -30
