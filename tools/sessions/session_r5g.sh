#!/bin/bash
# round 5 session g: copies and kernels of a pool of wide host-trace proofs under rocprofv3
R=$PWD; O=$R/gpurun_out/r5g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/host -- python3 $R/tools/pool_copy_trace.py 20 72 8 3 > $O/host.log 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/res -- python3 $R/tools/pool_copy_trace.py 20 72 8 3 resident > $O/res.log 2>&1
cd $R
tail -2 $O/host.log; python3 tools/pool_copy_trace.py --summarise $O/host | tee $O/host_summary.txt
tail -2 $O/res.log; python3 tools/pool_copy_trace.py --summarise $O/res | tee $O/res_summary.txt
f=$(find $O/host -name "*memory_copy_trace.csv" | head -1); cp $f $O/host_memory_copy_trace.csv; head -3 $f
f=$(find $O/host -name "*kernel_trace.csv" | head -1); cp $f $O/host_kernel_trace.csv
f=$(find $O/res -name "*kernel_trace.csv" | head -1); cp $f $O/res_kernel_trace.csv
rm -rf $O/host $O/res
