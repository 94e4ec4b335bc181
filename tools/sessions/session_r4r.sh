#!/bin/bash
# Round 4, session r: per-kernel times of single proofs on the final code (rocprofv3 --kernel-trace --stats)
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4r; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 80 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/tools/single_latency.py 20 2 60 > $OUT/kt.log 2>&1
tail -2 $OUT/kt.log
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/single_kernel_stats.csv \;
rm -rf $OUT/kt/*/*trace*.csv
ls $OUT
