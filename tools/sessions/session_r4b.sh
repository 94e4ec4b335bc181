#!/bin/bash
OUT=gpurun_out/r4b; mkdir -p $OUT
export AERO_CRASH_TRACE=1 AERO_CRASH_LOG=$PWD/$OUT/crash.log
cat /sys/kernel/mm/transparent_hugepage/enabled > $OUT/host.txt 2>&1; cat /proc/sys/kernel/numa_balancing >> $OUT/host.txt 2>&1; nproc >> $OUT/host.txt
timeout 120 tools/vmm_probe > $OUT/vmm_probe.txt 2>&1; echo "probe rc=$?" | tee -a $OUT/summary.txt
timeout 120 tools/vmm_probe oob > $OUT/vmm_probe_oob.txt 2>&1; echo "probe oob rc=$?" | tee -a $OUT/summary.txt
cat $OUT/vmm_probe.txt; tail -5 $OUT/vmm_probe_oob.txt
AERO_POOL_GUARD=1 timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider --tb=short -k "not large_properties and not full_size" > $OUT/guard_x.log 2>&1
echo "guard -x rc=$?" | tee -a $OUT/summary.txt
head -c 6000 $OUT/guard_x.log
