#!/bin/bash
# round 5 session ae: every bench workload with the DEEP composition in coefficient form and in evaluation form; program workload too
mkdir -p gpurun_out/r5ae
for f in 1 0; do echo "AERO_DEEP_COEFF=$f"; AERO_DEEP_COEFF=$f bash tools/all_workloads.sh; done 2>&1 | tee gpurun_out/r5ae/all.txt
for f in 1 0; do echo "AERO_DEEP_COEFF=$f"; AERO_DEEP_COEFF=$f python3 bench.py --workload 'program_vm_shape_2^22x(72+9aux)_fold4' --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | cut -c1-400; done | tee gpurun_out/r5ae/program.txt
