#!/bin/bash
# Sweep of the code-generation switches of the run-time compiled AIR kernel (AERO_AIR_JIT_TUNE, aero_amd/csrc/air_jit.hip) on FibAir(72)
# as a program (against the hard-wired kernel) and on the VM-shaped program. usage: bash tools/air_jit_variants.sh <tag> ["tune1" "tune2" ...]
TAG=${1:-rX}; shift
mkdir -p gpurun_out/$TAG
[ $# -eq 0 ] && set -- "" "rows=2" "rows=2,barrier=0" "barrier=0" "rows=1" "early=0"
for tune in "$@"; do
  echo "{\"tune\": \"$tune\"}"
  AERO_AIR_JIT_TUNE="$tune" python tools/air_bench.py --width 72 --log-n 20 --reps 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'wl': d['workload'], 'hard': d['hard_wired_ms']['fib_constraints_kernel'], 'prog': d['program_ms'], 'ratio': d['constraint_kernel_ratio']}))"
  AERO_AIR_JIT_TUNE="$tune" python tools/air_bench.py --width 72 --log-n 18 --aux 9,16,8 --reps 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'wl': d['workload'], 'hard': d['hard_wired_ms']['fib_constraints_kernel'], 'prog': d['program_ms'], 'ratio': d['constraint_kernel_ratio']}))"
  AERO_AIR_JIT_TUNE="$tune" python tools/air_bench.py --vm 26,9,16 --log-n 20 --fold 4 --reps 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'wl': d['workload'], 'ms': d['ms']}))"
done > gpurun_out/$TAG/air_jit_variants.jsonl 2>&1
cat gpurun_out/$TAG/air_jit_variants.jsonl
