#!/bin/bash
# A/B of NTT variants on one box: per-kernel times (tools/ntt_ab.py) and the headline with 8 proofs in flight for each setting.
# usage: tools/ntt_ab.sh <tag> "<ENV=val ...>" "<ENV=val ...>" ...     (first setting "" = defaults)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for setting in "$@"; do
  echo "=== [$setting]" | tee -a $OUT/ab.txt
  env $setting python3 tools/ntt_ab.py 20x2 20x72 21x2 2>&1 | tail -1 | tee -a $OUT/ab.txt
  env $setting python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-air-program 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d.get('single_proof_ms'), d.get('single_proof_ms_hbm_resident'))" | tee -a $OUT/ab.txt
done
