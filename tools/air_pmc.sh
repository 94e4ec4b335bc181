#!/bin/bash
# SQ counters of the AIR-program evaluator on the VM-shaped program (air_jit_kernel by default, the interpreter air_constraints_kernel with AERO_AIR_JIT=0), four separate --pmc passes
# (counters only, no other trace domain). usage: bash tools/air_pmc.sh <tag> [log_n]   -> gpurun_out/<tag>/air_pmc_pass{1..4}.csv
TAG=${1:-rX}; LOGN=${2:-18}
OUT=$(pwd)/gpurun_out/$TAG; mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" \
           "SQ_IFETCH SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_WAVES"; do
  i=$((i + 1))
  rm -rf "/tmp/air_pmc_$i"
  rocprofv3 --pmc $set -d "/tmp/air_pmc_$i" --output-format csv -- python3 "$REPO/tools/air_bench.py" --vm 26,9,16 --log-n "$LOGN" --fold 4 --reps 1 > "$OUT/air_pmc_pass$i.log" 2>&1
  f=$(find "/tmp/air_pmc_$i" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && { head -1 "$f"; grep -E "air_constraints_kernel|air_jit_kernel" "$f"; } > "$OUT/air_pmc_pass$i.csv"
done
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
acc = collections.defaultdict(list)
for p in sorted(glob.glob(sys.argv[1] + "/air_pmc_pass*.csv")):
    for r in csv.DictReader(open(p)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:24s} launches {len(v):3d}  mean {sum(v)/len(v):16.1f}")
PY
