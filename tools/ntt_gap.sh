#!/bin/bash
# Counter attribution of the NTT family's gap to its in-register ceiling: kernel durations + five PMC passes
# over tools/ntt_workload.py, summarised per kernel by profiles/ntt_gap_summary.py.   usage: tools/ntt_gap.sh <tag> [shape ...]
set -u
TAG=${1:-r4_ntt}; shift
SHAPES=${*:-20x2 20x72}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$ROOT/tools/ntt_workload.py" 5 $SHAPES > "$OUT/kt.log" 2>&1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" \
           "SQ_WAVES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/tools/ntt_workload.py" 2 $SHAPES > "$OUT/p$i.log" 2>&1
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/p${i}_counter_collection.csv"
  rm -rf "$OUT/p$i"
done
f=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
f=$(find "$OUT/kt" -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_trace.csv"
rm -rf "$OUT/kt"
cd "$ROOT"
python3 profiles/ntt_gap_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
