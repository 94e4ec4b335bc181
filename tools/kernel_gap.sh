#!/bin/bash
# Counter attribution for any kernel family: a kernel trace + the five PMC passes of tools/ntt_gap.sh over an arbitrary python tool, summarised
# per kernel by profiles/ntt_gap_summary.py. usage: tools/kernel_gap.sh <tag> <kernel prefixes, comma separated> <tool.py> [tool args ...]
set -u
TAG=$1; PREF=$2; TOOL=$3; shift 3
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$ROOT/$TOOL" "$@" > "$OUT/kt.log" 2>&1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/$TOOL" "$@" > "$OUT/p$i.log" 2>&1
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/p${i}_counter_collection.csv"
  rm -rf "$OUT/p$i"
done
f=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
rm -rf "$OUT/kt"
cd "$ROOT"
GAP_PREFIX=$PREF python3 profiles/ntt_gap_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
rm -f "$OUT"/p*_counter_collection.csv
cat "$OUT/summary.txt"
