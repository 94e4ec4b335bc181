#!/bin/bash
# One workload's evidence set (round 6: every BASELINE config, not only the headline): the bench line with its per-kernel table, a
# single-stream rocprofv3 kernel trace, the SQ counter pass and the two traffic passes (FETCH_SIZE / WRITE_SIZE, each in a run of its own,
# --kernel-trace only), then profiles/summarize_workload.py. usage: tools/profile_workload.sh <tag> <short> <workload> [bench steps]
# Leaves gpurun_out/<tag>/<short>_{bench.json,bench_stages.txt,kernel_stats_single_stream.csv,sq_counters.csv,pmc_traffic.json,summary.json}
set -u
TAG=$1; SHORT=$2; WL=$3; STEPS=${4:-6}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
python bench.py --workload "$WL" --stages --no-cpu-baseline --no-air-program --steps "$STEPS" --warmup 1 > "$OUT/${SHORT}_bench.json" 2> "$OUT/${SHORT}_bench_stages.txt"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --workload $WL --no-cpu-baseline --no-air-program --concurrent 1 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${SHORT}_c1" -- python3 $B --steps 3 > "$OUT/${SHORT}_c1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/${SHORT}_ps" -- python3 $B --steps 1 > "$OUT/${SHORT}_ps.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${SHORT}_pf" -- python3 $B --steps 1 > "$OUT/${SHORT}_pf.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${SHORT}_pw" -- python3 $B --steps 1 > "$OUT/${SHORT}_pw.log" 2>&1
cd "$ROOT"
f=$(find "$OUT/${SHORT}_c1" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${SHORT}_kernel_stats_single_stream.csv"
f=$(find "$OUT/${SHORT}_ps" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python profiles/pmc_to_sq.py "$f" "$OUT/${SHORT}_sq_counters.csv" > /dev/null
pf=$(find "$OUT/${SHORT}_pf" -name "*counter_collection.csv" | head -1); pw=$(find "$OUT/${SHORT}_pw" -name "*counter_collection.csv" | head -1)
[ -n "$pf" ] && [ -n "$pw" ] && python profiles/pmc_to_traffic.py "$pf" "$pw" "$WL" "$OUT/${SHORT}_pmc_traffic.json" > /dev/null
rm -rf "$OUT/${SHORT}_c1" "$OUT/${SHORT}_ps" "$OUT/${SHORT}_pf" "$OUT/${SHORT}_pw"
python profiles/summarize_workload.py "$OUT" "$SHORT" "$WL" | tee "$OUT/${SHORT}_summary.txt"
