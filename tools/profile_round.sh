#!/bin/bash
# Collects the evidence files of profiles/ on a GPU box: bench line + stages, rocprofv3 kernel stats (4 streams and 1 stream),
# two separate PMC passes (FETCH_SIZE, WRITE_SIZE). usage: tools/profile_round.sh <tag>   (run from the repo root through gpurun)
set -u
TAG=${1:-rX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
python bench.py --stages > "$OUT/bench.json" 2> "$OUT/bench_stages.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c4" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-air-program > "$OUT/c4.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c1" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-air-program --concurrent 1 > "$OUT/c1.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pf" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-air-program --concurrent 1 > "$OUT/pf.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pw" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-air-program --concurrent 1 > "$OUT/pw.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/ps" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-air-program --concurrent 1 > "$OUT/ps.log" 2>&1
cd "$ROOT"
for d in c4 c1; do f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${d}_kernel_stats.csv"; done
for d in pf pw ps; do f=$(find "$OUT/$d" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${d}_counter_collection.csv"; done
rm -rf "$OUT/c4" "$OUT/c1" "$OUT/pf" "$OUT/pw" "$OUT/ps"
ls -la "$OUT"
tail -c 600 "$OUT/bench.json"
