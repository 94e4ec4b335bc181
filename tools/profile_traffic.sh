#!/bin/bash
# PMC traffic (two separate passes: FETCH_SIZE, WRITE_SIZE) of one bench workload -> profiles/pmc_traffic.json.
# usage: tools/profile_traffic.sh <tag> <workload>   (run from the repo root through gpurun)
set -u
TAG=${1:-rX}
WL=${2:-fib_2^20x72_blowup8_blake2s_base}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pf" -- python3 "$ROOT/bench.py" --workload "$WL" --steps 1 --warmup 1 --no-cpu-baseline --concurrent 1 > "$OUT/pf.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pw" -- python3 "$ROOT/bench.py" --workload "$WL" --steps 1 --warmup 1 --no-cpu-baseline --concurrent 1 > "$OUT/pw.log" 2>&1
cd "$ROOT"
for d in pf pw; do f=$(find "$OUT/$d" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${d}_counter_collection.csv"; done
rm -rf "$OUT/pf" "$OUT/pw"
python profiles/pmc_to_traffic.py "$OUT/pf_counter_collection.csv" "$OUT/pw_counter_collection.csv" "$WL" "$OUT/pmc_traffic_$TAG.json"
