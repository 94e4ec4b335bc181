#!/bin/bash
# Rebuilds and runs the instruction-rate / field / BLAKE2s microbenchmarks on the GPU box and keeps their RAW output plus a parsed
# summary (the ceilings bench.py and DESIGN.md quote). usage: bash tools/ceilings.sh <tag>   -> gpurun_out/<tag>/ubench_*.txt, ceilings.json
set -u
TAG=${1:-rX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
for t in ubench_valu ubench_rates ubench_field bench_hash; do
  hipcc --offload-arch=gfx950 -O3 -I "$ROOT/aero_amd/csrc" "$ROOT/tools/$t.hip" -o "/tmp/$t" 2> "$OUT/$t.build.log" || { echo "build of $t failed"; cat "$OUT/$t.build.log"; continue; }
  { echo "# $t on $(rocminfo 2>/dev/null | grep -m1 'Marketing Name' | sed 's/.*: *//') / $(cat /opt/rocm/.info/version 2>/dev/null) / $(date -u +%FT%TZ)"; "/tmp/$t"; } > "$OUT/$t.txt" 2>&1
  rm -f "$OUT/$t.build.log"
done
# lazy-reduction bound for the NTT's register transforms (canonical vs uncanonicalised add)
{ echo "# ubench_dft on $(rocminfo 2>/dev/null | grep -m1 'Marketing Name' | sed 's/.*: *//') / $(cat /opt/rocm/.info/version 2>/dev/null) / $(date -u +%FT%TZ)";
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I "$ROOT/aero_amd/csrc" "$ROOT/tools/ubench_dft.hip" -o /tmp/ubench_dft && /tmp/ubench_dft;
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGL_LAZY_ADD_UNSAFE -I "$ROOT/aero_amd/csrc" "$ROOT/tools/ubench_dft.hip" -o /tmp/ubench_dft_lazy && /tmp/ubench_dft_lazy; } > "$OUT/ubench_dft.txt" 2>&1
python3 - "$OUT" <<'PY'
import json, re, sys, os
out = sys.argv[1]
def rows(name, unit):
    d = {}
    p = os.path.join(out, name + ".txt")
    if not os.path.exists(p): return d
    for line in open(p):
        m = re.match(r"(.+?)\s+([0-9.]+) ms\s+([0-9.]+) " + unit, line.rstrip())
        if m: d[m.group(1).strip()] = float(m.group(3))
    return d
hash_ = rows("bench_hash", "G compress/s")
rates = rows("ubench_rates", "Tlane-op/s")
valu = rows("ubench_valu", "Gop/s")
res = {"source": "tools/ceilings.sh (raw outputs next to this file)",
       "blake2s_Gcompress_per_s": hash_, "lane_op_rates_T_per_s": rates, "valu_Gop_per_s": valu,
       # the register-only BLAKE2s loops of ubench_valu.hip: `merge` = a 64-byte node block, `elems` = a block of two zero-padded elements
       "blake2s_in_register_ceiling_Gcomp_per_s": max(valu.get("blake2s merge (compr/s)", 0.0), valu.get("blake2s elems (compr/s)", 0.0)),
       "nominal_valu_lane_ops_per_s": 256 * 128 * 2.4e9}
json.dump(res, open(os.path.join(out, "ceilings.json"), "w"), indent=1)
print(json.dumps(res)[:600])
PY
