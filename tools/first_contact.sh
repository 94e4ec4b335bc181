#!/bin/bash
# First contact with a multi-GPU node: everything that has never run with more than one GPU, in the order
# in which a failure is cheapest to understand, each step's output kept. No speed claims are made from a 1-GPU box: there the script is run
# with --share-gpu to exercise its control flow only (N ranks share cuda:0, exchanges over gloo).
#   usage: bash tools/first_contact.sh [--share-gpu] [out_dir]          (from the repo root; takes 10 - 20 minutes on 8 GPUs)
# Steps: 0 inventory (GPUs, their NUMA nodes, librccl)  1 two ranks over native RCCL = the single-GPU proof  2 replicas at 1/2/4/8 GPUs
# (the driver's scaling line)  3 ONE proof over all GPUs, configs 4 and 5, exchange in 1 / 4 / 16 pieces per peer - self-verify is on for
# every sharded proof (aero_ctx_set_self_verify AUTO), the line names how many ranks RCCL counted and where each rank's host side sits.
set -u
SHARE=""
if [ "${1:-}" = "--share-gpu" ]; then SHARE="--share-gpu"; shift; fi
OUT=${1:-gpurun_out/first_contact}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $NG  (share-gpu: ${SHARE:-no})" | tee "$OUT/summary.txt"
python3 - > "$OUT/0_inventory.txt" 2>&1 <<'PY'
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import aero_amd
lib = aero_amd.lib()
n = aero_amd.device_count()
print("devices:", n)
for d in range(n):
    node = C.c_int32(-1)
    lib.aero_numa_device_node(C.c_int32(d), C.byref(node))
    print(f"  device {d}: NUMA node {node.value}")
rc = lib.aero_rccl_available()
lib.aero_rccl_last_error.restype = C.c_char_p
print("librccl:", "bound" if rc == 0 else "NOT available: " + lib.aero_rccl_last_error(None).decode())
PY
cat "$OUT/0_inventory.txt" | tee -a "$OUT/summary.txt"
WORLD=$NG; [ -n "$SHARE" ] && WORLD=8
if [ "$NG" -ge 2 ]; then
  timeout 900 python -m pytest "tests/test_gpu_rccl.py::test_two_ranks_over_rccl_give_the_single_gpu_proof" -x -q -m gpu -p no:cacheprovider > "$OUT/1_rccl_two_ranks.log" 2>&1
  echo "1 two ranks over RCCL: rc=$? $(tail -1 "$OUT/1_rccl_two_ranks.log")" | tee -a "$OUT/summary.txt"
else
  echo "1 two ranks over RCCL: skipped (one GPU)" | tee -a "$OUT/summary.txt"
fi
for n in 1 2 4 8; do
  [ "$n" -gt "$WORLD" ] && continue
  timeout 900 python bench.py --gpus $n $SHARE --no-cpu-baseline --no-air-program --steps 10 --warmup 1 > "$OUT/2_replicas_$n.json" 2> "$OUT/2_replicas_$n.err"
  echo "2 replicas on $n GPU(s): rc=$? $(python3 -c "import json,sys; d=json.loads(open('$OUT/2_replicas_$n.json').read().strip().splitlines()[-1]); print('%.3f G cells/s' % (d['value']/1e9))" 2>/dev/null)" | tee -a "$OUT/summary.txt"
done
if [ "$WORLD" -ge 2 ]; then
  for wl in "fib_2^24x2_blowup8_blake2s_base" "standin_miden_shape_2^22x(72+9aux)_deg8_fold4"; do
    short=$(echo "$wl" | cut -c1-12 | tr -c 'a-zA-Z0-9\n' '_')
    for ch in 1 4 16; do
      f="$OUT/3_sharded_${short}_chunks$ch.json"
      timeout 1200 python bench.py --gpus $WORLD $SHARE --mode sharded --workload "$wl" --exchange-chunks $ch --steps 3 --warmup 1 > "$f" 2> "${f%.json}.err"
      echo "3 sharded $wl over $WORLD ranks, $ch piece(s): rc=$? $(python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d["sharded_proof"]
    ranks = s.get("ranks") or []
    counted = sorted({(r.get("rccl") or {}).get("ranks_counted_by_rccl") for r in ranks if r.get("rccl")}) or "n/a (ranks share one GPU: exchanges over gloo)"
    print("%.1f ms per proof, %.2fx one GPU, identical on every rank: %s, RCCL counted %s ranks, NUMA nodes %s, CPUs bound %s" % (
        s["ms_per_proof"], s["speedup_vs_single_gpu"], s["proof_identical_to_single_gpu_on_every_rank"], counted,
        [r["gpu_numa_node"] for r in ranks], [r["cpus_bound_to"] for r in ranks]))
except Exception as e:
    print("no result line (%s)" % e)
PY
)" | tee -a "$OUT/summary.txt"
    done
  done
fi
echo "kept in $OUT" | tee -a "$OUT/summary.txt"
