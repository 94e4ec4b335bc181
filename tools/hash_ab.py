"""Per-kernel HIP-event time of the row hashing kernels on the shapes the workloads commit to (A/B runs: another build of the library through AERO_LIB_PATH). Prints one JSON line: per shape the kernel's time and its BLAKE2s compressions per second against profiles/ceilings.json.
usage: hash_ab.py [rowsLog2xcols ...] [friLog2Rows:fold ...]      default: 23x72 23x8 23x9 23x18 fri20:8 fri23:4"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aero_amd

CEIL = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "ceilings.json")))["blake2s_in_register_ceiling_Gcomp_per_s"]
shapes = sys.argv[1:] or ["23x72", "23x8", "23x9", "23x18", "fri20:8", "fri23:4"]
ctx = aero_amd.Context(0)
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("AERO_HASH")}, "ceiling_Gcomp_per_s": CEIL}
P = 0xFFFFFFFF00000001
for sh in shapes:
    if sh.startswith("fri"):
        log_rows, fold = (int(v) for v in sh[3:].split(":"))
        rows = 1 << log_rows
        # a FRI layer through the stage entry point: evaluations of fold * rows points, one commit phase; the first layer's row hash is the launch measured
        rng = np.random.default_rng(5)
        vals = (rng.integers(0, 1 << 63, (1, rows * fold), dtype=np.uint64) % np.uint64(P)).astype(np.uint64)
        m = ctx.trace_upload(vals)
        opts = aero_amd.ProofOptions(27, 8, 16, 4, 1, fold, 8)
        reps = 5
        for _ in range(2):
            f = ctx.fri_build_layers(m, opts, b"\0" * 32); f[0].free()
        ctx.set_kernel_timing(True, "hash_fri_rows_kernel")
        for _ in range(reps):
            f = ctx.fri_build_layers(m, opts, b"\0" * 32); f[0].free()
        rep = ctx.kernel_timing_report()
        ctx.set_kernel_timing(False)
        calls, ms, _ = rep["hash_fri_rows_kernel"]
        # all layers' row hashes are in the sum: the first carries (fold - 1) / fold of the rows (geometric series)
        total_rows = 0
        r = rows
        n_l = calls // reps
        for _ in range(n_l):
            total_rows += r
            r //= fold
        comp = total_rows * fold // 2
        us = 1e3 * ms / reps
        out[sh] = {"launches_per_commit": n_l, "us_all_layers": round(us, 1), "Gcomp_per_s": round(comp / us / 1e3, 2), "of_ceiling": round(comp / us / 1e3 / CEIL, 3)}
        m.free()
        continue
    leaf = sh.startswith("leaf")          # "leaf27x2": the fused leaf + 3 levels kernel of narrow matrices (<= 4 columns)
    kname = "merkle_leaf8_kernel" if leaf else "hash_rows_kernel"
    log_rows, cols = (int(v) for v in sh[4 if leaf else 0:].split("x"))
    rows = 1 << log_rows
    base = aero_amd.fib_trace(2, min(log_rows, 20))
    col = np.tile(base[0], rows // base.shape[1])
    big = np.empty((cols, rows), np.uint64)       # values do not matter for the rate
    for c in range(cols):
        big[c] = col
        big[c, 0] = c
    m = ctx.trace_upload(big)
    del big
    reps = 5
    for _ in range(2):
        ctx.merkle_commit_rows(m).free()          # wide matrices: hash_rows_kernel + the tree build (only the former is timed)
    ctx.set_kernel_timing(True, kname)
    for _ in range(reps):
        ctx.merkle_commit_rows(m).free()
    rep = ctx.kernel_timing_report()
    ctx.set_kernel_timing(False)
    calls, ms, _ = rep[kname]
    us = 1e3 * ms / calls
    comp = rows * ((cols + 1) // 2) + (rows * 7 // 8 if leaf else 0)      # the fused kernel also builds the 3 levels above its 8 leaves
    out[sh] = {"us": round(us, 1), "Gcomp_per_s": round(comp / us / 1e3, 2), "of_ceiling": round(comp / us / 1e3 / CEIL, 3),
               "GBps_algorithmic": round(rows * (cols * 8 + 32) / us / 1e3, 1)}
    m.free()
print(json.dumps(out))
