"""Per-kernel HIP-event time of interpolate + 8x LDE for the shapes given ("20x2 20x72"), for A/B runs of NTT variants selected
through the environment (AERO_NTT_R128, AERO_NTT_2PHASE, ...) or a whole other build (AERO_LIB_PATH=<libaero_stark_<variant>.so>, `make VARIANT=...`). Prints one JSON line.   usage: ntt_ab.py [shape ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aero_amd

shapes = sys.argv[1:] or ["20x2", "20x72"]
os.environ.setdefault("AERO_NTT_NAMES", "1")
ctx = aero_amd.Context(0)
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("AERO_") and k not in ("AERO_CRASH_TRACE",)}}
for sh in shapes:
    log_n, cols = (int(v) for v in sh.split("x"))
    dev = ctx.trace_upload(aero_amd.fib_trace(cols + (cols & 1), log_n)[:cols])      # odd widths: one column less of the next even width
    for _ in range(3):
        p = ctx.interpolate_columns(dev); l = ctx.evaluate_columns_over(p, 3); l.free(); p.free()
    reps = 10
    ctx.set_kernel_timing(True)
    for _ in range(reps):
        p = ctx.interpolate_columns(dev); l = ctx.evaluate_columns_over(p, 3); l.free(); p.free()
    rep = ctx.kernel_timing_report()
    ctx.set_kernel_timing(False)
    out[sh] = {k: round(1e3 * ms / reps, 1) for k, (c, ms, b) in rep.items()}
    out[sh]["total_us"] = round(sum(1e3 * ms / reps for (c, ms, b) in rep.values()), 1)
    dev.free()
print(json.dumps(out))
