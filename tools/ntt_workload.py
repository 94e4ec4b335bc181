"""The NTT family alone, for counter collection (tools/ntt_gap.sh): interpolate + 8x LDE of `cols` columns of 2^log_n rows, `reps` times,
for each (log_n, cols) given as arguments "20x2 20x72 22x16".   usage: ntt_workload.py [reps] [shape ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aero_amd

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
shapes = sys.argv[2:] or ["20x2", "20x72"]
ctx = aero_amd.Context(0)
for sh in shapes:
    log_n, cols = (int(v) for v in sh.split("x"))
    dev = ctx.trace_upload(aero_amd.fib_trace(cols, log_n))
    for _ in range(reps):
        polys = ctx.interpolate_columns(dev)
        lde = ctx.evaluate_columns_over(polys, 3)
        lde.free()
        polys.free()
    ctx.synchronize()
    dev.free()
ctx.close()
print("ntt_workload done", shapes)
