"""Time the LDE of `cols` columns 2^log_n -> 2^(log_n+3) per kernel (HIP events). usage: python tools/time_lde.py [log_n] [cols]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aero_amd
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = aero_amd.Context(0)
polys = ctx.interpolate_columns(ctx.trace_upload(aero_amd.fib_trace(cols, log_n)))
for _ in range(3):
    ctx.evaluate_columns_over(polys, 3).free()
ctx.set_kernel_timing(True)
reps = 10
for _ in range(reps):
    ctx.evaluate_columns_over(polys, 3).free()
rep = ctx.kernel_timing_report()
tot = 0.0
for k, (c, ms, b) in rep.items():
    print(f"{k:24s} launches {c // reps}  us/LDE {1e3 * ms / reps:9.1f}")
    tot += ms
print(f"total us/LDE {1e3 * tot / reps:.1f}")
