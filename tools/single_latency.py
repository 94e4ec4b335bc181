"""Wall-clock of ONE proof with nothing else in flight: median / min of `reps` runs, trace resident in HBM and (argument 4 = "host") handed
over in pinned host memory - the two figures of bench.py's `single_proof_ms_*`. AERO_HOST_GAPS=1 adds the library's per-proof host marks.
usage: python tools/single_latency.py [log_n] [width] [reps] [resident|host|both]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aero_amd
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
what = sys.argv[4] if len(sys.argv) > 4 else "resident"
ctx = aero_amd.Context(0)
opts = aero_amd.ProofOptions.with_96_bit_security()
host = aero_amd.fib_trace(W, log_n)
srcs = {}
if what in ("resident", "both"):
    srcs["resident"] = ctx.trace_upload(host)
if what in ("host", "both"):
    srcs["pinned host"] = aero_amd.PinnedTrace(host)
for name, trace in srcs.items():
    for _ in range(5):
        ctx.prove_fib(trace, opts)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.prove_fib(trace, opts)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f"single proof 2^{log_n} x {W} ({name}): median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f}  p90 {ts[int(len(ts) * 0.9)]:.4f}")
