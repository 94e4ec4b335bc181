"""Experiment: S contexts (one HIP stream each) proving concurrently on ONE GPU from S host threads."""
import sys, time, threading
sys.path.insert(0, '/root/repo')
import aero_amd
opt = aero_amd.ProofOptions.with_96_bit_security()
trace = aero_amd.fib_trace(2, 20)
for S in (1, 2, 4, 6, 8):
    ctxs = [aero_amd.Context(0) for _ in range(S)]
    devs = [c.trace_upload(trace) for c in ctxs]
    for c, d in zip(ctxs, devs):
        c.prove_fib(d, opt); c.prove_fib(d, opt)
    K = 20
    def work(i):
        for _ in range(K):
            ctxs[i].prove_fib(devs[i], opt)
    ths = [threading.Thread(target=work, args=(i,)) for i in range(S)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print(f"S={S}: {S*K} proofs in {dt*1e3:.1f} ms -> {dt*1e3/(S*K):.3f} ms/proof aggregate, {S*K*(1<<21)/dt/1e6:.0f} M cells/s, latency {dt*1e3/K:.2f} ms")
    for d in devs: d.free()
    for c in ctxs: c.close()
