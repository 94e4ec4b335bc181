"""Prove one large trace on one GPU and verify it with the library's host verifier. usage: python tools/big_proof.py [log_n] [width]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aero_amd
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
width = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = aero_amd.Context(0)
dev = ctx.trace_upload(aero_amd.fib_trace(width, log_n))
opt = aero_amd.ProofOptions.with_96_bit_security()
ctx.prove_fib(dev, opt)
t0 = time.perf_counter()
proof, pub = ctx.prove_fib(dev, opt)
ms = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter()
aero_amd.verify_fib(proof, pub, (0, 0, 2), expected_log_n=log_n)
vms = (time.perf_counter() - t0) * 1e3
print(f"2^{log_n} x {width}: {ms:.1f} ms ({(width << log_n) / ms / 1e3:.0f} M cells/s), {len(proof)} proof bytes, verified on the host in {vms:.1f} ms, "
      f"device memory peak {ctx.memory_stats()[1] / 2**30:.1f} GiB")
