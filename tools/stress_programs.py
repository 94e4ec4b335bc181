"""Stress of the round-3 host machinery: pools of program proofs (host and resident traces), thread-rank sharded proofs through the
local group, trace validation - repeated, every proof compared with the first one of its kind. usage: python tools/stress_programs.py [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aero_amd
from aero_amd.shard import LocalGroup

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
log_n, pairs, A, R = 14, 6, 4, 5
program = aero_amd.synth_vm_program(log_n, pairs, A, R)
trace, pub = aero_amd.synth_vm_trace(log_n, pairs)
air = aero_amd.Air(program)
opt = aero_amd.ProofOptions(27, 8, 16, 4, 1, 4, 7)
ctx = aero_amd.Context(0)
want = ctx.prove_air(air, trace, pub, opt)
aero_amd.verify_air(want, pub, air, expected_log_n=log_n)
assert ctx.validate_trace(air, ctx.trace_upload(trace), pub) is None
pool = aero_amd.Pool(0, 4)
devs = [pool.ctx(i).trace_upload(trace) for i in range(4)]
pinned = [aero_amd.PinnedTrace(trace.copy()) for _ in range(4)]
fib = aero_amd.fib_trace(8, 15)
fopt = aero_amd.ProofOptions.with_96_bit_security()
fwant = ctx.prove_fib_aux(ctx.trace_upload(fib), 3, 4, fopt, aux_degree=5)[0]
fpinned = [aero_amd.PinnedTrace(fib.copy()) for _ in range(4)]
t0 = time.time()
for it in range(iters):
    assert pool.prove_air(air, devs, pub, opt, rounds=2) == [want] * 4
    assert pool.prove_air(air, pinned, pub, opt, rounds=2) == [want] * 4
    assert [p for p, _ in pool.prove_fib_host(fpinned, fopt, (3, 4, 5), rounds=2)] == [fwant] * 4
    for world in (2, 4, 8):
        g = LocalGroup(world, min_peer_digests=64)
        try:
            proofs = g.run(lambda r, c, comm: c.prove_air(air, pinned[0], pub, opt, comm=comm))
        finally:
            g.close()
        assert proofs == [want] * world, f"iteration {it}, world {world}"
print(f"{iters} iterations ok in {time.time() - t0:.1f} s: pools (resident, host, built-in AIR) and thread-rank sharded proofs of a program AIR are deterministic")
