import sys, time; sys.path.insert(0, '.')
import aero_amd
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
S, rounds = 3, 3
opt = aero_amd.ProofOptions(27, 8, 16, 4, 1, 4, 8)
pool = aero_amd.Pool(0, S)
# stand-in
tr = aero_amd.fib_trace(72, log_n)
hosts = [aero_amd.PinnedTrace(tr.copy()) for _ in range(S)]
devs = [pool.ctx(i).trace_upload(tr) for i in range(S)]
for name, fn in [("standin host", lambda r: pool.prove_fib_host(hosts, opt, (9, 16, 8), rounds=r)), ("standin resident", lambda r: pool.prove_fib(devs, opt, (9, 16, 8), rounds=r))]:
    fn(1); t0 = time.perf_counter(); fn(rounds); dt = time.perf_counter() - t0
    print(name, round(dt / rounds * 1e3, 1), "ms per round of", S)
for d in devs: d.free()
for h in hosts: h.release()
air = aero_amd.Air(aero_amd.synth_vm_program(log_n, 26, 9, 16))
vt, pub = aero_amd.synth_vm_trace(log_n, 26)
hosts = [aero_amd.PinnedTrace(vt.copy()) for _ in range(S)]
devs = [pool.ctx(i).trace_upload(vt) for i in range(S)]
for name, fn in [("program host", lambda r: pool.prove_air(air, hosts, pub, opt, rounds=r)), ("program resident", lambda r: pool.prove_air(air, devs, pub, opt, rounds=r)),
                 ("program host (unpinned ndarray)", lambda r: pool.prove_air(air, [vt] * S, pub, opt, rounds=r))]:
    fn(1); t0 = time.perf_counter(); fn(rounds); dt = time.perf_counter() - t0
    print(name, round(dt / rounds * 1e3, 1), "ms per round of", S)
