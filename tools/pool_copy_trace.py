"""Pool of host-trace proofs under rocprofv3 (--kernel-trace --memory-copy-trace): what the copies of a wide trace cost while other proofs run.
usage (on the GPU box): rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python3 tools/pool_copy_trace.py [log_n] [width] [slots] [rounds] [resident]
then: python3 tools/pool_copy_trace.py --summarise <dir>"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarise(d):
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    mc = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(kt))]
    t0, t1 = min(s for s, _ in ks), max(e for _, e in ks)
    print(f"kernels: {len(ks)} over {(t1 - t0) / 1e6:.1f} ms, sum of durations {sum(e - s for s, e in ks) / 1e6:.1f} ms")
    if mc:
        rows = list(csv.DictReader(open(mc[0])))
        big = [r for r in rows if int(r.get("Bytes", r.get("Size", 0)) or 0) >= (8 << 20)]
        print("copy columns:", list(rows[0].keys()) if rows else None)
        by = {}
        for r in big:
            k = (r.get("Direction", "?"), int(r.get("Bytes", r.get("Size", 0))))
            by.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        for k, v in sorted(by.items()):
            v.sort()
            print(f"  {k[0]:24s} {k[1] / 2**20:8.1f} MiB x {len(v):4d}: median {v[len(v) // 2]:7.2f} ms  min {v[0]:7.2f}  max {v[-1]:7.2f}  -> {k[1] / 1e9 / (v[len(v) // 2] * 1e-3):6.1f} GB/s at the median")
        iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in big)
        busy, cs, ce = 0, None, None
        for s, e in iv:
            if ce is None or s > ce:
                if ce is not None:
                    busy += ce - cs
                cs, ce = s, e
            else:
                ce = max(ce, e)
        if ce is not None:
            busy += ce - cs
        print(f"  link busy (union of copies >= 8 MiB): {busy / 1e6:.1f} ms = {100.0 * busy / (t1 - t0):.1f} % of the kernel span; sum of copy durations {sum(e - s for s, e in iv) / 1e6:.1f} ms")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
        sys.exit(0)
    import time
    import aero_amd
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 72
    slots = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    resident = len(sys.argv) > 5 and sys.argv[5] == "resident"
    pool = aero_amd.Pool(0, slots)
    trace = aero_amd.fib_trace(width, log_n)
    opt = aero_amd.ProofOptions.with_96_bit_security()
    hosts = [aero_amd.PinnedTrace(trace.copy()) for _ in range(slots)]
    devs = [pool.ctx(i).trace_upload(trace) for i in range(slots)]
    run = (lambda r: pool.prove_fib(devs, opt, (0, 0, 2), rounds=r)) if resident else (lambda r: pool.prove_fib_host(hosts, opt, (0, 0, 2), rounds=r))
    run(1)
    t = time.perf_counter()
    run(rounds)
    dt = time.perf_counter() - t
    print(f"{'resident' if resident else 'host'}: {slots * rounds} proofs of 2^{log_n} x {width} in {dt * 1e3:.1f} ms = {slots * rounds * (width << log_n) / dt / 1e9:.3f} G cells/s")
