"""GPU busy fraction from a rocprofv3 kernel trace CSV (union of kernel intervals / span) over the densest window.
usage: python tools/busy_fraction.py kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# steady-state window: the last 40 % of the trace by time (the timed region of bench.py sits at the end)
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
w0 = t0 + int(0.6 * (t1 - t0))
iv = [(max(s, w0), e, n) for s, e, n in iv if e > w0]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(e for _, e, _ in iv) - w0
tot = sum(e - s for s, e, _ in iv)
print(f"window {span / 1e6:.2f} ms, union busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), sum of kernel durations {tot / 1e6:.2f} ms (avg concurrency {tot / busy:.2f})")
