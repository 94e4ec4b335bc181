"""Idle gaps inside ONE proof from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`, bench.py --concurrent 1).

usage: python tools/timeline_gaps.py <kernel_trace.csv> [min_gap_us]
Takes the last complete proof of the trace (from the kernel after the second-to-last `gather_addr_kernel` to the last one) and
prints every kernel with the idle time before it; the sum of the gaps is what host round trips and launch latency cost."""
import csv
import re
import sys


def short(n):
    return re.sub(r"^void ", "", n).split("(")[0].replace("aero::", "")


def main(path, min_gap=2.0):
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "gather_addr_kernel" in r["Kernel_Name"] or "openings_kernel" in r["Kernel_Name"]]
    if len(ends) < 2:
        raise SystemExit("need two proofs in the trace")
    seq = rows[ends[-2] + 1:ends[-1] + 1]
    t0 = int(seq[0]["Start_Timestamp"])
    prev_end = t0
    busy = gaps = 0.0
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3
        dur = (e - s) / 1e3
        busy += dur
        if gap > 0:
            gaps += gap
        flag = " <-- gap" if gap >= min_gap else ""
        print(f"{(s - t0) / 1e3:9.1f} us  gap {gap:7.1f}  dur {dur:7.1f}  {short(r['Kernel_Name'])[:60]}{flag}")
        prev_end = max(prev_end, e)
    print(f"kernels {len(seq)}  busy {busy:.1f} us  idle {gaps:.1f} us  span {(prev_end - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 2.0)
