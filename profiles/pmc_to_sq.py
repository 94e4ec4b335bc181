"""Per-kernel SQ counter summary from one rocprofv3 PMC pass (`--pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU`,
`--kernel-trace` only; tools/profile_round.sh pass `ps`).

usage: python profiles/pmc_to_sq.py <ps_counter_collection.csv> <out.csv>
Counters are summed over all launches of a kernel; the derived columns are per wave and per wave-cycle."""
import csv
import re
import sys
from collections import defaultdict

COUNTERS = ["SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU"]


def short(kernel_name):
    return re.sub(r"^void ", "", kernel_name).split("(")[0].replace("aero::", "")


def main(src, out):
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for r in csv.DictReader(open(src)):
        if r["Counter_Name"] not in COUNTERS:
            continue
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    rows = []
    for k, c in acc.items():
        waves, cyc = max(c["SQ_WAVES"], 1.0), max(c["SQ_WAVE_CYCLES"], 1.0)
        rows.append([k, len(launches[k])] + [round(c[n], 3) for n in COUNTERS] +
                    [round(c["SQ_INSTS_VALU"] / waves, 1), round(c["SQ_WAVE_CYCLES"] / waves, 3), round(c["SQ_INSTS_VALU"] / cyc, 3)])
    rows.sort(key=lambda r: -r[4])
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches"] + COUNTERS + ["valu_insts_per_wave", "wave_cycles_per_wave", "valu_insts_per_wave_cycle"])
        w.writerows(rows)
    for r in rows[:10]:
        print(f"{r[0]:50s} launches {r[1]:4d}  VALU/wave {r[6]:9.1f}  VALU/wave-cycle {r[8]:.3f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
