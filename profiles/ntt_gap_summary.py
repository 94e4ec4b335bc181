"""Per-kernel table from the passes of tools/ntt_gap.sh: counters summed over all launches of a kernel, divided by its waves.
usage: python profiles/ntt_gap_summary.py <dir with p*_counter_collection.csv and kernel_stats.csv>"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    return re.sub(r"^void ", "", name).split("(")[0].replace("aero::", "")


def main(d):
    acc = defaultdict(lambda: defaultdict(float))
    for f in sorted(glob.glob(os.path.join(d, "p*_counter_collection.csv"))):
        per_pass = defaultdict(lambda: defaultdict(float))
        for r in csv.DictReader(open(f)):
            per_pass[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, c in per_pass.items():
            waves = max(c.get("SQ_WAVES", 0.0), 1.0)
            for name, v in c.items():
                if name != "SQ_WAVES":
                    acc[k][name] = v / waves              # per wave, normalised inside its own pass
            acc[k]["waves_per_launch_set"] = waves
    dur = {}
    ks = os.path.join(d, "kernel_stats.csv")
    if os.path.exists(ks):
        for r in csv.DictReader(open(ks)):
            dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    names = sorted(set(n for c in acc.values() for n in c))
    prefixes = tuple(os.environ.get("GAP_PREFIX", "ntt_").split(","))     # GAP_PREFIX=hash_,merkle_: other kernel families (tools/kernel_gap.sh)
    kernels = [k for k in acc if k.startswith(prefixes)]
    print("per-wave counters (each from its own pass); avg_us from the kernel trace")
    for k in sorted(kernels, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0)):
        c = acc[k]
        print(f"\n== {k}   calls {dur.get(k, (0, 0))[0]}  avg_us {dur.get(k, (0, 0))[1]:.1f}")
        for n in names:
            if n in c:
                print(f"   {n:28s} {c[n]:14.1f}")
        wc = c.get("SQ_WAVE_CYCLES", 0)
        if wc:
            print(f"   -- VALU issue share of wave-cycles   {c.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}   (quad-cycles a wave spent issuing VALU / cycles it was resident)")
            print(f"   -- waiting on any instruction        {c.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}")
            print(f"   -- waiting on LDS                    {c.get('SQ_WAIT_INST_LDS', 0) / wc:.3f}")
            print(f"   -- VMEM issue cycles                 {(c.get('SQ_INST_CYCLES_VMEM_RD', 0) + c.get('SQ_INST_CYCLES_VMEM_WR', 0)) / wc:.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
