#!/usr/bin/env python3
"""One row of profiles/README.md's per-configuration table from the files tools/profile_workload.sh leaves (round 6).

usage: summarize_workload.py <dir> <short> <workload>      reads <dir>/<short>_{bench.json, bench_stages.txt, kernel_stats_single_stream.csv,
sq_counters.csv, pmc_traffic.json}, writes <dir>/<short>_summary.json and prints the table.

Per kernel (rocprofv3 names, single stream, one proof in flight):
  time      rocprofv3 --kernel-trace --stats: calls, average ns, share of the kernel time
  alg       algorithmic bytes per launch = the library's own count (AERO_LAUNCH abytes; bench.py --stages: GB/s x ms per proof / calls per proof
            of the launch NAME; kernels sharing a name share it in proportion to their rocprof time)
  hbm_frac  alg bytes / average ns / 8 TB/s
  traffic   PMC FETCH_SIZE (x 2 on gfx950) + WRITE_SIZE per launch of the launch name, and traffic / alg
  valu      SQ_INSTS_VALU per launch / average ns / (256 CU x 4 SIMD x 2.4 GHz / 2 = 1228.8 G wave-instructions/s): the VALU issue fraction
  best      max(hbm_frac, valu): how close the kernel is to the nearer of its two roofs
The dominant kernel is the one with the largest share of time; "furthest below" is the kernel with the smallest `best` among those that
take at least 3 % of the kernel time."""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_to_traffic import launch_name

HBM = 8.0e12
VALU_NOMINAL = 256 * 4 * 2.4e9 / 2


def short(kernel_name):
    return re.sub(r"^void ", "", kernel_name).split("(")[0].replace("aero::", "")


def main(d, tag, workload):
    p = lambda s: os.path.join(d, f"{tag}_{s}")
    bench = json.loads(open(p("bench.json")).read().strip().splitlines()[-1])
    stage = {}
    for line in open(p("bench_stages.txt")):
        m = re.match(r"\s+(\S+)\s+calls/proof\s+([\d.]+)\s+ms/proof\s+([\d.]+)\s+alg GB/s\s+([\d.]+)", line)
        if m:
            stage[m.group(1)] = {"calls": float(m.group(2)), "ms": float(m.group(3)), "GBps": float(m.group(4))}
    stats = []
    for r in csv.DictReader(open(p("kernel_stats_single_stream.csv"))):
        if "aero::" not in r["Name"] and "air_jit" not in r["Name"] and "aero_" not in r["Name"]:
            continue
        stats.append({"kernel": short(r["Name"]), "launch": launch_name(r["Name"]), "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                      "total_ns": float(r["TotalDurationNs"])})
    tot = sum(s["total_ns"] for s in stats) or 1.0
    sq = {}
    if os.path.exists(p("sq_counters.csv")):
        for r in csv.DictReader(open(p("sq_counters.csv"))):
            sq[r["kernel"]] = {"valu_per_launch": float(r["SQ_INSTS_VALU"]) / max(int(r["launches"]), 1), "valu_per_wave": float(r["valu_insts_per_wave"])}
    traffic = {}
    if os.path.exists(p("pmc_traffic.json")):
        traffic = json.load(open(p("pmc_traffic.json"))).get(workload, {})
    by_launch = {}
    for s in stats:
        by_launch.setdefault(s["launch"], []).append(s)
    rows = []
    for s in stats:
        share = s["total_ns"] / tot
        st = stage.get(s["launch"])
        alg = None
        if st and st["GBps"] > 0:
            name_bytes_per_proof = st["GBps"] * 1e9 * st["ms"] * 1e-3                       # all launches of the name in one proof
            name_total = sum(x["total_ns"] for x in by_launch[s["launch"]]) or 1.0
            proofs = sum(x["calls"] for x in by_launch[s["launch"]]) / max(st["calls"], 1e-9)   # proofs in the trace
            alg = name_bytes_per_proof * proofs * (s["total_ns"] / name_total) / s["calls"]
        hbm_frac = alg / (s["avg_ns"] * 1e-9) / HBM if alg else None
        q = sq.get(s["kernel"])
        valu = q["valu_per_launch"] / (s["avg_ns"] * 1e-9) / VALU_NOMINAL if q else None
        tr = traffic.get(s["launch"], {}).get("hbm_bytes_per_launch")
        best = max([x for x in (hbm_frac, valu) if x is not None], default=None)
        rows.append({**s, "share": share, "alg_bytes_per_launch": alg, "hbm_frac": hbm_frac, "valu_issue_frac": valu, "pmc_bytes_per_launch_of_name": tr,
                     "best": best})
    rows.sort(key=lambda r: -r["share"])
    dom = rows[0]
    cand = [r for r in rows if r["share"] >= 0.03 and r["best"] is not None]
    worst = min(cand, key=lambda r: r["best"]) if cand else None
    pr = bench.get("path_roofline", {})
    out = {"workload": workload, "value_cells_per_s": bench["value"], "hbm_resident_value": bench.get("hbm_resident_value"),
           "single_proof_ms_resident": bench.get("single_proof_ms_hbm_resident"), "proofs_in_flight": bench["config"].get("proofs_in_flight_per_gpu"),
           "path_frac_of_hbm_peak": pr.get("frac_of_hbm_peak", pr.get("frac")), "path_bytes_per_cell": pr.get("bytes_per_cell"),
           "dominant": dom, "furthest_below": worst, "kernels": rows[:14]}
    json.dump(out, open(p("summary.json"), "w"), indent=1)
    f = lambda x, n=3: "-" if x is None else f"{x:.{n}f}"
    print(f"{workload}: {bench['value'] / 1e9:.3f} G cells/s ({f((bench.get('hbm_resident_value') or 0) / 1e9)} resident), path fraction of HBM {f(out['path_frac_of_hbm_peak'])}")
    print(f"{'kernel':44s} {'share':>6s} {'calls':>6s} {'avg us':>9s} {'alg MB':>9s} {'hbm':>6s} {'valu':>6s} {'pmc/alg':>7s}")
    for r in rows[:12]:
        ratio = (r["pmc_bytes_per_launch_of_name"] / r["alg_bytes_per_launch"]) if (r["pmc_bytes_per_launch_of_name"] and r["alg_bytes_per_launch"] and len(by_launch[r["launch"]]) == 1) else None
        print(f"{r['kernel'][:44]:44s} {100 * r['share']:6.1f} {r['calls']:6d} {r['avg_ns'] / 1e3:9.1f} {f((r['alg_bytes_per_launch'] or 0) / 1e6, 1):>9s} {f(r['hbm_frac']):>6s} {f(r['valu_issue_frac']):>6s} {f(ratio, 2):>7s}")
    print(f"dominant: {dom['kernel']} ({100 * dom['share']:.1f} % of kernel time), hbm {f(dom['hbm_frac'])}, valu {f(dom['valu_issue_frac'])}")
    if worst:
        print(f"furthest below its nearer roof (>= 3 % of time): {worst['kernel']} at {f(worst['best'])}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
