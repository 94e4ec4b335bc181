#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs with
--kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes).

Units and corrections applied (MI355X_MICROARCH.md "HBM" / cdna_hip_programming.md section 7):
  * FETCH_SIZE and WRITE_SIZE are reported in KiB: bytes = value * 1024;
  * on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (128-byte requests tallied
    at 64 bytes): the read side is doubled. WRITE_SIZE is uncalibrated in the guide and is taken as reported.
Output: profiles/pmc_traffic.json = {workload: {kernel: {hbm_bytes_per_launch, fetch_bytes_per_launch (corrected),
write_bytes_per_launch, launches}}} averaged over the steady-state launches of each kernel.
usage: pmc_to_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv WORKLOAD out.json"""
import csv
import json
import re
import sys
from collections import defaultdict


# device function -> the name its launches carry in bench.py's per-kernel table (AERO_LAUNCH name)
ALIASES = {
    "merkle_leaf8_rows_kernel": "merkle_leaf8_kernel",
    "hash_fri_rows_fixed_kernel": "hash_fri_rows_kernel", "hash_fri_rows_fixed2_kernel": "hash_fri_rows_kernel", "hash_rows_wide_kernel": "hash_rows_kernel",
    "ntt_fwd_strided_reg": "ntt_fwd_pass", "ntt_fwd_strided_reg6x2": "ntt_fwd_pass", "ntt_fwd_strided_reg6x2_v": "ntt_fwd_pass", "ntt_fwd_strided_reg6x2_buf": "ntt_fwd_pass", "ntt_fwd_strided_reg7x2": "ntt_fwd_pass", "ntt_inv_last_pass_11": "ntt_inv_pass", "ntt_inv_last_pass_12": "ntt_inv_pass", "ntt_fwd_first_pass": "ntt_fwd_pass", "ntt_fwd_first_pass_8": "ntt_fwd_pass",
    "merkle_multi_quad_kernel": "merkle_multi_kernel",
    "ntt_inv_strided_reg": "ntt_inv_pass",
    "merkle_up3_parts_kernel": "merkle_up3_kernel",
    "fri_fold_fft_kernel": "fri_fold_kernel", "eval_bitrev_multi_kernel": "eval_bitrev_kernel", "eval_ktab_multi_kernel": "eval_ktab_kernel",
    "aux_block_totals_kernel": "aux_columns_kernel", "aux_scan_totals_kernel": "aux_columns_kernel", "aux_apply_kernel": "aux_columns_kernel",
}


def launch_name(kernel_name):
    name = re.sub(r"^void ", "", kernel_name).split("(")[0].replace("aero::", "")
    base = re.sub(r"<.*", "", name)
    if base == "hash_rows_kernel" and "FriSrc" in name:
        return "hash_fri_rows_kernel"
    return ALIASES.get(base, base)


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc[launch_name(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main(fetch_csv, write_csv, workload, out):
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fv, wv = f.get(k, []), w.get(k, [])
        # drop the warm-up third of the launches (tables being built, first-touch)
        fv = fv[len(fv) // 3:] or fv
        wv = wv[len(wv) // 3:] or wv
        fb = 2.0 * 1024.0 * sum(fv) / max(len(fv), 1)
        wb = 1024.0 * sum(wv) / max(len(wv), 1)
        res[k] = {"hbm_bytes_per_launch": fb + wb, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                  "launches": len(fv)}
    try:
        allw = json.load(open(out))
    except Exception:
        allw = {}
    allw[workload] = res
    allw["_method"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE and, separately, --pmc WRITE_SIZE on "
                       "`bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent 1`; KiB -> bytes, FETCH_SIZE doubled "
                       "(gfx950 wide-read correction, MI355X_MICROARCH.md), averaged per launch over steady-state launches")
    json.dump(allw, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print(f"{k:28s} launches {v['launches']:4d}  fetch {v['fetch_bytes_per_launch'] / 1e6:9.2f} MB  write {v['write_bytes_per_launch'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main(*sys.argv[1:5])
