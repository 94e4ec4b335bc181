"""The per-configuration evidence sets of profiles/ (round 6, tools/profile_workload.sh) are self-checking: the figures quoted in
profiles/README.md and BASELINE.md section 3 - dominant kernel, its algorithmic bytes / rocprof time / 8 TB/s, its VALU issue fraction, the
workload's path fraction - are recomputed here from the tracked CSV / JSON files by the same script that produced the summaries. (SURVEY 8(d):
`roofline.achieved` = algorithmic bytes per launch / the kernel's average launch duration; the rocprof summary must agree.)"""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")

# short name -> (workload, dominant kernel, its fraction of the HBM peak, its VALU issue fraction) as quoted in profiles/README.md
SETS = {
    "config2": ("fib_2^20x2_blowup8_blake2s_base", "merkle_leaf8_rows_kernel<2>", 0.053, 0.487),
    "config3": ("fib_2^20x2_blowup8_blake2s_quadratic", "merkle_leaf8_rows_kernel<4>", 0.059, 0.485),
    "config4": ("fib_2^24x2_blowup8_blake2s_base", "merkle_leaf8_rows_kernel<2>", 0.048, 0.441),
    "wide72": ("fib_2^20x72_blowup8_blake2s_base", "hash_rows_kernel<RowSrc>", 0.079, 0.464),
    "config5": ("standin_miden_shape_2^22x(72+9aux)_deg8_fold4", "hash_rows_kernel<RowSrc>", 0.087, 0.484),
    "config5vm": ("program_vm_shape_2^22x(72+9aux)_fold4", "hash_rows_kernel<RowSrc>", 0.087, 0.484),
}


@pytest.mark.parametrize("short", sorted(SETS))
def test_round6_evidence_set_reproduces_its_quoted_figures(short, tmp_path):
    workload, kernel, hbm, valu = SETS[short]
    for part in ("bench.json", "bench_stages.txt", "kernel_stats_single_stream.csv", "sq_counters.csv", "pmc_traffic.json"):
        shutil.copy(os.path.join(PROF, f"r6_{short}_{part}"), tmp_path / f"{short}_{part}")
    r = subprocess.run([sys.executable, os.path.join(PROF, "summarize_workload.py"), str(tmp_path), short, workload], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.load(open(tmp_path / f"{short}_summary.json"))
    dom = got["dominant"]
    assert dom["kernel"] == kernel
    assert abs(dom["hbm_frac"] - hbm) < 0.0015, dom
    assert abs(dom["valu_issue_frac"] - valu) < 0.0015, dom
    # the same numbers are what the tracked summary holds, and the bench line's own dominant kernel is this kernel's launch name
    kept = json.load(open(os.path.join(PROF, f"r6_{short}_summary.json")))
    assert kept["dominant"]["kernel"] == kernel and abs(kept["dominant"]["hbm_frac"] - dom["hbm_frac"]) < 1e-9
    bench = json.loads(open(os.path.join(PROF, f"r6_{short}_bench.json")).read().strip().splitlines()[-1])
    assert bench["config"]["workload"] == workload and bench["roofline"]["kernel"] == dom["launch"]
    # HIP events inside bench.py and the rocprofv3 trace agree on the dominant kernel's duration alone on the GPU (two separate runs of the
    # workload: within 6 %)
    ev = bench["roofline"]["one_proof_in_flight"]["avg_launch_us"]
    if len([k for k in got["kernels"] if k["launch"] == dom["launch"]]) == 1:
        assert abs(ev - dom["avg_ns"] / 1e3) / ev < 0.06, (ev, dom["avg_ns"])


def test_headline_evidence_of_the_round_is_consistent():
    bench = json.loads(open(os.path.join(PROF, "r6_bench.json")).read().strip().splitlines()[-1])
    assert bench["config"]["workload"] == "fib_2^20x2_blowup8_blake2s_base" and bench["n_gpus"] == 1
    cells = bench["config"]["trace_rows"] * bench["config"]["trace_cols"] * bench["config"]["traces_per_step_per_gpu"] * bench["steps"]
    assert abs(bench["value"] - cells / (bench["ms_per_step"] * 1e-3 * bench["steps"])) / bench["value"] < 1e-6
    rf = bench["roofline"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["unit"] == "GB/s"
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9) / rf["achieved"] < 1e-6
    assert 0.95 < rf["traffic"] / rf["algorithmic_bytes_per_launch"] < 1.10        # PMC traffic ~ algorithmic bytes: no wasted re-reads
    assert bench["cpu_baseline"]["kind"] == "port" and bench["cpu_baseline"]["cores"] >= 1
    assert bench["self_verify"]["in_timed_region"] is False
