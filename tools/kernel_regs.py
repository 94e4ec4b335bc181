"""Register / LDS footprint of the kernels of one object file's gfx950 code object (no GPU needed).

    python tools/kernel_regs.py aero_amd/csrc/ntt.o [name-substring]

Prints, per kernel: VGPRs, spilled VGPRs, SGPRs, static LDS bytes and the waves per SIMD that fit (512 VGPRs per SIMD lane in
steps of 8, 160 KiB of LDS per CU shared by its 4 SIMDs)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(obj):
    tmp = tempfile.mkdtemp(prefix="regs_")
    shutil.copy(obj, os.path.join(tmp, "in.o"))
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "in.o"], cwd=tmp, stdout=subprocess.DEVNULL)
    co = [f for f in os.listdir(tmp) if "amdgcn" in f][0]
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, co)], text=True)
    shutil.rmtree(tmp, ignore_errors=True)
    out = []
    for blk in notes.split("- .agpr_count:")[1:]:
        def g(key):
            m = re.search(r"\.%s:\s+(\S+)" % key, blk)
            return m.group(1) if m else "0"
        out.append({"name": g("name"), "vgpr": int(g("vgpr_count")), "spill": int(g("vgpr_spill_count")), "sgpr": int(g("sgpr_count")),
                    "lds": int(g("group_segment_fixed_size")), "wg": int(g("max_flat_workgroup_size"))})
    return out


def waves_per_simd(k):
    by_vgpr = min(8, 512 // max(8, (k["vgpr"] + 7) // 8 * 8))
    if k["lds"]:
        wg_waves = max(1, k["wg"] // 64)
        by_lds = (160 * 1024 // k["lds"]) * wg_waves / 4.0
        return min(by_vgpr, by_lds)
    return by_vgpr


if __name__ == "__main__":
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in kernels(sys.argv[1]):
        if pat in k["name"]:
            name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip() or k["name"]
            print(f"{name[:90]:90s} vgpr {k['vgpr']:4d} spill {k['spill']:3d} sgpr {k['sgpr']:4d} lds {k['lds']:6d} wg {k['wg']:5d} waves/SIMD {waves_per_simd(k):g}")
