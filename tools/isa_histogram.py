"""ISA op histogram of one kernel of libaero_stark.so's gfx950 code object (no GPU needed).

    python tools/isa_histogram.py hash.o 'merkle_leaf8_rows_kernelILi2E' [compressions_per_thread]

Disassembles the device code object bundled in the given object file (llvm-objdump from /opt/rocm), takes the named kernel
and prints its instruction mix: VALU ops by mnemonic with their encoding class (VOP2 = 4-byte encoding, VOP3 = 8-byte encoding,
which issues at about 0.62 of the VOP2 rate on this chip: tools/ubench_valu.hip), scalar and memory instructions, and - when
the number of BLAKE2s compressions a thread executes is given - the per-compression figures the DESIGN.md ceiling is built
from. The kernel is straight-line (fully unrolled), so static counts are dynamic counts per thread."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    tmp = tempfile.mkdtemp(prefix="isa_")
    import shutil
    shutil.copy(obj, os.path.join(tmp, "in.o"))      # the bundles are extracted next to the input
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "in.o"], cwd=tmp, stdout=subprocess.DEVNULL)
    co = [f for f in os.listdir(tmp) if "amdgcn" in f][0]
    return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, co)], text=True)


def main():
    obj, kernel = sys.argv[1], sys.argv[2]
    per_thread = float(sys.argv[3]) if len(sys.argv) > 3 else None
    text = disassemble(obj)
    m = re.search(r"^[0-9a-f]+ <([^>]*%s[^>]*)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)" % re.escape(kernel), text, re.S | re.M)
    if not m:
        raise SystemExit(f"kernel {kernel} not found")
    ops = collections.Counter()
    cls = collections.Counter()      # VALU by encoding class: vop2 (4 bytes), vop2_literal (4 + 4-byte literal), vop3 (8 bytes)
    branches = 0
    for line in m.group(2).splitlines():
        # "\tv_add_u32_e32 v1, v2, v3        // 000000001234: 68020702" -> mnemonic + encoding size
        mm = re.match(r"\s+(\S+)\s.*//\s*[0-9A-Fa-f]+:\s*((?:[0-9A-Fa-f]{8}\s*)+)", line)
        if not mm:
            continue
        op = mm.group(1)
        ops[op] += 1
        if op.startswith("v_"):
            nbytes = 4 * len(mm.group(2).split())
            e32 = op.endswith(("_e32", "_dpp", "_sdwa"))
            cls["vop2" if (e32 and nbytes == 4) else "vop2_literal" if e32 else "vop3"] += 1
        if op.startswith("s_cbranch") or op == "s_branch":
            branches += 1
    valu = {k: v for k, v in ops.items() if k.startswith("v_")}
    n_valu = sum(valu.values())
    n4 = cls["vop2"] + cls["vop2_literal"]
    n8 = cls["vop3"]
    out = {
        "kernel": m.group(1), "instructions": sum(ops.values()), "valu": n_valu, "valu_vop2": cls["vop2"], "valu_vop2_with_literal": cls["vop2_literal"],
        "valu_vop3": n8, "vop2_equivalents_at_0.62_rate_for_vop3": n4 + n8 / 0.62,
        "salu": sum(v for k, v in ops.items() if k.startswith("s_")), "vmem": sum(v for k, v in ops.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_"))),
        "lds": sum(v for k, v in ops.items() if k.startswith("ds_")), "branches": branches,
        "valu_by_mnemonic": dict(sorted(valu.items(), key=lambda kv: -kv[1])),
    }
    if per_thread:
        out["compressions_per_thread"] = per_thread
        out["per_compression"] = {"valu": n_valu / per_thread, "vop2": n4 / per_thread, "vop3": n8 / per_thread,
                                  "vop2_equivalents": (n4 + n8 / 0.62) / per_thread,
                                  "by_mnemonic": {k: round(v / per_thread, 1) for k, v in sorted(valu.items(), key=lambda kv: -kv[1])[:6]}}
        # ceiling implied by this instruction mix at the measured issue rates (tools/ubench_valu.hip: 56 T lane-ops/s for VOP2)
        out["implied_ceiling_Gcomp_per_s_at_56T_vop2_lane_ops"] = 56e12 / ((n4 + n8 / 0.62) / per_thread) / 1e9
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
