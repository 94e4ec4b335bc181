"""Interpreter vs hand-written constraint kernel on the same AIR: FibAir(width) proven through
aero_prove_fib_air (fib_constraints_kernel) and through aero_prove_air with the same AIR as an AEROAIR program
(air_constraints_kernel), per-kernel HIP-event times of the constraint stage.

    python tools/air_bench.py [--width 72] [--log-n 20] [--aux 0,0,2] [--ext 1] [--reps 5]
    python tools/air_bench.py --vm 26,9,16 [--log-n 20] [--fold 4]      # the VM-shaped program (aero_air_synth_vm_*), no hard-wired twin
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aero_amd  # noqa: E402


def kernel_ms(ctx, fn, names, reps):
    fn()
    ctx.set_kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    wall = (time.perf_counter() - t0) / reps * 1e3
    rep = ctx.kernel_timing_report()
    ctx.set_kernel_timing(False)
    out = {k: round(rep[k][1] / reps, 4) for k in names if k in rep}
    out["proof_wall_ms"] = round(wall, 3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=72)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--aux", default="0,0,2")
    ap.add_argument("--ext", type=int, default=1)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--vm", default="", help="pairs,aux,rands: prove the VM-shaped synthetic program instead")
    ap.add_argument("--fold", type=int, default=8)
    ap.add_argument("--table", action="store_true", help="--vm: also print the whole per-kernel table of one proof (HIP events) to stderr")
    a = ap.parse_args()
    if a.vm:
        pairs, A, R = (int(x) for x in a.vm.split(","))
        opt = aero_amd.ProofOptions(27, 8, 16, 4, a.ext, a.fold, 8)
        ctx = aero_amd.Context(0)
        air = aero_amd.Air(aero_amd.synth_vm_program(a.log_n, pairs, A, R))
        trace, pub = aero_amd.synth_vm_trace(a.log_n, pairs)
        dev = ctx.trace_upload(trace)
        proof = ctx.prove_air(air, dev, pub, opt)
        aero_amd.verify_air(proof, pub, air, expected_log_n=a.log_n)
        ms = kernel_ms(ctx, lambda: ctx.prove_air(air, dev, pub, opt), ["air_jit_kernel", "air_constraints_kernel", "air_aux_factors_kernel", "air_divide_kernel"], a.reps)
        host_ms = kernel_ms(ctx, lambda: ctx.prove_air(air, trace, pub, opt), [], a.reps)["proof_wall_ms"]
        if a.table:
            ctx.set_kernel_timing(True)
            ctx.prove_air(air, dev, pub, opt)
            for name, (c, m, b) in sorted(ctx.kernel_timing_report().items(), key=lambda kv: -kv[1][1]):
                print(f"  {name:28s} calls {c:5.0f}  ms {m:8.3f}  alg GB/s {(b / (m * 1e-3) / 1e9 if m > 0 else 0.0):8.1f}", file=sys.stderr)
            ctx.set_kernel_timing(False)
        print(json.dumps({"workload": f"synth_vm_2^{a.log_n}x({20 + 2 * pairs}+{A}aux)_fold{a.fold}" + ("_quadratic" if a.ext == 2 else ""), "ms": ms,
                          "proof_wall_ms_pageable_host_trace": host_ms, "program_info": air.info(), "proof_bytes": len(proof), "verified": True}))
        return
    aux = tuple(int(x) for x in a.aux.split(","))
    opt = aero_amd.ProofOptions(27, 8, 16, 4, a.ext, 8, 8)
    ctx = aero_amd.Context(0)
    dev = ctx.trace_upload(aero_amd.fib_trace(a.width, a.log_n))
    air = aero_amd.Air(aero_amd.fib_program(a.width, aux))
    want, pub = ctx.prove_fib_aux(dev, aux[0], aux[1], opt, aux_degree=aux[2])
    assert ctx.prove_air(air, dev, pub, opt) == want, "program proof differs from the hard-wired proof"
    hard = kernel_ms(ctx, lambda: ctx.prove_fib_aux(dev, aux[0], aux[1], opt, aux_degree=aux[2]), ["fib_constraints_kernel", "aux_columns_kernel"], a.reps)
    prog = kernel_ms(ctx, lambda: ctx.prove_air(air, dev, pub, opt), ["air_jit_kernel", "air_constraints_kernel", "air_aux_factors_kernel"], a.reps)
    res = {"workload": f"fib_2^{a.log_n}x{a.width}" + (f"+aux{aux}" if aux[0] else "") + ("_quadratic" if a.ext == 2 else ""),
           "hard_wired_ms": hard, "program_ms": prog, "program_info": air.info(),
           "evaluator": "compiled at run time (air_jit_kernel)" if "air_jit_kernel" in prog else "interpreter (air_constraints_kernel; AERO_AIR_JIT=0)",
           "constraint_kernel_ratio": round(prog.get("air_jit_kernel", prog.get("air_constraints_kernel", 0.0)) / hard["fib_constraints_kernel"], 3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
