"""Ablation of the AIR-program INTERPRETER on FibAir(72) (which instruction classes cost what): programs with / without transition
constraints, boundary assertions, fused EMIT forms. Run with AERO_AIR_JIT=0 (the default evaluator is the run-time compiled kernel).
usage: AERO_AIR_JIT=0 python tools/air_variants.py"""
import sys, json
sys.path.insert(0, '.')
import aero_amd
from aero_amd import air as A
W, log_n = 72, 20
opt = aero_amd.ProofOptions(27, 8, 16, 4, 1, 8, 8)
ctx = aero_amd.Context(0)
dev = ctx.trace_upload(aero_amd.fib_trace(W, log_n))
def build(trans=True, bounds=True, fused=True, ntr=None):
    b = A.AirBuilder(W, num_pub=W // 2)
    for k in range(W // 2):
        a, bb, na, nb = b.main(2 * k), b.main(2 * k + 1), b.main_next(2 * k), b.main_next(2 * k + 1)
        if trans and (ntr is None or k < ntr):
            b.transition(na - (a + bb), 1); b.transition(nb - (bb + na), 1)
    if not trans or ntr == 0:
        b.transition(b.main_next(0) - (b.main(0) + b.main(1)), 1)
    if bounds:
        for c in range(W): b.assert_single(c, 0, 1 + c)
        for k in range(W // 2): b.assert_single(2 * k + 1, -1, b.pub(k))
    else:
        b.assert_single(0, 0, 1)
    return b
def t(name, b):
    air = aero_amd.Air(b.to_bytes())
    pub = [1] * (W // 2)
    ctx.prove_air(air, dev, pub, opt)
    ctx.set_kernel_timing(True)
    for _ in range(3): ctx.prove_air(air, dev, pub, opt)
    r = ctx.kernel_timing_report(); ctx.set_kernel_timing(False)
    i = air.info()
    print(name, "instrs", i["instructions"], "ms", round(r["air_constraints_kernel"][1] / 3, 4))
t("full", build())
t("no_bounds", build(bounds=False))
t("no_trans", build(trans=False))
t("neither(1+1)", build(trans=False, bounds=False))
t("half_trans_no_bounds", build(bounds=False, ntr=18))
