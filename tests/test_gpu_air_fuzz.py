"""Differential fuzzing of the three constraint evaluators (the run-time compiled kernel, the interpreter, the oracle's ProgramAir)
over RANDOM constraint programs: random expression DAGs over main / auxiliary / periodic columns, public inputs and random elements,
declared degrees equal to the computed ones, random single and periodic assertions, auxiliary running products with and without
denominators, 1 - 2 exemptions, both fields, several blowups and folding factors. The traces are random field elements - they do
NOT satisfy the constraints and the proofs do not verify; the prover does not validate a trace (release-mode Winterfell neither),
it evaluates, divides pointwise, interpolates and commits, and every byte of that must be identical in the three implementations.
(The accepted-proof direction is covered by tests/test_gpu_air.py on satisfiable systems.)"""
import os
import random

import numpy as np
import pytest

import aero_amd
from aero_amd import air as A

pytestmark = pytest.mark.gpu
P = A.P


class Gen:
    def __init__(self, seed):
        self.r = random.Random(seed)
        r = self.r
        self.W = r.randint(2, 6)
        self.Aw = r.choice([0, 0, 1, 2])
        self.R = r.randint(2, 3) if self.Aw else 0
        self.npub = r.randint(1, 2)
        self.b = A.AirBuilder(self.W, self.Aw, self.R, num_pub=self.npub, exemptions=r.randint(1, 2))
        self.periodic = []
        for _ in range(r.randint(0, 2)):
            cyc = r.choice([2, 4, 8])
            self.periodic.append((self.b.periodic([r.randrange(P) for _ in range(cyc)]), cyc))

    # (expression, polynomial degree in the trace columns, set of periodic columns used)
    def leaf(self, aux):
        r, b = self.r, self.b
        k = r.random()
        if k < 0.45:
            c = r.randrange(self.W)
            return (b.main(c) if r.random() < 0.6 else b.main_next(c)), 1, frozenset()
        if aux and k < 0.65:
            c = r.randrange(self.Aw)
            return (b.aux(c) if r.random() < 0.5 else b.aux_next(c)), 1, frozenset()
        if aux and k < 0.75:
            return b.rand(r.randrange(self.R)), 0, frozenset()
        if self.periodic and k < 0.85:
            i = r.randrange(len(self.periodic))
            return self.periodic[i][0], 0, frozenset([i])
        if k < 0.93:
            return b.const(r.choice([0, 1, 2, 3, P - 1, r.randrange(P)])), 0, frozenset()
        return b.pub(r.randrange(self.npub)), 0, frozenset()

    def expr(self, depth, aux, max_deg):
        if depth == 0 or self.r.random() < 0.25:
            return self.leaf(aux)
        x, dx, cx = self.expr(depth - 1, aux, max_deg)
        y, dy, cy = self.expr(depth - 1, aux, max_deg)
        op = self.r.choice("+-**")
        if op == "*" and dx + dy + len(cx | cy) <= max_deg:
            return x * y, dx + dy, cx | cy
        return (x + y if op == "+" else x - y), max(dx, dy), cx | cy

    def constraint(self, aux, max_deg):
        for _ in range(50):
            e, d, cyc = self.expr(self.r.randint(1, 4), aux, max_deg)
            if d >= 1 and d + len(cyc) <= max_deg and (e.ref >> 24) == A.NODE:
                return e, d, [self.periodic[i][1] for i in sorted(cyc)]
        c = self.r.randrange(self.W)
        return self.b.main_next(c) - self.b.main(c), 1, []

    def build(self, log_n, max_deg, v2=False):
        r, b, n = self.r, self.b, 1 << log_n
        for _ in range(r.randint(1, 4)):
            e, d, cyc = self.constraint(False, max_deg)
            b.transition(e, d, cycles=cyc)
        if self.Aw:
            for _ in range(r.randint(1, 2)):
                for _ in range(50):
                    e, d, cyc = self.constraint(True, max_deg)
                    break
                b.aux_transition(e, d, cycles=cyc)
        used = set()
        for _ in range(r.randint(1, 4)):
            col = r.randrange(self.W)
            if r.random() < 0.3 and log_n >= 4:
                stride = r.choice([2, 4, 8])
                key = (col, stride, r.randrange(stride))
                if key not in used:
                    used.add(key)
                    b.assert_periodic(col, key[2], stride, b.const(r.randrange(P)))
            else:
                step = r.choice([0, -1, -2, r.randrange(n)])
                key = (col, 0, step % n)
                if key not in used:
                    used.add(key)
                    b.assert_single(col, step, b.pub(r.randrange(self.npub)) if r.random() < 0.4 else b.const(r.randrange(P)))
        if v2:
            # Assertion::sequence on main (and auxiliary) columns: values at first + i * stride, as many as the trace length asks for
            for _ in range(r.randint(1, 3)):
                aux_col = self.Aw and r.random() < 0.3
                col = r.randrange(self.Aw if aux_col else self.W)
                stride = r.choice([2, 4, 8, n // 2])
                key = ("a" if aux_col else "m", col, stride, r.randrange(stride))
                if key in used or (not aux_col and (col, stride, key[3]) in used):
                    continue
                used.add(key)
                used.add((col, stride, key[3]))
                vals = [r.randrange(P) for _ in range(n // stride)]
                (b.aux_assert_sequence if aux_col else b.assert_sequence)(col, key[3], stride, vals)
        for c in range(self.Aw):
            init = b.const(1) if r.random() < 0.5 else b.rand(0) * b.rand(self.R - 1) + 1
            num = b.rand(r.randrange(self.R)) + b.main(r.randrange(self.W)) * (b.main_next(r.randrange(self.W)) if r.random() < 0.5 else 1)
            den = (b.rand(r.randrange(self.R)) + b.main(r.randrange(self.W))) if r.random() < 0.5 else None
            if v2 and r.random() < 0.45:             # general recurrence (host-built): non-affine in its own previous value, may read earlier aux columns
                other = b.aux(r.randrange(c + 1))
                b.aux_builder_general(c, init, b.aux(c) * other + b.main(r.randrange(self.W)) * b.rand(r.randrange(self.R)) + 1)
                b.aux_assert_single(c, 0, init)
                continue
            if v2 and r.random() < 0.7:              # affine builder: additive term, with or without its own denominator
                add = b.main(r.randrange(self.W)) * (b.rand(r.randrange(self.R)) if r.random() < 0.5 else 3)
                add_den = (b.rand(r.randrange(self.R)) + b.main_next(r.randrange(self.W))) if r.random() < 0.5 else None
                b.aux_builder(c, init, num if r.random() < 0.7 else 1, den, add, add_den)
                b.aux_assert_single(c, 0, init)
                continue
            b.aux_builder(c, init, num, den)
            b.aux_assert_single(c, 0, init)
            if r.random() < 0.3:
                b.aux_assert_single(c, -1, b.rand(0) + 5)
        return b.to_bytes()


@pytest.fixture(scope="module")
def ctxs():
    jit = aero_amd.Context(0)
    old = os.environ.get("AERO_AIR_JIT")
    os.environ["AERO_AIR_JIT"] = "0"
    try:
        interp = aero_amd.Context(0)
    finally:
        if old is None:
            del os.environ["AERO_AIR_JIT"]
        else:
            os.environ["AERO_AIR_JIT"] = old
    yield jit, interp
    jit.close()
    interp.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_programs_three_evaluators_one_proof(ctxs, oracle, seed):
    jit, interp = ctxs
    rng = random.Random(1000 + seed)
    log_n = rng.randint(4, 10)
    blowup = rng.choice([8, 8, 16])
    ext = rng.choice([1, 1, 2])
    opt = [rng.randint(3, 8), blowup, rng.randint(0, 4), 4, ext, rng.choice([2, 4, 8]), rng.randint(3, 6)]
    while True:                                    # Winterfell's FRI options: the remainder must not be smaller than the folding factor
        dom = blowup << log_n
        while dom > (1 << opt[6]):
            dom //= opt[5]
        if dom >= opt[5]:
            break
        opt[6] += 1
    g = Gen(seed)
    program = g.build(log_n, max_deg=blowup if blowup <= 8 else 8)
    air = aero_amd.Air(program)
    info = air.info()
    assert info["ce_blowup"] <= blowup
    nrng = np.random.default_rng(seed)
    trace = (nrng.integers(0, P, size=(g.W, 1 << log_n), dtype=np.uint64, endpoint=False))
    pub = [int(x) for x in nrng.integers(0, P, size=g.npub, dtype=np.uint64)]
    want, _ = oracle.prove_air(program, trace, pub, opt)
    got = jit.prove_air(air, trace, pub, aero_amd.ProofOptions(*opt))
    assert got == want, f"compiled kernel vs oracle: seed {seed} {info}"
    ref = interp.prove_air(air, interp.trace_upload(trace), pub, aero_amd.ProofOptions(*opt))          # resident trace on this one
    assert ref == want, f"interpreter vs oracle: seed {seed} {info}"


@pytest.mark.parametrize("seed", range(100, 116))
def test_random_version2_programs_three_evaluators_one_proof(ctxs, oracle, seed):
    """The same differential over AEROAIR version 2: sequence assertions (main and auxiliary, strides 2 .. n / 2) and affine auxiliary builders."""
    jit, interp = ctxs
    rng = random.Random(5000 + seed)
    log_n = rng.randint(4, 10)
    blowup = rng.choice([8, 8, 16])
    ext = rng.choice([1, 1, 2])
    opt = [rng.randint(3, 8), blowup, rng.randint(0, 4), 4, ext, rng.choice([2, 4, 8]), rng.randint(3, 6)]
    while True:
        dom = blowup << log_n
        while dom > (1 << opt[6]):
            dom //= opt[5]
        if dom >= opt[5]:
            break
        opt[6] += 1
    g = Gen(seed)
    if seed % 2 == 0 and not g.Aw:                 # half of the cases carry an auxiliary segment for sure
        g = Gen(seed + 1000)
    program = g.build(log_n, max_deg=blowup if blowup <= 8 else 8, v2=True)
    air = aero_amd.Air(program)
    nrng = np.random.default_rng(seed)
    trace = (nrng.integers(0, P, size=(g.W, 1 << log_n), dtype=np.uint64, endpoint=False))
    pub = [int(x) for x in nrng.integers(0, P, size=g.npub, dtype=np.uint64)]
    want, _ = oracle.prove_air(program, trace, pub, opt)
    got = jit.prove_air(air, trace, pub, aero_amd.ProofOptions(*opt))
    assert got == want, f"compiled kernel vs oracle: seed {seed} {air.info()}"
    ref = interp.prove_air(air, interp.trace_upload(trace), pub, aero_amd.ProofOptions(*opt))
    assert ref == want, f"interpreter vs oracle: seed {seed} {air.info()}"
