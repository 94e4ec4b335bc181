"""Seeded random sweep over the option space (blowup 2..128, folds 2/4/8/16, remainder sizes, query counts, grinding, both
fields, trace widths, auxiliary segments of every constraint degree): GPU proof bytes must equal the oracle's for every
valid combination, and the oracle verifier (with the OOD check) must accept them."""
import random

import pytest

import aero_amd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = aero_amd.Context(0)
    yield c
    c.close()


def composition_columns(A, D):
    d, e = (D if A else 1), 2
    while e < d:
        e <<= 1
    return e


def valid(log_n, W, A, R, D, o):
    q, B, g, _, ext, F, lr = o
    n, N = 1 << log_n, B << log_n
    if composition_columns(A, D) > B or N > (1 << 22):
        return False
    dom = N
    while dom > (1 << lr):
        dom //= F
    if dom < F or dom * 8 * (2 if ext == 2 else 1) > 0xFFFF:
        return False
    return q <= N // 4 and W + A <= 255


def configs(count, seed):
    rng = random.Random(seed)
    out = []
    while len(out) < count:
        log_n = rng.choice([3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
        W = 2 * rng.choice([1, 1, 1, 2, 3, 4, 8, 17, 36])
        A = rng.choice([0, 0, 1, 2, 3, 9])
        R = rng.choice([1, 2, 5, 16])
        D = rng.choice([2, 2, 3, 4, 5, 7, 8])
        o = [rng.choice([1, 4, 16, 27, 40, 64]), rng.choice([2, 4, 8, 8, 16, 32, 64, 128]), rng.choice([0, 4, 8, 12, 16]), 4,
             rng.choice([1, 1, 2]), rng.choice([2, 4, 8, 8, 16]), rng.choice([3, 4, 5, 6, 7, 8, 10])]
        if valid(log_n, W, A, R, D, o) and (W + A) * (o[1] << log_n) <= (1 << 21):
            out.append((log_n, W, A, R, D, o))
    return out


@pytest.mark.parametrize("log_n,W,A,R,D,o", configs(32, 20240607))
def test_random_configuration(ctx, oracle, log_n, W, A, R, D, o):
    dev = ctx.trace_upload(aero_amd.fib_trace(W, log_n))
    got, pub = ctx.prove_fib_aux(dev, A, R, aero_amd.ProofOptions(*o), aux_degree=D)
    want, want_pub, _ = oracle.prove_fib_aux(W, log_n, A, R, o, D=D)
    assert pub == want_pub
    assert got == want, "proof bytes differ"
    oracle.verify_fib_aux(got, pub, W, log_n, A, R, D=D)
    dev.free()
