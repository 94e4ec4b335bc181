"""GPU parity for the auxiliary trace segment (aero_prove_fib_aux): proof bytes identical to the CPU oracle's, accepted
by the oracle verifier including the OOD constraint check; aux column values against a direct prefix product."""
import numpy as np
import pytest

import aero_amd

pytestmark = pytest.mark.gpu
P = aero_amd.P


@pytest.fixture(scope="module")
def ctx():
    c = aero_amd.Context(0)
    yield c
    c.close()


def opts(**kw):
    o = aero_amd.ProofOptions.with_96_bit_security()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


CASES = [
    # log_n, width, aux_width, aux_rands, option overrides
    (3, 2, 1, 1, {"num_queries": 4, "grinding_factor": 0, "fri_folding_factor": 2, "fri_log_max_remainder": 3}),
    (8, 2, 1, 1, {}),
    (10, 2, 3, 2, {}),
    (9, 4, 9, 16, {"field_extension": 2}),                                   # 9 aux columns from 16 elements over F_p^2
    (8, 6, 2, 5, {"fri_folding_factor": 4, "num_queries": 20, "fri_log_max_remainder": 6}),
    (12, 2, 4, 4, {"blowup_factor": 16, "fri_folding_factor": 16, "fri_log_max_remainder": 4}),
    (12, 72, 9, 16, {"fri_folding_factor": 4, "num_queries": 16}),           # config-5 shape (72 + 9 columns, fold 4)
    (13, 4, 2, 3, {}),                                                       # more than one scan block per column
    (16, 2, 2, 2, {"field_extension": 2}),
]


@pytest.mark.parametrize("log_n,width,A,R,kw", CASES)
def test_aux_proof_bytes_identical_to_oracle(ctx, oracle, log_n, width, A, R, kw):
    o = opts(**kw)
    dev = ctx.trace_upload(aero_amd.fib_trace(width, log_n))
    got, pub = ctx.prove_fib_aux(dev, A, R, o)
    want, want_pub, _ = oracle.prove_fib_aux(width, log_n, A, R, o.to_list())
    assert pub == want_pub
    assert got == want, "proof bytes differ"
    oracle.verify_fib_aux(got, pub, width, log_n, A, R)
    again, _ = ctx.prove_fib_aux(dev, A, R, o)
    assert again == got
    dev.free()


DEGREE_CASES = [
    # log_n, width, aux_width, aux_rands, aux_degree, option overrides  (composition columns: 4 for degree 3-4, 8 for 5-8)
    (8, 2, 1, 1, 3, {}),
    (9, 2, 2, 3, 4, {"field_extension": 2}),
    (10, 4, 3, 4, 5, {}),
    (8, 2, 2, 2, 8, {"field_extension": 2}),
    (12, 72, 9, 16, 8, {"fri_folding_factor": 4, "num_queries": 16}),        # Miden's shape incl. 8 composition columns
    (14, 2, 2, 2, 8, {}),
    (7, 2, 1, 1, 4, {"blowup_factor": 16, "fri_folding_factor": 4, "fri_log_max_remainder": 5, "num_queries": 20, "grinding_factor": 8}),
]


@pytest.mark.parametrize("log_n,width,A,R,D,kw", DEGREE_CASES)
def test_aux_degree_proof_bytes_identical_to_oracle(ctx, oracle, log_n, width, A, R, D, kw):
    o = opts(**kw)
    dev = ctx.trace_upload(aero_amd.fib_trace(width, log_n))
    got, pub = ctx.prove_fib_aux(dev, A, R, o, aux_degree=D)
    want, want_pub, _ = oracle.prove_fib_aux(width, log_n, A, R, o.to_list(), D=D)
    assert pub == want_pub
    assert got == want, "proof bytes differ"
    oracle.verify_fib_aux(got, pub, width, log_n, A, R, D=D)
    aero_amd.verify_fib(got, pub, (A, R, D), min_query_security_bits=0)
    dev.free()


def test_aux_degree_needs_enough_blowup(ctx):
    dev = ctx.trace_upload(aero_amd.fib_trace(2, 8))
    with pytest.raises(aero_amd.AeroError):
        ctx.prove_fib_aux(dev, 1, 1, opts(blowup_factor=4, fri_folding_factor=4), aux_degree=8)   # needs 8 composition columns
    with pytest.raises(aero_amd.AeroError):
        ctx.prove_fib_aux(dev, 1, 1, opts(), aux_degree=9)
    dev.free()


def test_aux_zero_width_is_plain_proof(ctx):
    dev = ctx.trace_upload(aero_amd.fib_trace(2, 10))
    a, _ = ctx.prove_fib(dev, opts())
    b, _ = ctx.prove_fib_aux(dev, 0, 0, opts())
    assert a == b
    dev.free()


def test_aux_bad_shapes_fail_loudly(ctx):
    dev = ctx.trace_upload(aero_amd.fib_trace(2, 8))
    for A, R in [(3, 0), (254, 1), (1, 256)]:
        with pytest.raises(aero_amd.AeroError):
            ctx.prove_fib_aux(dev, A, R, opts())
    dev.free()


def test_full_size_aux_verifies(ctx, oracle):
    """2^20 rows (fused leaf kernel, unstored low Merkle levels, 512 scan blocks per aux column), 2 main + 2 aux columns:
    byte-identical to the oracle and verified with the OOD check."""
    log_n, W, A, R = 20, 2, 2, 2
    dev = ctx.trace_upload(aero_amd.fib_trace(W, log_n))
    got, pub = ctx.prove_fib_aux(dev, A, R, opts())
    oracle.verify_fib_aux(got, pub, W, log_n, A, R)
    want, _, _ = oracle.prove_fib_aux(W, log_n, A, R, opts().to_list())
    assert got == want
    dev.free()


EXTREMES = [
    # log_n, width, A, R, D, options
    (5, 254, 1, 1, 2, [16, 8, 4, 4, 1, 8, 5]),            # widest trace the one-byte width fields allow (254 + 1 columns)
    (10, 2, 0, 0, 2, [255, 8, 8, 4, 1, 8, 6]),            # most queries the one-byte count allows
    (9, 2, 1, 255, 3, [8, 8, 20, 4, 2, 4, 5]),            # 255 aux random elements, grinding 20, F_p^2
    (8, 4, 251, 7, 2, [8, 2, 0, 4, 1, 2, 4]),             # 251 aux columns, blowup 2 (= the number of composition columns)
    (3, 2, 1, 1, 8, [4, 8, 0, 4, 2, 2, 3]),               # 8-row trace with 8 composition columns over F_p^2
]


@pytest.mark.parametrize("log_n,width,A,R,D,o", EXTREMES)
def test_extreme_shapes(ctx, oracle, log_n, width, A, R, D, o):
    dev = ctx.trace_upload(aero_amd.fib_trace(width, log_n))
    got, pub = ctx.prove_fib_aux(dev, A, R, aero_amd.ProofOptions(*o), aux_degree=D)
    want, want_pub, _ = oracle.prove_fib_aux(width, log_n, A, R, o, D=D)
    assert pub == want_pub and got == want
    oracle.verify_fib_aux(got, pub, width, log_n, A, R, D=D)
    aero_amd.verify_fib(got, pub, (A, R, D), min_query_security_bits=0)
    dev.free()
