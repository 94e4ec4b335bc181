"""Stage-level entry points of the fork's split prover API (composition polynomial, DEEP composition, FRI layers and FRI
openings) against the oracle's intermediates, replaying the oracle's transcript with its own coin primitives."""
import struct

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


def sections(proof: bytes, n_trace_segments=1):
    """Byte ranges of the StarkProof sections."""
    off = 22
    (clen,) = struct.unpack_from("<H", proof, off)
    commitments = proof[off + 2:off + 2 + clen]
    off += 2 + clen
    for _ in range(2 * (n_trace_segments + 1)):          # values + paths per trace segment, then constraint queries
        (l,) = struct.unpack_from("<I", proof, off)
        off += 4 + l
    for _ in range(2):                                   # ood trace states, ood evaluations
        (l,) = struct.unpack_from("<H", proof, off)
        off += 2 + l
    fri = proof[off:len(proof) - 8]
    (nonce,) = struct.unpack_from("<Q", proof, len(proof) - 8)
    return commitments, fri, nonce


@pytest.mark.parametrize("log_n,width,opt", [(10, 2, [27, 8, 16, 4, 1, 8, 8]), (8, 4, [20, 8, 8, 4, 1, 4, 5]), (12, 2, [27, 8, 12, 4, 1, 8, 8])])
def test_composition_deep_fri_stages_match_oracle(ctx, oracle, log_n, width, opt):
    n, N, C = 1 << log_n, 8 << log_n, 2
    proof, pub, _ = oracle.prove_fib(width, log_n, opt, keep_artifacts=True)
    commitments, fri_bytes, nonce = sections(proof)
    roots = [commitments[32 * i:32 * i + 32] for i in range(len(commitments) // 32)]
    # ---- composition polynomial from the oracle's numerator columns
    ce = oracle.artifact("ce_cols", 3 * C * n).reshape(3, C * n)
    comp_polys = ctx.composition_poly_fib(ce, log_n)
    assert comp_polys.shape == (C, n)
    comp_lde = ctx.evaluate_columns_over(comp_polys, 3)
    want_clde = oracle.artifact("comp_lde", C * N).reshape(C, N)
    assert (comp_lde.download() == want_clde).all()
    # ---- replay the transcript up to the DEEP coefficients
    ood_cur, ood_next, ood_h = (oracle.artifact(k, 64) for k in ("ood_cur", "ood_next", "ood_h"))
    seed = oracle.coin_new(pub)
    seed = oracle.coin_reseed(seed, roots[0])
    seed = oracle.coin_reseed(seed, roots[1])
    z, _ = oracle.coin_draw(seed, 0)
    for v in (ood_cur, ood_next, ood_h):
        seed = oracle.coin_reseed(seed, oracle.hash_elements(v))
    ctr, coeffs = 0, []
    for _ in range(3 * width + C + 2):
        v, ctr = oracle.coin_draw(seed, ctr)
        coeffs.append(v)
    # ---- DEEP composition
    trace_lde = ctx.evaluate_columns_over(ctx.interpolate_columns(ctx.trace_upload(aero_amd.fib_trace(width, log_n))), 3)
    deep = ctx.deep_compose(trace_lde, comp_lde, 3, z, np.concatenate([ood_cur, ood_next]), ood_h, coeffs)
    assert deep.shape == (1, N)
    assert (deep.download()[0] == oracle.artifact("deep", N)).all()
    # ---- FRI commit phase: roots, coin state, openings = the FRI section of the oracle's proof
    o = aero_amd.ProofOptions(*opt)
    fri, got_roots, seed_out = ctx.fri_build_layers(deep, o, seed)
    assert got_roots == roots[2:]
    for r in roots[2:]:
        seed = oracle.coin_reseed(seed, r)
    assert seed_out == seed
    seed = oracle.coin_reseed_int(seed, nonce)
    positions, _ = oracle.coin_draw_integers(seed, 0, opt[0], N)
    assert fri.open(positions) == fri_bytes
    with pytest.raises(aero_amd.AeroError):
        fri.open([N])
    fri.free()


def test_stage_argument_checks(ctx):
    with pytest.raises(aero_amd.AeroError):
        ctx.composition_poly_fib(np.full((3, 16), P, np.uint64), 3)            # non-canonical element
    t = ctx.evaluate_columns_over(ctx.interpolate_columns(ctx.trace_upload(aero_amd.fib_trace(2, 6))), 3)
    with pytest.raises(aero_amd.AeroError):
        ctx.fri_build_layers(t, aero_amd.ProofOptions.with_96_bit_security(), bytes(32))   # 2 columns for a base-field FRI
