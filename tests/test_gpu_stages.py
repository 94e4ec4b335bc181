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


def bitrev(x, bits):
    return int(format(x, f"0{bits}b")[::-1], 2) if bits else 0


@pytest.mark.parametrize("log_n,width,A,R,D,ext", [(8, 2, 2, 3, 2, 1), (9, 4, 3, 2, 5, 1), (8, 2, 2, 2, 8, 2), (7, 72, 9, 16, 8, 1), (8, 4, 1, 1, 3, 2)])
def test_aux_segment_stage_entry_points_match_oracle(ctx, oracle, log_n, width, A, R, D, ext):
    """The AIR-specific stages with the auxiliary segment, one by one against the intermediates of the oracle's own prover run:
    aux columns (the step of commit_to_trace_and_validate, proving_worker.rs:323-332), their LDE, the constraint seam with
    aux_rand_elements (ConstraintComputeWorkItem, utils.rs:302-347; 1 and 4 fragments) and the composition polynomial with 2 / 4 / 8
    columns."""
    opt = [27, 8, 16, 4, ext, 8, 5]
    deg = 2 if ext == 2 else 1
    n, N = 1 << log_n, 8 << log_n
    Cc = 2 if D <= 2 else (4 if D <= 4 else 8)
    proof, pub, _ = oracle.prove_fib_aux(width, log_n, A, R, opt, D=D, keep_artifacts=True)
    rands = oracle.artifact("aux_rands", R * deg)
    assert rands.size == R * deg
    dev = ctx.trace_upload(aero_amd.fib_trace(width, log_n))
    aux = ctx.aux_columns_fib(dev, (A, R, D), rands, ext)
    assert (aux.download() == oracle.artifact("aux_cols", A * deg * n).reshape(A * deg, n)).all()
    aux_lde = ctx.evaluate_columns_over(ctx.interpolate_columns(aux), 3)
    assert (aux_lde.download() == oracle.artifact("aux_lde", A * deg * N).reshape(A * deg, N)).all()
    lde = ctx.evaluate_columns_over(ctx.interpolate_columns(dev), 3)
    ncoef = 2 * deg * ((width + A) + (width + width // 2 + A))
    coeffs = oracle.artifact("cons_coeffs", ncoef)
    assert coeffs.size == ncoef
    ce_n = Cc * n
    want = oracle.artifact("ce_cols", 3 * deg * ce_n).reshape(3 * deg, ce_n)
    fi, cols = ctx.eval_constraints_air(lde, aux_lde, (A, R, D), 3, pub, rands, coeffs, field_extension=ext)
    assert fi == 0 and (cols == want).all()
    parts = np.zeros_like(cols)
    for k in range(4):
        fi, part = ctx.eval_constraints_air(lde, aux_lde, (A, R, D), 3, pub, rands, coeffs, field_extension=ext, fragment_offset=k, num_fragments=4)
        assert fi == k * (ce_n // 4)
        parts[:, fi:fi + part.shape[1]] = part
    assert (parts == want).all()
    comp = ctx.composition_poly_air(cols, log_n, Cc, ext)
    assert comp.shape == (Cc * deg, n)
    got = ctx.evaluate_columns_over(comp, 3).download()
    want_clde = oracle.artifact("comp_lde", Cc * deg * N).reshape(Cc * deg, N)          # oracle: column c * deg + d
    lc = Cc.bit_length() - 1
    for q in range(Cc):
        for d in range(deg):
            assert (got[d * Cc + q] == want_clde[bitrev(q, lc) * deg + d]).all(), (q, d)
    # without an aux segment the generalised entry points are the plain ones
    with pytest.raises(aero_amd.AeroError):
        ctx.eval_constraints_air(lde, None, (A, R, D), 3, pub, rands, coeffs, field_extension=ext)     # aux LDE missing
    with pytest.raises(aero_amd.AeroError):
        ctx.composition_poly_air(cols, log_n, 3, ext)


def test_stage_argument_checks(ctx):
    with pytest.raises(aero_amd.AeroError):
        ctx.composition_poly_fib(np.full((3, 16), P, np.uint64), 3)            # non-canonical element
    t = ctx.evaluate_columns_over(ctx.interpolate_columns(ctx.trace_upload(aero_amd.fib_trace(2, 6))), 3)
    with pytest.raises(aero_amd.AeroError):
        ctx.fri_build_layers(t, aero_amd.ProofOptions.with_96_bit_security(), bytes(32))   # 2 columns for a base-field FRI
