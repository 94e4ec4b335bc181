"""AIR-as-data on the GPU (include/aero_air.h): a constraint program interpreted per row by the device replaces the hard-wired
FibAir kernel. Reference seam: `ProcessorAir::new` -> `ConstraintEvaluator::new` -> `evaluate_fragment`
(aero-sdk/miden-wasm/src/constraints_worker.rs:32-59, proving_worker.rs:374-437).

Parity (bit-exact, through the C ABI):
  (a) FibAir and the stand-in AIR re-expressed as programs give the SAME proof bytes as the hard-wired kernels and as the oracle,
      both fields, 2^5 ... 2^20 rows;
  (b) a VM-shaped synthetic AIR (periodic columns, >= 8 degree groups up to degree 8, interior and periodic assertions, two
      exemptions, auxiliary running products with denominators; > 49 transition constraints) is GPU == oracle bytes and accepted
      by both verifiers with the out-of-domain constraint check;
  (c) the stage entry points (numerator fragments, auxiliary columns, composition polynomial) against the oracle's intermediates."""
import numpy as np
import pytest

import aero_amd
from aero_amd import air as A
from tests import air_examples as ex

pytestmark = pytest.mark.gpu

OPT = [27, 8, 16, 4, 1, 8, 8]


def options(o):
    return aero_amd.ProofOptions(*o)


@pytest.fixture(scope="module")
def ctx():
    c = aero_amd.Context(0)
    yield c
    c.close()


# ---- the program as a run-time compiled kernel (air_jit.hip, the default) and as interpreted code (AERO_AIR_JIT=0) ----------------
@pytest.fixture(scope="module")
def ctx_interp():
    import os
    old = os.environ.get("AERO_AIR_JIT")
    os.environ["AERO_AIR_JIT"] = "0"
    try:
        c = aero_amd.Context(0)
    finally:
        if old is None:
            del os.environ["AERO_AIR_JIT"]
        else:
            os.environ["AERO_AIR_JIT"] = old
    yield c
    c.close()


@pytest.mark.parametrize("width,log_n,aux,opt", [
    (2, 5, (0, 0, 2), [6, 4, 0, 4, 1, 2, 3]),
    (2, 10, (0, 0, 2), OPT),
    (2, 10, (0, 0, 2), [27, 8, 16, 4, 2, 8, 8]),
    (8, 12, (3, 4, 2), [27, 8, 8, 4, 1, 4, 7]),
    (4, 11, (2, 3, 3), [20, 8, 8, 4, 2, 8, 6]),
    (6, 10, (9, 16, 8), [27, 8, 8, 4, 1, 4, 7]),
    (72, 10, (9, 16, 8), [27, 8, 16, 4, 1, 4, 8]),
    (4, 13, (2, 2, 5), [27, 16, 8, 4, 2, 2, 6]),
    (2, 16, (0, 0, 2), OPT),
    (2, 15, (1, 1, 2), [27, 8, 16, 4, 2, 8, 8]),
])
def test_fibair_as_program_gives_the_hard_wired_bytes(ctx, oracle, width, log_n, aux, opt):
    trace = aero_amd.fib_trace(width, log_n)
    dev = ctx.trace_upload(trace)
    want, pub = ctx.prove_fib_aux(dev, aux[0], aux[1], options(opt), aux_degree=aux[2])
    ref, ref_pub, _ = oracle.prove_fib_aux(width, log_n, aux[0], aux[1], opt, D=aux[2]) if aux[0] else oracle.prove_fib(width, log_n, opt)
    assert pub == ref_pub and want == ref
    for program in (aero_amd.fib_program(width, aux), A.fib_air(width, aux).to_bytes()):      # the library's emitter and the Python builder
        air = aero_amd.Air(program)
        got = ctx.prove_air(air, dev, pub, options(opt))
        assert got == want
        assert ctx.prove_air(air, trace, pub, options(opt)) == want                           # trace handed over in host memory
        aero_amd.verify_air(got, pub, air, min_query_security_bits=0)


@pytest.mark.parametrize("ext", [1, 2])
def test_fibair_program_at_full_size(ctx, ext):
    # BASELINE config 2 / 3 shape: 2^20 x 2; the program path against the hard-wired kernels (which are pinned to the oracle
    # at this size in test_gpu_parity.py / test_gpu_full_configs.py)
    opt = [27, 8, 16, 4, ext, 8, 8]
    dev = ctx.trace_upload(aero_amd.fib_trace(2, 20))
    want, pub = ctx.prove_fib(dev, options(opt))
    assert ctx.prove_air(aero_amd.Air(aero_amd.fib_program(2)), dev, pub, options(opt)) == want


@pytest.mark.parametrize("log_n,pairs,aux,opt", [
    (5, 1, 0, [6, 8, 0, 4, 1, 2, 4]),
    (8, 2, 3, [27, 8, 8, 4, 1, 8, 6]),
    (10, 13, 9, [27, 8, 8, 4, 1, 4, 7]),          # 50 main + 9 aux transition constraints
    (9, 4, 4, [20, 8, 8, 4, 2, 8, 6]),            # quadratic extension
    (11, 26, 9, [27, 8, 16, 4, 1, 4, 8]),         # 72 + 9 columns: Miden's shape, fold 4
    (12, 2, 2, [27, 16, 8, 4, 2, 4, 6]),          # blowup 16 > constraint blowup 8
    (14, 3, 5, [27, 8, 16, 4, 1, 8, 8]),
])
def test_vm_shaped_program_matches_the_oracle(ctx, oracle, log_n, pairs, aux, opt):
    b, trace, pub = ex.synth_vm(log_n, pairs, aux)
    program = b.to_bytes()
    oracle.air_check_trace(program, trace, pub)
    air = aero_amd.Air(program)
    info = air.info()
    assert info["ce_blowup"] == 8 and info["main_transition"] == 24 + 2 * pairs and info["aux_transition"] == aux
    want, _ = oracle.prove_air(program, trace, pub, opt)
    got = ctx.prove_air(air, ctx.trace_upload(trace), pub, options(opt))
    assert got == want
    assert ctx.prove_air(air, trace, pub, options(opt)) == want
    oracle.verify_air(got, program, pub, log_n)
    aero_amd.verify_air(got, pub, air, min_query_security_bits=0, expected_log_n=log_n)
    with pytest.raises(aero_amd.AeroError):                       # another statement
        aero_amd.verify_air(got, [pub[0] ^ 1] + pub[1:], air, min_query_security_bits=0)


def test_a_trace_that_violates_the_program_does_not_verify(ctx, oracle):
    b, trace, pub = ex.synth_vm(8, 2, 3)
    air = aero_amd.Air(b.to_bytes())
    bad = trace.copy()
    bad[7][100] ^= 1
    proof = ctx.prove_air(air, bad, pub, options([27, 8, 8, 4, 1, 8, 6]))       # the prover does not validate the trace (release-mode winterfell neither)
    with pytest.raises(aero_amd.AeroError) as e:
        aero_amd.verify_air(proof, pub, air, min_query_security_bits=0)
    assert e.value.code == -7
    with pytest.raises(RuntimeError):
        oracle.verify_air(proof, b.to_bytes(), pub, 8)


def test_smallest_program_and_a_degree_nine_constraint(ctx, oracle):
    b, trace, pub = ex.tiny_no_assertion_groups(6)
    opt = [8, 4, 0, 4, 1, 2, 3]
    want, _ = oracle.prove_air(b.to_bytes(), trace, pub, opt)
    assert ctx.prove_air(aero_amd.Air(b.to_bytes()), trace, pub, options(opt)) == want
    # degree 8 gated by a periodic selector: base 8 + 1 cycle = 9 -> constraint-evaluation blowup 16 (needs blowup >= 16)
    n = 1 << 7
    bb = A.AirBuilder(2, num_pub=1)
    sel = bb.periodic([1, 0, 1, 1])
    bb.transition(bb.main_next(0) - (sel * bb.main(0) ** 8 + (1 - sel) * (bb.main(0) + bb.main(1))), 8, cycles=[4])
    bb.transition(bb.main_next(1) - bb.main(1) - 3, 1)
    bb.assert_single(0, 0, 7)
    bb.assert_single(1, 0, 1)
    bb.assert_single(0, -1, bb.pub(0))
    c0, c1 = [7], [1]
    for i in range(n - 1):
        c0.append(pow(c0[i], 8, A.P) if [1, 0, 1, 1][i % 4] else (c0[i] + c1[i]) % A.P)
        c1.append((c1[i] + 3) % A.P)
    tr = np.array([c0, c1], dtype=np.uint64)
    air = aero_amd.Air(bb.to_bytes())
    assert air.info()["ce_blowup"] == 16
    opt = [27, 16, 8, 4, 1, 4, 6]
    want, _ = oracle.prove_air(bb.to_bytes(), tr, [c0[-1]], opt)
    got = ctx.prove_air(air, tr, [c0[-1]], options(opt))
    assert got == want
    aero_amd.verify_air(got, [c0[-1]], air, min_query_security_bits=0)
    with pytest.raises(aero_amd.AeroError):      # blowup 8 < constraint blowup 16
        ctx.prove_air(air, tr, [c0[-1]], options([27, 8, 8, 4, 1, 4, 6]))


@pytest.mark.parametrize("evaluator", ["compiled", "interpreted"])
@pytest.mark.parametrize("ext,nfrag", [(1, 1), (1, 8), (2, 4)])
def test_stage_entry_points_against_the_oracle_intermediates(ctx, ctx_interp, oracle, ext, nfrag, evaluator):
    ctx = ctx if evaluator == "compiled" else ctx_interp
    log_n, pairs, aux = 9, 3, 4
    deg = 2 if ext == 2 else 1
    opt = [27, 8, 8, 4, ext, 8, 6]
    b, trace, pub = ex.synth_vm(log_n, pairs, aux)
    program = b.to_bytes()
    air = aero_amd.Air(program)
    n, W = 1 << log_n, trace.shape[0]
    ncols, C = air.num_divisors(log_n), air.info()["ce_blowup"]
    oracle.prove_air(program, trace, pub, opt, keep_artifacts=True)
    nt, na = air.info()["main_transition"] + air.info()["aux_transition"], air.info()["main_assertions"] + air.info()["aux_assertions"]
    coeffs = oracle.artifact("cons_coeffs", 2 * deg * (nt + na))
    rands = oracle.artifact("aux_rands", deg * 4)
    dev = ctx.trace_upload(trace)
    # build_aux_segment
    auxm = ctx.aux_columns_program(air, dev, pub, rands, ext)
    assert (auxm.download() == oracle.artifact("aux_cols", aux * deg * n).reshape(aux * deg, n)).all()
    lde = ctx.evaluate_columns_over(ctx.interpolate_columns(dev), 3)
    alde = ctx.evaluate_columns_over(ctx.interpolate_columns(auxm), 3)
    # evaluate_fragment: numerator columns, stitched from `nfrag` fragments
    want = oracle.artifact("ce_cols", ncols * deg * C * n).reshape(ncols * deg, C * n)
    got = np.zeros_like(want)
    for k in range(nfrag):
        fi, cols = ctx.eval_constraints_program(air, lde, alde, 3, pub, rands, coeffs, ext, k, nfrag)
        assert fi == k * (C * n // nfrag)
        got[:, fi:fi + cols.shape[1]] = cols
    assert (got == want).all()
    # ConstraintEvaluationTable::into_poly -> composition columns, then their LDE
    polys = ctx.composition_poly_program(air, want, log_n, ext)
    clde = ctx.evaluate_columns_over(polys, 3).download()
    ref = oracle.artifact("comp_lde", C * deg * 8 * n).reshape(C * deg, 8 * n)            # column c * deg + d
    for d in range(deg):
        for q in range(C):
            c = int(format(q, "03b")[::-1], 2)                                         # chunk q = composition column bitrev(q)
            assert (clde[d * C + q] == ref[c * deg + d]).all()


def test_constraint_worker_message_with_a_program(ctx, oracle):
    from aero_amd import messages
    log_n, pairs, aux = 8, 2, 3
    opt = [27, 8, 8, 4, 1, 8, 6]
    b, trace, pub = ex.synth_vm(log_n, pairs, aux)
    air = aero_amd.Air(b.to_bytes())
    n = 1 << log_n
    ncols, C = air.num_divisors(log_n), 8
    oracle.prove_air(b.to_bytes(), trace, pub, opt, keep_artifacts=True)
    info = air.info()
    nt, na = info["main_transition"] + info["aux_transition"], info["main_assertions"] + info["aux_assertions"]
    coeffs = oracle.artifact("cons_coeffs", 2 * (nt + na)).reshape(-1, 2)
    rands = oracle.artifact("aux_rands", 4)
    want = oracle.artifact("ce_cols", ncols * C * n).reshape(ncols, C * n)
    dev = ctx.trace_upload(trace)
    lde = ctx.evaluate_columns_over(ctx.interpolate_columns(dev), 3).download()
    alde = ctx.evaluate_columns_over(ctx.interpolate_columns(ctx.aux_columns_program(air, dev, pub, rands, 1)), 3).download()
    pub_bytes = messages.miden_public_inputs([1, 2, 3, 4], [0, 1], [5])
    got = np.zeros_like(want)
    for k in range(4):
        item = messages.encode_constraint_work_item((trace.shape[0], aux, 4), n, pub_bytes, opt, [rands], coeffs[:nt], coeffs[nt:], list(lde), [list(alde)], 8, k, 4)
        fi, fn, cols = messages.decode_constraint_result(ctx.worker_eval_constraints(item, air, pub))
        assert fn == 4 and cols.shape == (ncols, C * n // 4)
        got[:, fi:fi + cols.shape[1]] = cols
    assert (got == want).all()


def _kernels_of(c, fn):
    c.set_kernel_timing(True)
    out = fn()
    names = set(c.kernel_timing_report())
    c.set_kernel_timing(False)
    return out, names


@pytest.mark.parametrize("case", ["fib8_aux", "vm_small", "vm_quadratic", "vm_72_9", "no_groups"])
def test_compiled_kernel_and_interpreter_give_the_same_bytes(ctx, ctx_interp, oracle, case):
    if case == "fib8_aux":
        program, trace, opt = aero_amd.fib_program(8, (3, 4, 2)), aero_amd.fib_trace(8, 12), [27, 8, 8, 4, 1, 4, 7]
        _, pub = ctx.prove_fib_aux(ctx.trace_upload(trace), 3, 4, options(opt), aux_degree=2)
    elif case == "no_groups":
        b, trace, pub = ex.tiny_no_assertion_groups(6) if hasattr(ex, "tiny_no_assertion_groups") else ex.synth_vm(6, 1, 0)
        program, opt = b.to_bytes(), [6, 8, 0, 4, 1, 2, 4]
    else:
        log_n, pairs, aux, opt = {"vm_small": (8, 2, 3, [27, 8, 8, 4, 1, 8, 6]), "vm_quadratic": (9, 4, 4, [20, 8, 8, 4, 2, 8, 6]),
                                  "vm_72_9": (12, 26, 9, [27, 8, 16, 4, 1, 4, 8])}[case]
        program = aero_amd.synth_vm_program(log_n, pairs, aux, 16 if pairs == 26 else 4)
        trace, pub = aero_amd.synth_vm_trace(log_n, pairs)
    air = aero_amd.Air(program)
    got, names = _kernels_of(ctx, lambda: ctx.prove_air(air, trace, pub, options(opt)))
    ref, names_i = _kernels_of(ctx_interp, lambda: ctx_interp.prove_air(air, trace, pub, options(opt)))
    assert "air_jit_kernel" in names and "air_constraints_kernel" not in names, names
    assert "air_constraints_kernel" in names_i and "air_jit_kernel" not in names_i, names_i
    assert got == ref
    want, _ = oracle.prove_air(program, trace, pub, opt)
    assert got == want
    aero_amd.verify_air(got, pub, air, min_query_security_bits=0)


def test_pool_proves_program_airs_in_flight(oracle):
    # aero_pool_prove_air / aero_pool_prove_air_host: several proofs of one program in flight on one GPU, resident and host traces
    log_n, pairs, aux = 9, 3, 4
    program = aero_amd.synth_vm_program(log_n, pairs, aux, 4)
    trace, pub = aero_amd.synth_vm_trace(log_n, pairs)
    air = aero_amd.Air(program)
    opt = [27, 8, 8, 4, 1, 4, 6]
    want, _ = oracle.prove_air(program, trace, pub, opt)
    pool = aero_amd.Pool(0, 3)
    try:
        devs = [pool.ctx(i).trace_upload(trace) for i in range(3)]
        assert pool.prove_air(air, devs, pub, options(opt), rounds=2) == [want] * 3
        assert pool.prove_air(air, [trace, trace.copy()], pub, options(opt)) == [want] * 2
        with pytest.raises(aero_amd.AeroError):                      # a statement the trace does not satisfy is not detected by the prover,
            pool.prove_air(air, devs, pub[:-1], options(opt))        # a wrong NUMBER of public inputs is
        assert pool.prove_air(air, devs[:1], pub, options(opt)) == [want]      # the pool is still usable
    finally:
        pool.close()


@pytest.mark.parametrize("ext", [1, 2])
def test_trace_validation_on_the_device(ctx, oracle, ext):
    # aero_air_validate_trace = Trace::validate of a debug-mode prover: silent on a good trace, names the first bad row otherwise
    log_n, pairs, aux = 9, 3, 4
    b, trace, pub = ex.synth_vm(log_n, pairs, aux)
    air = aero_amd.Air(b.to_bytes())
    info = air.info()
    dev = ctx.trace_upload(trace)
    assert ctx.validate_trace(air, dev, pub) is None                                   # main segment only
    rands = np.arange(7, 7 + 4 * ext, dtype=np.uint64)
    auxm = ctx.aux_columns_program(air, dev, pub, rands, ext)
    assert ctx.validate_trace(air, dev, pub, aux=auxm, rands=rands, field_extension=ext) is None
    # a flipped cell in a state column (column 7 = s_1): the transition constraints of row 99 -> 100 and 100 -> 101 fail; the first is reported
    bad = trace.copy()
    bad[7][100] ^= 1
    row, kind, idx = ctx.validate_trace(air, ctx.trace_upload(bad), pub)
    assert (row, kind) == (99, "transition") and idx < info["main_transition"]
    with pytest.raises(RuntimeError):                                                  # the oracle's own check agrees
        oracle.air_check_trace(b.to_bytes(), bad, pub)
    # a wrong public input: an assertion at the last step
    wrong = [pub[0] ^ 1] + pub[1:]
    row, kind, idx = ctx.validate_trace(air, dev, wrong)
    assert kind == "assertion" and row == (1 << log_n) - 1
    # a corrupted auxiliary column is found only when the auxiliary segment is handed over
    a = auxm.download()
    a[0][5] ^= 1
    got = ctx.validate_trace(air, dev, pub, aux=ctx.trace_upload(a), rands=rands, field_extension=ext)
    assert got is not None and got[0] in (4, 5) and got[1] == "transition" and got[2] >= info["main_transition"]


def test_one_program_handle_validates_traces_of_different_lengths(ctx):
    # the validation kernel carries the steps of the assertions as literals; "last step" follows the trace length, so a second
    # length on the same aero_air handle must get its own kernel (round-3 advice: the cache key ignored the length)
    air = aero_amd.Air(aero_amd.fib_program(4))
    for log_n in (8, 10, 8, 9):
        trace = aero_amd.fib_trace(4, log_n)
        pub = [int(trace[2 * k + 1][-1]) for k in range(2)]
        assert ctx.validate_trace(air, ctx.trace_upload(trace), pub) is None, log_n
        row, kind, _ = ctx.validate_trace(air, ctx.trace_upload(trace), [pub[0] ^ 1, pub[1]])
        assert (row, kind) == ((1 << log_n) - 1, "assertion"), log_n


# ---- AEROAIR version 2: Assertion::sequence and affine auxiliary builders ------------------------------------------------------
@pytest.mark.parametrize("log_n,stride,opt", [
    (5, 4, [6, 4, 0, 4, 1, 2, 3]),
    (8, 4, [27, 8, 8, 4, 1, 8, 6]),
    (10, 2, [20, 8, 8, 4, 2, 4, 6]),              # quadratic extension; a sequence of n / 2 values
    (12, 16, [27, 16, 8, 4, 1, 4, 7]),            # blowup 16 > constraint blowup 4
    (14, 8, [27, 8, 16, 4, 1, 8, 8]),
])
def test_version2_program_compiled_interpreted_and_oracle_agree(ctx, ctx_interp, oracle, log_n, stride, opt):
    b, trace, pub = ex.v2_air(log_n, stride)
    program = b.to_bytes()
    assert program[7] == 2
    air = aero_amd.Air(program)
    want, _ = oracle.prove_air(program, trace, pub, opt)
    got, names = _kernels_of(ctx, lambda: ctx.prove_air(air, ctx.trace_upload(trace), pub, options(opt)))
    assert "air_jit_kernel" in names and "air_scatter_kernel" in names, names          # the sequence tables were built on the device
    assert got == want
    ref, names_i = _kernels_of(ctx_interp, lambda: ctx_interp.prove_air(air, trace, pub, options(opt)))      # host hand-over + interpreter
    assert "air_constraints_kernel" in names_i and ref == want
    oracle.verify_air(got, program, pub, log_n)
    aero_amd.verify_air(got, pub, air, min_query_security_bits=0, expected_log_n=log_n)
    # another sequence: the same proof must not verify
    b2, _, _ = ex.v2_air(log_n, stride)
    b2.sequences[0][1] ^= 1
    with pytest.raises(aero_amd.AeroError):
        aero_amd.verify_air(got, pub, aero_amd.Air(b2.to_bytes()), min_query_security_bits=0)


@pytest.mark.parametrize("ext,nfrag", [(1, 1), (2, 4)])
def test_version2_stage_entry_points(ctx, oracle, ext, nfrag):
    # the affine builders against the oracle's auxiliary columns, the numerator fragments (with the sequence tables) against its table
    log_n, opt = 9, [27, 8, 8, 4, ext, 8, 6]
    deg = 2 if ext == 2 else 1
    b, trace, pub = ex.v2_air(log_n)
    program = b.to_bytes()
    air = aero_amd.Air(program)
    info = air.info()
    n, C, ncols = 1 << log_n, info["ce_blowup"], air.num_divisors(log_n)
    oracle.prove_air(program, trace, pub, opt, keep_artifacts=True)
    nt, na = info["main_transition"] + info["aux_transition"], info["main_assertions"] + info["aux_assertions"]
    coeffs = oracle.artifact("cons_coeffs", 2 * deg * (nt + na))
    rands = oracle.artifact("aux_rands", deg * 4)
    dev = ctx.trace_upload(trace)
    auxm = ctx.aux_columns_program(air, dev, pub, rands, ext)
    A_ = info["aux_width"]                                 # 5: product with denominator, running sum, mixed affine, constant, general (host-built)
    assert (auxm.download() == oracle.artifact("aux_cols", A_ * deg * n).reshape(A_ * deg, n)).all()
    lde = ctx.evaluate_columns_over(ctx.interpolate_columns(dev), 3)
    alde = ctx.evaluate_columns_over(ctx.interpolate_columns(auxm), 3)
    want = oracle.artifact("ce_cols", ncols * deg * C * n).reshape(ncols * deg, C * n)
    got = np.zeros_like(want)
    for k in range(nfrag):
        fi, cols = ctx.eval_constraints_program(air, lde, alde, 3, pub, rands, coeffs, ext, k, nfrag)
        got[:, fi:fi + cols.shape[1]] = cols
    assert (got == want).all()
    # Trace::validate on the device knows the sequences too
    assert ctx.validate_trace(air, dev, pub, aux=auxm, rands=rands, field_extension=ext) is None
    bad = trace.copy()
    bad[2][1 + 4 * 5] ^= 1                                 # the 6th asserted step of the counter's sequence (and two transitions around it)
    got_bad = ctx.validate_trace(air, ctx.trace_upload(bad), pub)
    assert got_bad is not None and got_bad[0] in (20, 21)


@pytest.mark.parametrize("world", [2, 4])
def test_version2_program_sharded_over_thread_ranks(world, oracle):
    from aero_amd.shard import LocalGroup
    log_n, opt = 10, [27, 8, 8, 4, 1, 4, 6]
    b, trace, pub = ex.v2_air(log_n)
    program = b.to_bytes()
    air = aero_amd.Air(program)
    want, _ = oracle.prove_air(program, trace, pub, opt)
    g = LocalGroup(world, min_peer_digests=64)
    try:
        proofs = g.run(lambda r, c, comm: c.prove_air(air, c.trace_upload(trace), pub, options(opt), comm=comm))
    finally:
        g.close()
    assert all(p == want for p in proofs)


# ---- round-4 review items ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ext", [1, 2])
def test_zero_denominator_in_an_aux_builder_affects_its_own_row_only(ctx, oracle, ext):
    """A builder's denominator that vanishes on one row: 1 / 0 = 0 element by element (winter-math's inversion; the oracle inverts each
    row on its own). The device shares one inversion between the rows of a lane: a zero must not zero the lane's other rows - the rows
    BEFORE it keep their values (running product) / only the offending row's term vanishes (affine forms)."""
    log_n = 10
    n = 1 << log_n
    b = A.AirBuilder(1, 3, 1, num_pub=0)
    m, mn, a, an = b.main, b.main_next, b.aux, b.aux_next
    b.transition(mn(0) - m(0) - 1, 1)
    b.aux_transition(an(0) * (m(0) - 5) - a(0) * (m(0) + 1), 2)               # p' = p (m + 1) / (m - 5): the denominator vanishes on row 5
    b.aux_transition((an(1) - a(1)) * (m(0) - 9) - m(0), 2)                   # s' = s + m / (m - 9): row 9
    b.aux_transition((an(2) - a(2) * (m(0) + 2)) * (m(0) - 700) - 3, 2)       # u' = u (m + 2) + 3 / (m - 700): row 700 (another lane, another block)
    b.assert_single(0, 0, 0)
    b.aux_assert_single(0, 0, 1)
    b.aux_assert_single(1, 0, 0)
    b.aux_assert_single(2, 0, 1)
    b.aux_builder(0, 1, m(0) + 1, m(0) - 5)
    b.aux_builder(1, 0, 1, None, m(0), m(0) - 9)
    b.aux_builder(2, 1, m(0) + 2, None, 3, m(0) - 700)
    trace = np.arange(n, dtype=np.uint64)[None, :]
    program = b.to_bytes()
    air = aero_amd.Air(program)
    opt = [27, 8, 8, 4, ext, 8, 6]
    oracle.prove_air(program, trace, [], opt, keep_artifacts=True)           # the trace violates the program where 1 / 0 = 0: a prover does not care
    rands = oracle.artifact("aux_rands", ext * 1)
    want = oracle.artifact("aux_cols", 3 * ext * n).reshape(3 * ext, n)
    got = ctx.aux_columns_program(air, ctx.trace_upload(trace), [], rands, ext).download()
    assert (want[0][:6] != 0).all() and (want[0][6:] == 0).all()             # what "its own row only" means for a running product
    assert (got == want).all(), [int(np.argmax(g != w)) for g, w in zip(got, want) if (g != w).any()]


def test_sequence_assertions_of_a_million_values_are_validated(ctx):
    """Trace::validate with the raw values of long sequence assertions (2^21 rows, a sequence on every second step = 2^20 values + two more
    sequences): the parameter pack outgrows the context's 8 MiB pinned staging block and travels in pieces."""
    log_n = 21
    b, trace, pub = ex.v2_air(log_n, 2)
    air = aero_amd.Air(b.to_bytes())
    dev = ctx.trace_upload(trace)
    assert ctx.validate_trace(air, dev, pub) is None
    bad = trace.copy()
    bad[2][1 + 2 * 777777] ^= 1                                               # the 777 778th value of the counter's sequence
    got = ctx.validate_trace(air, ctx.trace_upload(bad), pub)
    assert got is not None and got[0] in (1 + 2 * 777777 - 1, 1 + 2 * 777777)
