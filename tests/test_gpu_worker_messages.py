"""The reference's worker seam at the message level (include/aero_stark.h: aero_worker_hash_rows / aero_worker_eval_constraints):
the library is handed the bincode bytes the SDK's pool posts to its web workers (aero-sdk/miden-wasm/src/pool.rs:84-125,
utils.rs:302-450) and must answer with the bytes `blake2_hash_elements` (hashing_worker.rs:12-26) and `constraint_compute`
(constraints_worker.rs:14-79) would post back. Checked against hashlib's BLAKE2s over the 32-byte-padded elements, the known
answer of the reference's conventions, and the intermediates of the oracle's own prover run."""
import hashlib
import struct

import numpy as np
import pytest

import aero_amd
from aero_amd import messages

pytestmark = pytest.mark.gpu
P = aero_amd.P


@pytest.fixture(scope="module")
def ctx():
    c = aero_amd.Context(0)
    yield c
    c.close()


def hash_elements(row):
    return hashlib.blake2s(b"".join(int(e).to_bytes(8, "little") + bytes(24) for e in row)).digest()


def test_hashing_worker_on_the_reference_unit_test_item(ctx):
    """utils.rs:460-478: rows [[1, 2], [3, 4]], batch 0."""
    res = ctx.worker_hash_rows(messages.encode_hashing_work_item([[1, 2], [3, 4]], 0))
    batch, digests = messages.decode_hashing_result(res)
    assert batch == 0 and digests == [hash_elements([1, 2]), hash_elements([3, 4])]
    assert digests[0].hex() == "1466784a2149964c3bb5af60fb274365a73ced9e96459ea486fe330a3afa4177"
    assert res == struct.pack("<QQ", 0, 2) + digests[0] + digests[1]


def test_hashing_worker_uniform_and_ragged_batches(ctx, oracle):
    rng = np.random.default_rng(11)
    # the shape the pool posts: a chunk of LDE rows of one width (pool.rs:84-99)
    rows = rng.integers(0, P, size=(4096, 72), dtype=np.uint64)
    batch, digests = messages.decode_hashing_result(ctx.worker_hash_rows(messages.encode_hashing_work_item(rows, 5)))
    assert batch == 5
    want = oracle.hash_rows(np.ascontiguousarray(rows.T))                 # oracle: column-major matrix -> one digest per row
    assert b"".join(digests) == want.tobytes()
    for i in (0, 1, 4095):
        assert digests[i] == hash_elements(rows[i])
    # rows of different lengths are answered row by row, in order
    ragged = [rng.integers(0, P, size=int(k), dtype=np.uint64) for k in rng.integers(1, 82, size=300)]
    batch, digests = messages.decode_hashing_result(ctx.worker_hash_rows(messages.encode_hashing_work_item(ragged, 2 ** 40 + 3)))
    assert batch == 2 ** 40 + 3 and digests == [hash_elements(r) for r in ragged]
    # an empty work item is a valid message
    assert messages.decode_hashing_result(ctx.worker_hash_rows(messages.encode_hashing_work_item([], 9))) == (9, [])
    # a row without elements: hash_elements(&[]) = BLAKE2s of the empty string
    _, d = messages.decode_hashing_result(ctx.worker_hash_rows(messages.encode_hashing_work_item([[7], [], [1, 2, 3]], 0)))
    assert d == [hash_elements([7]), hashlib.blake2s(b"").digest(), hash_elements([1, 2, 3])]
    # Felt::new reduces: p + 5 is the element 5
    _, d = messages.decode_hashing_result(ctx.worker_hash_rows(messages.encode_hashing_work_item([[P + 5, 1]], 0)))
    assert d == [hash_elements([5, 1])]


def test_hashing_worker_rejects_malformed_items(ctx):
    good = messages.encode_hashing_work_item([[1, 2], [3, 4]], 0)
    for bad in (good[:-1], good + b"\0", good[:20], struct.pack("<Q", 2 ** 60) + good[8:], good[:8] + struct.pack("<Q", 2 ** 40) + good[16:], good[:-3]):
        with pytest.raises(aero_amd.AeroError):
            ctx.worker_hash_rows(bad)
    assert messages.decode_hashing_result(ctx.worker_hash_rows(good))[0] == 0          # the context is still usable


@pytest.mark.parametrize("log_n,width,A,R,D,nfrag", [(8, 2, 0, 0, 2, 8), (8, 4, 3, 2, 3, 4), (7, 72, 9, 16, 8, 2)])
def test_constraint_worker_matches_the_oracle_fragments(ctx, oracle, log_n, width, A, R, D, nfrag):
    """ConstraintComputeWorkItem built the way the proving worker builds it (proving_worker.rs:396-437: the whole trace LDE, the
    coefficients drawn from the channel, one fragment per item) -> the fragment of the oracle's merged numerator columns."""
    opt = [27, 8, 16, 4, 1, 8, 5]
    n, N = 1 << log_n, 8 << log_n
    Cc = 2 if (not A or D <= 2) else (4 if D <= 4 else 8)
    proof, pub, _ = oracle.prove_fib_aux(width, log_n, A, R, opt, D=D, keep_artifacts=True)
    want = oracle.artifact("ce_cols", 3 * Cc * n).reshape(3, Cc * n)
    ncoef = 2 * ((width + A) + (width + width // 2 + A))
    coeffs = oracle.artifact("cons_coeffs", ncoef).reshape(-1, 2)
    rands = oracle.artifact("aux_rands", R) if A else np.zeros(0, np.uint64)
    dev = ctx.trace_upload(aero_amd.fib_trace(width, log_n))
    lde = ctx.evaluate_columns_over(ctx.interpolate_columns(dev), 3).download()
    aux_segments = []
    if A:
        aux = ctx.aux_columns_fib(dev, (A, R, D), rands, 1)
        aux_segments = [list(ctx.evaluate_columns_over(ctx.interpolate_columns(aux), 3).download())]
    pub_bytes = messages.miden_public_inputs([1, 2, 3, 4], [0, 1], pub)
    nt = width + A
    program = aero_amd.Air(aero_amd.fib_program(width, (A, R, D)))     # the AIR travels as a program: the message names none
    got = np.zeros_like(want)
    for k in range(nfrag):
        item = messages.encode_constraint_work_item((width, A, R), n, pub_bytes, opt, [rands] if A else [], coeffs[:nt], coeffs[nt:], list(lde),
                                                    aux_segments, 8, k, nfrag)
        fi, fn, cols = messages.decode_constraint_result(ctx.worker_eval_constraints(item, program, pub))
        assert fn == nfrag and fi == k * (Cc * n // nfrag) and cols.shape == (3, Cc * n // nfrag)
        got[:, fi:fi + cols.shape[1]] = cols
    assert (got == want).all()


def test_constraint_worker_rejects_inconsistent_items(ctx, oracle):
    log_n, width = 6, 2
    opt = [27, 8, 16, 4, 1, 8, 5]
    n = 1 << log_n
    proof, pub, _ = oracle.prove_fib(width, log_n, opt, keep_artifacts=True)
    coeffs = oracle.artifact("cons_coeffs", 2 * (width + width + width // 2)).reshape(-1, 2)
    lde = list(ctx.evaluate_columns_over(ctx.interpolate_columns(ctx.trace_upload(aero_amd.fib_trace(width, log_n))), 3).download())
    pub_bytes = messages.miden_public_inputs([1, 2, 3, 4], [0, 1], pub)

    def item(**kw):
        a = dict(layout=(width, 0, 0), trace_len=n, public_inputs=pub_bytes, options=opt, aux_rand_elements=[], transition=coeffs[:width],
                 boundary=coeffs[width:], main_cols=lde, aux_segments=[], blowup=8, fragment_offset=0, num_fragments=2)
        a.update(kw)
        return messages.encode_constraint_work_item(**a)

    program = aero_amd.Air(aero_amd.fib_program(width))
    ok = ctx.worker_eval_constraints(item(), program, pub)
    assert messages.decode_constraint_result(ok)[2].shape == (3, n)
    bads = [item(blowup=4), item(layout=(4, 0, 0)), item(trace_len=2 * n), item(transition=coeffs[:1]), item(fragment_offset=2),
            item(num_fragments=3), item(main_cols=lde[:1]),
            item(options=[27, 8, 16, 4, 2, 8, 5]), item(layout=(width, 1, 1)), item()[:-5], item() + b"\1"]
    for bad in bads:
        with pytest.raises(aero_amd.AeroError):
            ctx.worker_eval_constraints(bad, program, pub)
    # public inputs taken from the message: the program reads 1 element, Miden's PublicInputs carry 4 + 2 + len(outputs) of them
    with pytest.raises(aero_amd.AeroError) as e:
        ctx.worker_eval_constraints(item(), program)
    assert e.value.code == -1
    with pytest.raises(aero_amd.AeroError) as e:       # malformed public inputs are a bad argument, not a verification failure
        ctx.worker_eval_constraints(item(public_inputs=b"\1\2\3"), program)
    assert e.value.code == -1
    assert ctx.worker_eval_constraints(item(), program, pub) == ok
