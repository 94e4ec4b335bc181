"""Oracle: FibAir with an auxiliary trace segment. What fib.bin pins for this path (it has
one aux segment): transcript order (main root, aux_rands draws, aux root), the context bytes, the second trace-query
block, the OOD frame layout main || aux, 3 DEEP coefficients per column over main then aux — all exercised by
test_oracle_golden.py through the same parser / verifier code. What it cannot pin (the Miden AIR is absent): the aux
constraint set itself, which is restatement-defined and checked here by prove -> verify WITH the OOD constraint check."""
import pytest

DEFAULT = [27, 8, 16, 4, 1, 8, 8]
QUAD = [27, 8, 16, 4, 2, 8, 8]

CASES = [
    (2, 8, 1, 1, DEFAULT),
    (2, 10, 3, 2, DEFAULT),
    (4, 9, 9, 16, QUAD),                 # Miden's aux shape: 9 columns from 16 random elements, over F_p^2
    (6, 8, 2, 5, [20, 8, 8, 4, 1, 4, 6]),
    (2, 6, 1, 1, [8, 16, 4, 4, 2, 2, 4]),
    (72, 7, 9, 16, [27, 8, 8, 4, 1, 4, 5]),   # config-5 shape (72 main + 9 aux columns, fold 4) at a toy length
]


@pytest.mark.parametrize("W,log_n,A,R,opt", CASES)
def test_aux_prove_verify(oracle, W, log_n, A, R, opt):
    proof, pub, _ = oracle.prove_fib_aux(W, log_n, A, R, opt)
    oracle.verify_fib_aux(proof, pub, W, log_n, A, R)
    # context bytes: main width, aux width, aux rands, log2(trace length)
    assert list(proof[:4]) == [W, A, R, log_n]
    # same transcript plumbing as the golden proof: the generic verifier (no AIR knowledge, like the Cairo one) accepts it
    oracle.verify(proof, pub, air_kind=0)


@pytest.mark.parametrize("W,log_n,A,R,D,opt", [(2, 8, 1, 1, 3, DEFAULT), (2, 9, 2, 3, 5, DEFAULT), (4, 8, 3, 4, 8, QUAD),
                                               (72, 7, 9, 16, 8, [27, 8, 8, 4, 1, 4, 5])])
def test_aux_constraint_degree(oracle, W, log_n, A, R, D, opt):
    """Aux constraints of degree D: 4 composition columns for D in 3..4, 8 for 5..8 (the golden proof's shape)."""
    import struct
    proof, pub, _ = oracle.prove_fib_aux(W, log_n, A, R, opt, D=D)
    oracle.verify_fib_aux(proof, pub, W, log_n, A, R, D=D)
    with pytest.raises(RuntimeError):
        oracle.verify_fib_aux(proof, pub, W, log_n, A, R, D=2 if D != 2 else 3)
    # OOD evaluations section length = composition columns * element bytes
    off = 22
    (clen,) = struct.unpack_from("<H", proof, off)
    off += 2 + clen
    for _ in range(6):
        (l,) = struct.unpack_from("<I", proof, off)
        off += 4 + l
    (l,) = struct.unpack_from("<H", proof, off)
    off += 2 + l
    (l,) = struct.unpack_from("<H", proof, off)
    deg = 2 if opt[4] == 2 else 1
    assert l == (4 if D <= 4 else 8) * 8 * deg


def test_aux_rejections(oracle):
    W, log_n, A, R = 2, 8, 3, 2
    proof, pub, _ = oracle.prove_fib_aux(W, log_n, A, R, DEFAULT)
    for i in (40, len(proof) // 3, len(proof) // 2, len(proof) - 20):
        bad = bytearray(proof)
        bad[i] ^= 1
        with pytest.raises(RuntimeError):
            oracle.verify_fib_aux(bytes(bad), pub, W, log_n, A, R)
    with pytest.raises(RuntimeError):
        oracle.verify_fib_aux(proof, pub, W, log_n, A, R + 1)
    with pytest.raises(RuntimeError):
        oracle.verify_fib_aux(proof, [pub[0] ^ 1], W, log_n, A, R)
    with pytest.raises(RuntimeError):
        oracle.prove_fib_aux(2, 8, 3, 0, DEFAULT)          # aux columns need random elements


def test_no_aux_is_plain_fib(oracle):
    a, _, _ = oracle.prove_fib(4, 9, DEFAULT)
    b, _, _ = oracle.prove_fib_aux(4, 9, 0, 0, DEFAULT)
    assert a == b


def test_aux_deterministic_across_threads(oracle):
    oracle.set_threads(1)
    a, _, _ = oracle.prove_fib_aux(4, 9, 3, 4, QUAD)
    oracle.set_threads(4)
    b, _, _ = oracle.prove_fib_aux(4, 9, 3, 4, QUAD)
    assert a == b
