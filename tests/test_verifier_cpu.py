"""aero_verify_fib — the library's own host-side verifier (aero_amd/csrc/verify.hip; no GPU needed).
Pinned on the reference's golden proof proofs/fib.bin (tests/golden/fib.bin), cross-checked against the oracle's verifier
(an independent implementation, oracle/stark.hpp) on accepted proofs and on a sweep of single-byte corruptions."""
import os
import random

import pytest

import aero_amd

DEFAULT = [27, 8, 16, 4, 1, 8, 8]
QUAD = [27, 8, 16, 4, 2, 8, 8]


def test_golden_miden_proof_accepted(oracle, golden_dir):
    """G1: the proof the reference's own prover produced (unknown AIR: no OOD constraint check, like the Cairo verifier)."""
    blob = open(os.path.join(golden_dir, "fib.bin"), "rb").read()
    inputs, proof = oracle.container_split(blob)
    pub = oracle.miden_pub_elements(inputs)
    aero_amd.verify_fib(proof, pub)                         # accepted
    with pytest.raises(aero_amd.AeroError) as e:
        aero_amd.verify_fib(proof, [int(pub[0]) ^ 1] + [int(v) for v in pub[1:]])
    assert e.value.code == -7
    rng = random.Random(1)
    for _ in range(40):                                      # any flipped byte must be caught somewhere
        i = rng.randrange(len(proof))
        bad = bytearray(proof)
        bad[i] ^= 1 << rng.randrange(8)
        with pytest.raises(aero_amd.AeroError):
            aero_amd.verify_fib(bytes(bad), pub)


SHAPES = [
    (2, 8, 0, 0, 2, DEFAULT), (4, 9, 0, 0, 2, QUAD), (6, 8, 0, 0, 2, [20, 8, 8, 4, 1, 4, 6]), (2, 6, 0, 0, 2, [8, 16, 4, 4, 2, 2, 4]),
    (2, 8, 1, 1, 2, DEFAULT), (4, 9, 9, 16, 2, QUAD), (2, 9, 2, 3, 5, DEFAULT), (4, 8, 3, 4, 8, QUAD), (72, 7, 9, 16, 8, [27, 8, 8, 4, 1, 4, 5]),
    (2, 3, 0, 0, 2, [4, 8, 0, 4, 1, 2, 3]), (2, 8, 9, 5, 2, [64, 128, 16, 4, 1, 2, 5]),
]


@pytest.mark.parametrize("W,log_n,A,R,D,opt", SHAPES)
def test_accepts_what_the_oracle_proves_and_agrees_on_corruptions(oracle, W, log_n, A, R, D, opt):
    proof, pub, _ = oracle.prove_fib_aux(W, log_n, A, R, opt, D=D)
    air = (A, R, D)
    aero_amd.verify_fib(proof, pub, air)
    aero_amd.verify_fib(proof, pub)                          # also as an unknown AIR
    # wrong statement / wrong AIR parameters
    with pytest.raises(aero_amd.AeroError):
        aero_amd.verify_fib(proof, [int(pub[0]) ^ 1] + [int(v) for v in pub[1:]], air)
    if A:
        with pytest.raises(aero_amd.AeroError):
            aero_amd.verify_fib(proof, pub, (A, R, 3 if D != 3 else 4))
    # single-bit corruptions: both independent verifiers must reject
    rng = random.Random(W * 1000 + log_n)
    for _ in range(30):
        i = rng.randrange(len(proof))
        bad = bytearray(proof)
        bad[i] ^= 1 << rng.randrange(8)
        bad = bytes(bad)
        ours = True
        try:
            aero_amd.verify_fib(bad, pub, air)
        except aero_amd.AeroError:
            ours = False
        theirs = True
        try:
            oracle.verify_fib_aux(bad, pub, W, log_n, A, R, D=D)
        except RuntimeError:
            theirs = False
        # the oracle follows the Cairo verifier, which does not bind every header byte (e.g. aux_rands without an aux
        # segment); the library verifier also checks the header against the AIR, so it may only be the stricter one
        assert not (ours and not theirs), f"library verifier accepts a corruption at byte {i} that the oracle rejects"
        assert not ours, f"corruption at byte {i} accepted"


def test_truncated_and_extended_proofs_are_rejected(oracle):
    proof, pub, _ = oracle.prove_fib(2, 8, DEFAULT)
    for cut in (0, 10, 21, 100, len(proof) - 1):
        with pytest.raises(aero_amd.AeroError):
            aero_amd.verify_fib(proof[:cut] if cut else b"\x00", pub, (0, 0, 2))
    with pytest.raises(aero_amd.AeroError):
        aero_amd.verify_fib(proof + b"\x00", pub, (0, 0, 2))


def test_command_line_verifies_the_golden_container(golden_dir):
    import subprocess
    import sys
    root = os.path.dirname(golden_dir.rstrip("/").rsplit("/tests", 1)[0] + "/x")
    r = subprocess.run([sys.executable, "-m", "aero_amd", "verify", os.path.join(golden_dir, "fib.bin"), "--miden"], capture_output=True,
                       text=True, cwd=root)
    assert r.returncode == 0 and "accepted" in r.stdout, r.stderr
    r = subprocess.run([sys.executable, "-m", "aero_amd", "verify", os.path.join(golden_dir, "fib.bin")], capture_output=True, text=True, cwd=root)
    assert r.returncode != 0                     # as a FibAir proof it must be rejected
