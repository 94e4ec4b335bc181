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
    unknown = dict(allow_unknown_air=True)
    aero_amd.verify_fib(proof, pub, None, **unknown)       # accepted (27 queries x 3 bits + 16 grinding = 97 >= the default floor of 96)
    aero_amd.verify_fib(proof, pub, None, cairo_compat=True, expected_log_n=10, **unknown)   # and it has the shape the Cairo verifier hard-codes
    assert aero_amd.proof_security_bits(proof) == (97, 64 - 13)
    with pytest.raises(aero_amd.AeroError) as e:
        aero_amd.verify_fib(proof, [int(pub[0]) ^ 1] + [int(v) for v in pub[1:]], None, **unknown)
    assert e.value.code == -7
    # the default call refuses to run without an AIR: skipping the OOD constraint check must be asked for explicitly
    with pytest.raises(aero_amd.AeroError) as e:
        aero_amd.verify_fib(proof, pub, None)
    assert e.value.code == -1 and "mandatory" in str(e.value)
    with pytest.raises(aero_amd.AeroError):
        aero_amd.verify_fib(proof, pub, None, expected_log_n=11, **unknown)
    rng = random.Random(1)
    for _ in range(40):                                      # any flipped byte must be caught somewhere
        i = rng.randrange(len(proof))
        bad = bytearray(proof)
        bad[i] ^= 1 << rng.randrange(8)
        with pytest.raises(aero_amd.AeroError):
            aero_amd.verify_fib(bytes(bad), pub, None, **unknown)


SHAPES = [
    (2, 8, 0, 0, 2, DEFAULT), (4, 9, 0, 0, 2, QUAD), (6, 8, 0, 0, 2, [20, 8, 8, 4, 1, 4, 6]), (2, 6, 0, 0, 2, [8, 16, 4, 4, 2, 2, 4]),
    (2, 8, 1, 1, 2, DEFAULT), (4, 9, 9, 16, 2, QUAD), (2, 9, 2, 3, 5, DEFAULT), (4, 8, 3, 4, 8, QUAD), (72, 7, 9, 16, 8, [27, 8, 8, 4, 1, 4, 5]),
    (2, 3, 0, 0, 2, [4, 8, 0, 4, 1, 2, 3]), (2, 8, 9, 5, 2, [64, 128, 16, 4, 1, 2, 5]),
]


@pytest.mark.parametrize("W,log_n,A,R,D,opt", SHAPES)
def test_accepts_what_the_oracle_proves_and_agrees_on_corruptions(oracle, W, log_n, A, R, D, opt):
    proof, pub, _ = oracle.prove_fib_aux(W, log_n, A, R, opt, D=D)
    air = (A, R, D)
    weak = dict(min_query_security_bits=0)                   # several shapes use few queries / no grinding on purpose
    aero_amd.verify_fib(proof, pub, air, expected_log_n=log_n, require_options=aero_amd.ProofOptions(*opt), **weak)
    aero_amd.verify_fib(proof, pub, None, allow_unknown_air=True, **weak)   # also as an unknown AIR
    # wrong statement / wrong AIR parameters / wrong trace length / other options than the caller requires
    with pytest.raises(aero_amd.AeroError):
        aero_amd.verify_fib(proof, [int(pub[0]) ^ 1] + [int(v) for v in pub[1:]], air, **weak)
    if A:
        with pytest.raises(aero_amd.AeroError):
            aero_amd.verify_fib(proof, pub, (A, R, 3 if D != 3 else 4), **weak)
    with pytest.raises(aero_amd.AeroError):
        aero_amd.verify_fib(proof, pub, air, expected_log_n=log_n + 1, **weak)
    other = list(opt)
    other[2] ^= 1
    with pytest.raises(aero_amd.AeroError):
        aero_amd.verify_fib(proof, pub, air, require_options=aero_amd.ProofOptions(*other), **weak)
    q_bits = opt[0] * (opt[1].bit_length() - 1) + opt[2]
    assert aero_amd.proof_security_bits(proof)[0] == q_bits
    if q_bits < 96:
        with pytest.raises(aero_amd.AeroError) as e:
            aero_amd.verify_fib(proof, pub, air)              # the default policy wants 96 query-security bits
        assert "security" in str(e.value)
    else:
        aero_amd.verify_fib(proof, pub, air)
    # single-bit corruptions: both independent verifiers must reject
    rng = random.Random(W * 1000 + log_n)
    for _ in range(30):
        i = rng.randrange(len(proof))
        bad = bytearray(proof)
        bad[i] ^= 1 << rng.randrange(8)
        bad = bytes(bad)
        ours = True
        try:
            aero_amd.verify_fib(bad, pub, air, **weak)
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


def test_forged_low_security_proof_is_rejected_by_default(oracle):
    """The attack of ADVICE r1: a proof that declares num_queries = 1, grinding = 0, blowup = 2 is internally consistent - the
    prover may pick any options - so a verifier that takes the options from the proof accepts ~1 bit of soundness. The default
    policy (96 query-security bits) rejects it; so does pinning the options; a policy that allows it accepts it."""
    weak_opt = [1, 2, 0, 4, 1, 2, 3]
    proof, pub, _ = oracle.prove_fib(2, 6, weak_opt)
    assert aero_amd.proof_security_bits(proof)[0] == 1
    with pytest.raises(aero_amd.AeroError) as e:
        aero_amd.verify_fib(proof, pub, (0, 0, 2))
    assert e.value.code == -7 and "security" in str(e.value)
    with pytest.raises(aero_amd.AeroError):
        aero_amd.verify_fib(proof, pub, (0, 0, 2), min_query_security_bits=0, require_options=aero_amd.ProofOptions(*DEFAULT))
    aero_amd.verify_fib(proof, pub, (0, 0, 2), min_query_security_bits=1)


def test_tampered_ood_frame_is_rejected_through_the_default_call(oracle):
    """A false statement with a valid low-degree commitment: change one OOD trace value. Every hash in the transcript follows
    from the proof bytes, so only the OOD constraint consistency check can notice in general - it runs on the default path."""
    proof, pub, _ = oracle.prove_fib(2, 8, DEFAULT)
    info = oracle.verify(proof, pub, air_kind=1, W=2, log_n=8, want_info=True)
    assert info
    # locate the OOD trace states: context (22) + commitments (2 + 32 * roots) + trace queries + constraint queries
    import struct
    off = 22
    (clen,) = struct.unpack_from("<H", proof, off)
    off += 2 + clen
    for _ in range(2 * 2):                                    # (values, paths) of the trace segment and of the constraint queries
        (ln,) = struct.unpack_from("<I", proof, off)
        off += 4 + ln
    (olen,) = struct.unpack_from("<H", proof, off)
    assert olen == 2 * 2 * 8
    bad = bytearray(proof)
    v = int.from_bytes(bad[off + 2:off + 10], "little")
    bad[off + 2:off + 10] = ((v + 1) % aero_amd.P).to_bytes(8, "little")
    with pytest.raises(aero_amd.AeroError):
        aero_amd.verify_fib(bytes(bad), pub, (0, 0, 2))


def test_command_line_verifies_the_golden_container(golden_dir):
    import subprocess
    import sys
    root = os.path.dirname(golden_dir.rstrip("/").rsplit("/tests", 1)[0] + "/x")
    r = subprocess.run([sys.executable, "-m", "aero_amd", "verify", os.path.join(golden_dir, "fib.bin"), "--miden"], capture_output=True,
                       text=True, cwd=root)
    assert r.returncode == 0 and "accepted" in r.stdout, r.stderr
    r = subprocess.run([sys.executable, "-m", "aero_amd", "verify", os.path.join(golden_dir, "fib.bin")], capture_output=True, text=True, cwd=root)
    assert r.returncode != 0                     # as a FibAir proof it must be rejected
