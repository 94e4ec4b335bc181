"""Pins the oracle (oracle/, CPU restatement) against every golden vector the reference's own tests hold for
this path: G1 proofs/fib.bin accepted by tests/integration/test_verifier.cairo:58-74 `test_verify`,
G2 `test_draw` :76-114, G3 `test_read_pub_inputs` :33-56, G4 tests/unit/test_math_g.cairo:5-75.
CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

P = 18446744069414584321


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "fib_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def fib(golden_dir):
    with open(os.path.join(golden_dir, "fib.bin"), "rb") as f:
        return f.read()


def test_container_and_layout_roundtrip(oracle, fib, kat):
    # bincode ProofData framing (miden-proof-generator/src/lib.rs:1-6, main.rs:49-51)
    assert len(fib) == kat["G1"]["file_bytes"]
    inputs, proof = oracle.container_split(fib)
    assert len(inputs) == kat["G1"]["input_bytes"] and len(proof) == kat["G1"]["proof_bytes"]
    assert proof[:22].hex() == kat["G1"]["header_hex"]
    # StarkProof::to_bytes layout: parse + re-serialise must reproduce every byte
    assert oracle.proof_roundtrip(proof) == proof
    assert int.from_bytes(proof[-8:], "little") == kat["G1"]["pow_nonce"]


def test_read_pub_inputs(oracle, fib, kat):
    # test_verifier.cairo:41-47
    inputs, _ = oracle.container_split(fib)
    assert oracle.miden_pub_elements(inputs)[:4] == kat["G3"]["program_hash_elements"]


def test_draw(oracle, fib, kat):
    # test_verifier.cairo:76-114: seed_with_pub_inputs -> random_coin_new -> draw -> draw_integers(20, 64)
    inputs, _ = oracle.container_split(fib)
    seed = oracle.coin_new(oracle.miden_pub_elements(inputs))
    assert seed.hex() == kat["G1"]["coin_seed"]
    v, ctr = oracle.coin_draw(seed, 0)
    assert v == kat["G2"]["first_draw"]
    ints, _ = oracle.coin_draw_integers(seed, ctr, 20, 64)
    assert ints == kat["G2"]["draw_integers_20_of_64"]


def test_verify_fib_bin(oracle, fib, kat):
    # test_verifier.cairo:58-74 `test_verify`: the restated verifier accepts the reference prover's golden proof
    # and derives exactly the transcript listed in SURVEY 8c G1 (roots, z, alphas, PoW, positions, DEEP, shapes).
    inputs, proof = oracle.container_split(fib)
    info = oracle.verify(proof, oracle.miden_pub_elements(inputs), air_kind=0, want_info=True)
    g = kat["G1"]
    assert info["roots"].split(",") == g["roots"]
    assert info["coin_seed0"] == g["coin_seed0"] and info["coin_seed"] == g["coin_seed"]
    assert int(info["z"].split(",")[0]) == g["z"]
    assert [int(x) for x in info["fri_alphas"].split(",")] == g["fri_alphas"]
    assert int(info["lambda"]) == g["lambda"] and int(info["mu"]) == g["mu"]
    assert info["post_nonce_seed"] == g["post_nonce_seed"]
    assert [int(x) for x in info["positions"].split(",")] == g["positions"]
    assert [int(x) for x in info["deep"].split(",")][:2] == g["deep_first_two"]
    assert [[int(a) for a in s.split(":")] for s in info["batch_shapes"].split(";")] == g["batch_shapes"]


def test_verify_rejects_tampering(oracle, fib):
    inputs, proof = oracle.container_split(fib)
    pub = oracle.miden_pub_elements(inputs)
    rng = np.random.default_rng(7)
    for off in rng.integers(30, len(proof) - 9, size=12):
        bad = bytearray(proof)
        bad[int(off)] ^= 0x40
        with pytest.raises(RuntimeError):
            oracle.verify(bytes(bad), pub, air_kind=0)
    # the PoW nonce is minimal: 45692 is the first nonce >= 1 with >= 16 leading zero bits
    pub2 = list(pub)
    pub2[0] ^= 1
    with pytest.raises(RuntimeError):
        oracle.verify(proof, pub2, air_kind=0)


def test_pow_nonce_is_first_hit(oracle, fib, kat):
    # Replay the transcript up to the grinding step and scan upward: fib.bin's nonce must be the minimum.
    inputs, proof = oracle.container_split(fib)
    pub = oracle.miden_pub_elements(inputs)
    info = oracle.verify(proof, pub, air_kind=0, want_info=True)
    # seed before the nonce is not exported; instead check minimality through the public API: every smaller
    # nonce must fail verification for PoW reasons once substituted.
    for nonce in (1, 2, 3, 1000, 45691):
        bad = proof[:-8] + nonce.to_bytes(8, "little")
        with pytest.raises(RuntimeError, match="proof of work"):
            oracle.verify(bad, pub, air_kind=0)
    assert info["post_nonce_seed"].startswith("0000")


def test_goldilocks_kats(oracle, kat):
    # tests/unit/test_math_g.cairo:5-75
    assert kat["G4"]["PG"] == P
    assert oracle.sub(2, 1) == 1 and oracle.sub(1, 2) == P - 1
    assert oracle.add(2, 1) == 3 and oracle.add(P - 1, 2) == 1
    assert oracle.mul(2, 5) == 10 and oracle.mul(P - 1, 2) == P - 2 and oracle.mul(P - 1, 4) == P - 4
    assert oracle.mul(25, oracle.inv(25)) == 1 and oracle.mul(55, oracle.inv(55)) == 1
    assert oracle.pow(5, 3) == 125 and oracle.pow(P - 5, 2) == oracle.mul(P - 5, P - 5)
    # roots of unity (fri_verifier.cairo:154-168)
    for k, v in kat["roots_of_unity"].items():
        assert oracle.root_of_unity(int(k)) == v


def test_field_against_python_ints(oracle):
    rng = np.random.default_rng(1)
    edge = [0, 1, 2, P - 1, P - 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, (1 << 63), P - (1 << 32)]
    vals = edge + [int(x) % P for x in rng.integers(0, 1 << 63, size=300, dtype=np.uint64) * 2 + rng.integers(0, 2, size=300, dtype=np.uint64)]
    for a in vals[:60]:
        for b in vals[:60]:
            assert oracle.mul(a, b) == a * b % P == oracle.mul_slow(a, b)
            assert oracle.add(a, b) == (a + b) % P
            assert oracle.sub(a, b) == (a - b) % P
    for a in vals:
        if a:
            assert oracle.mul(a, oracle.inv(a)) == 1
    # quadratic extension x^2 - x + 2: (a0 + a1 phi)(b0 + b1 phi), phi^2 = phi - 2
    for _ in range(200):
        a = [int(x) % P for x in rng.integers(0, 1 << 63, size=2, dtype=np.uint64)]
        b = [int(x) % P for x in rng.integers(0, 1 << 63, size=2, dtype=np.uint64)]
        want = ((a[0] * b[0] - 2 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0] + a[1] * b[1]) % P)
        assert oracle.e2_mul(a, b) == want
        assert oracle.e2_mul(a, oracle.e2_inv(a)) == (1, 0)


def test_blake2s_against_hashlib(oracle, kat):
    rng = np.random.default_rng(2)
    for ln in [0, 1, 31, 32, 33, 40, 63, 64, 65, 127, 128, 129, 256, 1000]:
        data = rng.integers(0, 256, size=ln, dtype=np.uint8).tobytes()
        assert oracle.blake2s(data) == hashlib.blake2s(data).digest()
    # hash_elements pads every element to 32 bytes (random.cairo:93-104)
    assert oracle.hash_elements([1, 2]).hex() == kat["blake2s_kat"]["hash_elements_1_2"]
    for w in [0, 1, 2, 3, 8, 9, 72, 81]:
        e = [int(x) % P for x in rng.integers(0, 1 << 63, size=w, dtype=np.uint64)]
        blob = b"".join(int(v).to_bytes(8, "little") + bytes(24) for v in e)
        assert oracle.hash_elements(e) == hashlib.blake2s(blob).digest()


def test_merkle_and_batch_proof_shapes(oracle, fib, kat):
    rng = np.random.default_rng(3)
    leaves = rng.integers(0, 256, size=(64, 32), dtype=np.uint8)
    nodes = oracle.merkle_nodes(leaves)
    for i in range(1, 64):
        assert nodes[i].tobytes() == hashlib.blake2s(nodes[2 * i].tobytes() + nodes[2 * i + 1].tobytes()).digest()
    assert (nodes[64:] == leaves).all()
    # adjacent pair shares everything: one vector, depth-1 digests
    pr = oracle.batch_proof(leaves, [10, 11])
    assert pr[0] == 1 and pr[1] == 5 and len(pr) == 2 + 5 * 32
    with pytest.raises(RuntimeError):
        oracle.batch_proof(leaves, [3, 3])
