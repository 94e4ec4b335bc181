"""The two re-encoders of a finished proof (aero_amd/csrc/export.hip; host only):

  * Cairo-memory JSON (`stark_parser <file> <command>`, miden-to-cairo-parser/src/{lib,memory,main}.rs) — read back here the way
    the Cairo side reads it: `write_into_memory` (src/stark_verifier/utils.py:10-23) turns the JSON into a flat memory with
    pointers, and the structs of src/stark_verifier/air/stark_proof.cairo / air/pub_inputs.cairo / channel.cairo:206-244 are
    overlaid on it. Every field must equal what a byte-level parse of the proof gives, and every authentication path must
    verify against its commitment exactly like `verify_merkle_proof` does (channel.cairo:206-244).
  * protobuf sdk.StarkProof / sdk.MidenPublicInputs (aero-sdk/proto/*.proto, contents after convert_proof.rs:13-307) — parsed by
    the official protobuf runtime against the same schema (rebuilt here as descriptors: the schema is data) and re-serialised
    to the identical bytes, which pins the wire format prost (absent third-party crate) would emit.

Inputs: the reference's golden proof tests/golden/fib.bin (a real Miden proof: 72 + 9 columns) and oracle proofs of the
stand-in AIR in the same shape."""
import hashlib
import json
import os
import struct

import pytest

import aero_amd

P = aero_amd.P


# ---- byte-level parse of StarkProof::to_bytes, independent of the library's parser ------------------------------
def parse_proof(b):
    o = 0

    def take(n):
        nonlocal o
        v = b[o:o + n]
        assert len(v) == n
        o += n
        return v

    def u(n):
        return int.from_bytes(take(n), "little")

    pr = {"W": u(1), "A": u(1), "R": u(1), "log_n": u(1)}
    pr["meta"] = take(u(2))
    assert u(1) == 8
    pr["modulus"] = take(8)
    pr["opt"] = list(take(7))
    pr["roots"] = [take(32) for _ in range(u(2) // 32)]
    pr["trace_q"] = []
    for _ in range(2 if pr["A"] else 1):
        pr["trace_q"].append((take(u(4)), take(u(4))))
    pr["cons_q"] = (take(u(4)), take(u(4)))
    pr["ood_states"] = take(u(2))
    pr["ood_evals"] = take(u(2))
    pr["fri"] = [(take(u(4)), take(u(4))) for _ in range(u(1))]
    pr["remainder"] = take(u(2))
    assert u(1) == 0
    pr["nonce"] = u(8)
    assert o == len(b)
    return pr


def felts(b):
    return list(struct.unpack(f"<{len(b) // 8}Q", b))


def golden(golden_dir):
    blob = open(os.path.join(golden_dir, "fib.bin"), "rb").read()
    (n,) = struct.unpack_from("<Q", blob, 0)
    inputs = blob[8:8 + n]
    (m,) = struct.unpack_from("<Q", blob, 8 + n)
    return inputs, blob[16 + n:16 + n + m]


# ---- the Cairo side: memory + struct overlay -----------------------------------------------------------------------------------
def load_memory(js):
    """write_into_memory with ptr = 0: hex strings are values, decimal strings are addresses (utils.py:16-22)."""
    arr = json.loads(js)
    assert all(isinstance(x, str) for x in arr)
    return [(int(x, 16), False) if x.startswith("0x") else (int(x), True) for x in arr], arr


class Mem:
    def __init__(self, js):
        self.cells, self.raw = load_memory(js)

    def val(self, a):
        v, is_ptr = self.cells[a]
        assert not is_ptr, f"cell {a} holds a pointer"
        return v

    def ptr(self, a):
        v, is_ptr = self.cells[a]
        assert is_ptr, f"cell {a} holds a value"
        assert 0 <= v <= len(self.cells)
        return v

    def vals(self, a, n):
        return [self.val(a + i) for i in range(n)]


def digest_words(d):
    return list(struct.unpack("<8I", d))


def hash_elements(elems):
    return hashlib.blake2s(b"".join(int(e).to_bytes(8, "little") + bytes(24) for e in elems)).digest()


def merge(a, b):
    return hashlib.blake2s(a + b).digest()


def verify_path(words, position, root):
    """channel.cairo:206-244: path[0] = leaf, then one sibling per level; the low bit of the position picks the side."""
    digs = [struct.pack("<8I", *words[8 * i:8 * i + 8]) for i in range(len(words) // 8)]
    acc = digs[0]
    for sib in digs[1:]:
        acc = merge(sib, acc) if position & 1 else merge(acc, sib)
        position >>= 1
    assert acc == root
    return digs[0]


def check_cairo_proof_image(proof, js):
    pr = parse_proof(proof)
    W, A, R, log_n, opt = pr["W"], pr["A"], pr["R"], pr["log_n"], pr["opt"]
    Q, N = opt[0], (1 << log_n) * opt[1]
    m = Mem(js)
    a = 0
    # ProofContext (stark_proof.cairo:28-38) starting with TraceLayout (:9-14)
    assert m.vals(a, 2) == [W, 1]
    assert m.vals(m.ptr(a + 2), 1) == [A] and m.vals(m.ptr(a + 3), 1) == [R]
    assert m.vals(a + 4, 3) == [1 << log_n, log_n, len(pr["meta"])]
    assert m.vals(m.ptr(a + 7), len(pr["meta"])) == list(pr["meta"])
    assert m.val(a + 8) == 8 and m.vals(m.ptr(a + 9), 8) == list(pr["modulus"])
    # ProofOptions (:16-26): num_queries, blowup, log_blowup, grinding, hash_fn, field_extension, folding factor, max remainder size
    assert m.vals(a + 10, 8) == [opt[0], opt[1], opt[1].bit_length() - 1, opt[2], opt[3], opt[4], opt[5], 1 << opt[6]]
    assert m.val(a + 18) == N
    a += 19
    # ParsedCommitments (:40-45)
    nseg = 2
    layers = len(pr["fri"])
    assert len(pr["roots"]) == nseg + 1 + layers + 1
    tr = m.ptr(a)
    for i in range(nseg):
        assert m.vals(tr + 8 * i, 8) == digest_words(pr["roots"][i])
    assert m.vals(m.ptr(a + 1), 8) == digest_words(pr["roots"][nseg])
    assert m.val(a + 2) == layers + 1
    fr = m.ptr(a + 3)
    for i in range(layers + 1):
        assert m.vals(fr + 8 * i, 8) == digest_words(pr["roots"][nseg + 1 + i])
    a += 4
    # ParsedOodFrame (:47-51): two EvaluationFrames (frame.cairo:70-75) and a Vec
    st = felts(pr["ood_states"])
    TW = W + A
    exp = [st[:W], st[TW:TW + W], st[W:TW], st[TW + W:]]
    for k in range(4):
        assert m.val(a + 2 * k) == len(exp[k]) and m.vals(m.ptr(a + 2 * k + 1), len(exp[k])) == exp[k]
    ev = felts(pr["ood_evals"])
    assert m.val(a + 8) == len(ev) and m.vals(m.ptr(a + 9), len(ev)) == ev
    a += 10
    assert m.val(a) == pr["nonce"]
    a += 1
    # TraceQueries (:53-56), ConstraintQueries (:58-60): Table = n_rows, n_cols, elements (table.cairo)
    for data, cols in ((pr["trace_q"][0][0], W), (pr["trace_q"][1][0], A), (pr["cons_q"][0], len(ev))):
        assert m.vals(a, 2) == [Q, cols]
        assert m.vals(m.ptr(a + 2), Q * cols) == felts(data)
        a += 3
    rem = felts(pr["remainder"])
    assert m.val(a) == len(rem) and m.vals(m.ptr(a + 1), len(rem)) == rem
    # the root segment is exactly the StarkProof struct: the first pointer target is where it ends
    assert m.ptr(2) == a + 2
    # formats (memory.rs:14-18, lib.rs:222-231): plain values upper-case without padding, field elements 16 lower-case digits
    assert m.raw[0] == f"0x{W:X}" and m.raw[m.ptr(a + 1)] == f"0x{rem[0]:016x}"
    return pr


def test_cairo_proof_image_of_the_golden_proof(golden_dir):
    inputs, proof = golden(golden_dir)
    js = aero_amd.cairo_memory("proof", proof)
    pr = check_cairo_proof_image(proof, js)
    assert (pr["W"], pr["A"], pr["R"], pr["log_n"]) == (72, 9, 16, 10)
    assert " " not in js and js.startswith('["0x48","0x1",')           # serde_json::to_string of a Vec<String>
    kat = json.load(open(os.path.join(golden_dir, "fib_kat.json")))["G1"]
    m = Mem(js)
    assert m.val(19 + 4 + 10) == kat["pow_nonce"]


def test_cairo_public_inputs_of_the_golden_proof(golden_dir):
    inputs, _ = golden(golden_dir)
    m = Mem(aero_amd.cairo_memory("public-inputs", input_bytes=inputs))
    kat = json.load(open(os.path.join(golden_dir, "fib_kat.json")))
    # PublicInputs (pub_inputs.cairo:17-23): program_hash_len, program_hash*, stack_inputs_len, stack_inputs*, outputs{stack_len, stack*, overflow_len, overflow*}
    assert m.val(0) == 4 and m.vals(m.ptr(1), 4) == kat["G3"]["program_hash_elements"]       # tests/integration/test_verifier.cairo:41-47
    n_in = m.val(2)
    stack_in = m.vals(m.ptr(3), n_in)
    n_out = m.val(4)
    stack_out = m.vals(m.ptr(5), n_out)
    n_ov = m.val(6)
    m.ptr(7)
    assert stack_in == [1, 0] and stack_out[:2] == [55, 34] and n_out == 16 and n_ov == 0        # fib(10) = 55
    # same elements, same order as the coin seed (crypto/random.cairo:254-280)
    off, parts = 32, []
    for _ in range(3):
        (c,) = struct.unpack_from("<Q", inputs, off)
        parts.append(list(struct.unpack_from(f"<{c}Q", inputs, off + 8)))
        off += 8 + 8 * c
    assert [stack_in, stack_out, []] == parts
    # Felt vs u64 formatting: hash / stack inputs are 16-digit lower-case, outputs are `{:#X}`
    assert m.raw[m.ptr(3)] == "0x0000000000000001" and m.raw[m.ptr(5)] == "0x37"


def test_cairo_query_paths_of_the_golden_proof_authenticate(golden_dir):
    inputs, proof = golden(golden_dir)
    pr = parse_proof(proof)
    pos = json.load(open(os.path.join(golden_dir, "fib_kat.json")))["G1"]["positions"]
    Q, N, W, A = 27, 8192, 72, 9
    depth = 13
    # trace-queries: QueriesProofs = one pointer per segment; each target holds Q x (length, digests*) (channel.cairo:196-204)
    m = Mem(aero_amd.cairo_memory("trace-queries", proof, indexes=pos))
    for s, width in enumerate((W, A)):
        base = m.ptr(s)
        rows = felts(pr["trace_q"][s][0])
        for i in range(Q):
            assert m.val(base + 2 * i) == depth + 1
            leaf = verify_path(m.vals(m.ptr(base + 2 * i + 1), 8 * (depth + 1)), pos[i], pr["roots"][s])
            assert leaf == hash_elements(rows[i * width:(i + 1) * width])
    m = Mem(aero_amd.cairo_memory("constraint-queries", proof, indexes=pos))
    base = m.ptr(0)
    rows = felts(pr["cons_q"][0])
    for i in range(Q):
        assert m.val(base + 2 * i) == depth + 1
        leaf = verify_path(m.vals(m.ptr(base + 2 * i + 1), 8 * (depth + 1)), pos[i], pr["roots"][2])
        assert leaf == hash_elements(rows[i * 8:(i + 1) * 8])
    # fri-queries: per layer a pointer; target holds per folded position (length, digests*, values*) (lib.rs:452-467)
    m = Mem(aero_amd.cairo_memory("fri-queries", proof, indexes=pos))
    dom, cur = N, pos
    for l in range(2):
        tgt = dom // 8
        folded = []
        for p_ in cur:
            if p_ % tgt not in folded:
                folded.append(p_ % tgt)
        base = m.ptr(l)
        vals = felts(pr["fri"][l][0])
        assert len(vals) == 8 * len(folded)
        d = tgt.bit_length() - 1
        for i, fp in enumerate(folded):
            assert m.val(base + 3 * i) == d + 1
            leaf = verify_path(m.vals(m.ptr(base + 3 * i + 1), 8 * (d + 1)), fp, pr["roots"][3 + l])
            row = m.vals(m.ptr(base + 3 * i + 2), 8)
            assert row == vals[8 * i:8 * i + 8] and leaf == hash_elements(row)
        dom, cur = tgt, folded
    # wrong positions cannot be encoded: the paths would not reach the commitments
    with pytest.raises(aero_amd.AeroError):
        aero_amd.cairo_memory("trace-queries", proof, indexes=[p_ ^ 1 for p_ in pos])
    with pytest.raises(aero_amd.AeroError):
        aero_amd.cairo_memory("trace-queries", proof, indexes=pos[:-1])


def test_cairo_image_of_a_standin_proof_in_the_cairo_shape(oracle):
    """A proof of the stand-in AIR in the shape the Cairo verifier hard-codes (72 + 9 columns, degree-8 constraints => 8
    composition columns, 27 queries, blowup 8, fold 8): the image has the same layout as the golden proof's, and the library's
    verifier accepts the proof in cairo-compat mode. A proof in another shape is refused by that mode; one without an auxiliary
    segment has no image at all (the reference's encoder unwraps the auxiliary frame)."""
    opt = [27, 8, 16, 4, 1, 8, 8]
    proof, pub, _ = oracle.prove_fib_aux(72, 7, 9, 16, opt, D=8)
    check_cairo_proof_image(proof, aero_amd.cairo_memory("proof", proof))
    aero_amd.verify_fib(proof, pub, (9, 16, 8), cairo_compat=True, expected_log_n=7)
    other, pub2, _ = oracle.prove_fib_aux(4, 7, 9, 16, opt, D=8)
    aero_amd.verify_fib(other, pub2, (9, 16, 8))
    with pytest.raises(aero_amd.AeroError) as e:
        aero_amd.verify_fib(other, pub2, (9, 16, 8), cairo_compat=True)
    assert "cairo-compat" in str(e.value)
    fold4, pub3, _ = oracle.prove_fib_aux(72, 7, 9, 16, [27, 8, 16, 4, 1, 4, 8], D=8)
    with pytest.raises(aero_amd.AeroError):
        aero_amd.verify_fib(fold4, pub3, (9, 16, 8), cairo_compat=True)
    plain, _, _ = oracle.prove_fib(2, 7, opt)
    with pytest.raises(aero_amd.AeroError) as e:
        aero_amd.cairo_memory("proof", plain)
    assert e.value.code == -5
    quad, _, _ = oracle.prove_fib_aux(4, 7, 2, 3, [27, 8, 16, 4, 2, 8, 8])
    with pytest.raises(aero_amd.AeroError) as e:
        aero_amd.cairo_memory("proof", quad)
    assert e.value.code == -5
    with pytest.raises(aero_amd.AeroError):
        aero_amd.cairo_memory("proof", proof[:-3])


# ---- protobuf ---------------------------------------------------------------------------------------------------------------
def sdk_messages():
    """The schema of aero-sdk/proto/{common,context,commitments,queries,ood_frame,fri_proof,stark_proof,miden_vm}.proto as
    descriptors for the official runtime."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="sdk_all.proto", package="sdk", syntax="proto3")

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, num, typ, rep, tname in fields:
            f = m.field.add(name=fname, number=num, type=typ, label=F.LABEL_REPEATED if rep else F.LABEL_OPTIONAL)
            if tname:
                f.type_name = ".sdk." + tname
        return m

    for ename, values in (("HashFunction", ["BLAKE2S"]), ("FieldExtension", ["NONE"]), ("PrimeField", ["GOLDILOCKS"])):
        e = fd.enum_type.add(name=ename)
        for i, v in enumerate(values):
            e.value.add(name=v, number=i)
    U32, U64, BYT, MSG, ENU = F.TYPE_UINT32, F.TYPE_UINT64, F.TYPE_BYTES, F.TYPE_MESSAGE, F.TYPE_ENUM
    msg("FieldElement", [("element", 2, BYT, 0, None)])
    msg("Table", [("n_rows", 1, U32, 0, None), ("n_cols", 2, U32, 0, None), ("elements", 3, MSG, 1, "FieldElement")])
    msg("Digest", [("data", 2, BYT, 0, None)])
    msg("ProofOptions", [("num_queries", 1, U32, 0, None), ("blowup_factor", 2, U32, 0, None), ("grinding_factor", 3, U32, 0, None),
                         ("hash_fn", 4, ENU, 0, "HashFunction"), ("field_extension", 5, ENU, 0, "FieldExtension"),
                         ("fri_folding_factor", 6, U32, 0, None), ("fri_max_remainder_size", 7, U32, 0, None), ("prime_field", 8, ENU, 0, "PrimeField")])
    msg("TraceLayout", [("main_segment_width", 1, U64, 0, None), ("aux_segment_widths", 2, U64, 1, None), ("aux_segment_rands", 3, U64, 1, None),
                        ("num_aux_segments", 4, U64, 0, None)])
    msg("Context", [("trace_layout", 1, MSG, 0, "TraceLayout"), ("trace_length", 2, U64, 0, None), ("trace_meta", 3, BYT, 0, None),
                    ("field_modulus", 4, MSG, 0, "FieldElement"), ("options", 5, MSG, 0, "ProofOptions")])
    msg("Commitments", [("trace_roots", 1, MSG, 1, "Digest"), ("constraint_root", 2, MSG, 0, "Digest"), ("fri_roots", 3, MSG, 1, "Digest")])
    msg("BatchMerkleProofLayer", [("nodes", 1, MSG, 1, "Digest")])
    msg("BatchMerkleProof", [("leaves", 1, MSG, 1, "Digest"), ("nodes", 2, MSG, 1, "BatchMerkleProofLayer"), ("depth", 3, U32, 0, None)])
    msg("TraceQueries", [("main_states", 1, MSG, 0, "Table"), ("aux_states", 2, MSG, 0, "Table"), ("query_proofs", 3, MSG, 1, "BatchMerkleProof")])
    msg("ConstraintQueries", [("evaluations", 1, MSG, 0, "Table"), ("query_proof", 2, MSG, 0, "BatchMerkleProof")])
    msg("EvaluationFrame", [("current", 1, MSG, 1, "FieldElement"), ("next", 2, MSG, 1, "FieldElement")])
    msg("OodFrame", [("main_frame", 1, MSG, 0, "EvaluationFrame"), ("aux_frame", 2, MSG, 0, "EvaluationFrame"), ("evaluations", 3, MSG, 1, "FieldElement")])
    msg("FriProofLayer", [("values", 1, MSG, 1, "FieldElement"), ("proofs", 2, MSG, 0, "BatchMerkleProof")])
    msg("FriProof", [("layers", 1, MSG, 1, "FriProofLayer"), ("remainder", 2, MSG, 1, "FieldElement"), ("num_partitions", 3, U32, 0, None)])
    msg("StarkProof", [("context", 1, MSG, 0, "Context"), ("commitments", 2, MSG, 0, "Commitments"), ("trace_queries", 3, MSG, 0, "TraceQueries"),
                       ("constraint_queries", 4, MSG, 0, "ConstraintQueries"), ("ood_frame", 5, MSG, 0, "OodFrame"), ("fri_proof", 6, MSG, 0, "FriProof"),
                       ("pow_nonce", 7, U64, 0, None)])
    msg("MidenProgramOutputs", [("stack", 1, MSG, 1, "FieldElement"), ("overflow_addrs", 2, MSG, 1, "FieldElement")])
    msg("MidenPublicInputs", [("program_hash", 1, MSG, 0, "Digest"), ("stack_inputs", 2, MSG, 1, "FieldElement"), ("outputs", 3, MSG, 0, "MidenProgramOutputs")])
    msg("ProofSubmissionRequest", [("proof", 1, MSG, 0, "StarkProof"), ("public_inputs", 2, MSG, 0, "MidenPublicInputs"),
                                   ("source_proof_system", 3, U32, 0, None), ("target_chain", 4, U32, 0, None)])   # enums: varints
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = message_factory.GetMessageClass
    return {n: get(pool.FindMessageTypeByName("sdk." + n)) for n in ("StarkProof", "MidenPublicInputs", "MidenProgramOutputs", "ProofSubmissionRequest")}


def fe(vals):
    return [int.from_bytes(v.element, "little") for v in vals]


def batch_vectors(paths):
    o, out = 1, []
    for _ in range(paths[0]):
        k = paths[o]
        o += 1
        out.append([paths[o + 32 * i:o + 32 * i + 32] for i in range(k)])
        o += 32 * k
    assert o == len(paths)
    return out


def check_protobuf(proof, msgs):
    pr = parse_proof(proof)
    W, A, R, log_n, opt = pr["W"], pr["A"], pr["R"], pr["log_n"], pr["opt"]
    Q, N = opt[0], (1 << log_n) * opt[1]
    data = aero_amd.proof_to_protobuf(proof)
    sp = msgs["StarkProof"]()
    sp.ParseFromString(data)
    assert sp.SerializeToString(deterministic=True) == data, "not the canonical proto3 encoding of this message"
    from google.protobuf import unknown_fields
    assert len(unknown_fields.UnknownFieldSet(sp)) == 0 and len(unknown_fields.UnknownFieldSet(sp.trace_queries)) == 0
    c = sp.context
    assert c.trace_layout.main_segment_width == W and list(c.trace_layout.aux_segment_widths) == ([A] if A else [])
    assert list(c.trace_layout.aux_segment_rands) == ([R] if A else []) and c.trace_layout.num_aux_segments == (1 if A else 0)
    assert c.trace_length == 1 << log_n and c.trace_meta == pr["meta"] and c.field_modulus.element == pr["modulus"]
    o = c.options
    assert [o.num_queries, o.blowup_factor, o.grinding_factor, o.hash_fn, o.field_extension, o.fri_folding_factor, o.fri_max_remainder_size,
            o.prime_field] == [opt[0], opt[1], opt[2], 0, 0, opt[5], 1 << opt[6], 0]
    nseg, layers = (2 if A else 1), len(pr["fri"])
    assert [d.data for d in sp.commitments.trace_roots] == pr["roots"][:nseg]
    assert sp.commitments.constraint_root.data == pr["roots"][nseg]
    assert [d.data for d in sp.commitments.fri_roots] == pr["roots"][nseg + 1:] and len(sp.commitments.fri_roots) == layers + 1
    depth = N.bit_length() - 1
    tq = sp.trace_queries
    assert (tq.main_states.n_rows, tq.main_states.n_cols) == (Q, W) and fe(tq.main_states.elements) == felts(pr["trace_q"][0][0])
    assert tq.HasField("aux_states") == bool(A) and len(tq.query_proofs) == nseg
    if A:
        assert (tq.aux_states.n_rows, tq.aux_states.n_cols) == (Q, A) and fe(tq.aux_states.elements) == felts(pr["trace_q"][1][0])
    for s, width in enumerate((W, A)[:nseg]):
        bp = tq.query_proofs[s]
        rows = felts(pr["trace_q"][s][0])
        assert [d.data for d in bp.leaves] == [hash_elements(rows[i * width:(i + 1) * width]) for i in range(Q)]
        assert [[d.data for d in layer.nodes] for layer in bp.nodes] == batch_vectors(pr["trace_q"][s][1]) and bp.depth == depth
    C = len(pr["ood_evals"]) // 8
    cq = sp.constraint_queries
    assert (cq.evaluations.n_rows, cq.evaluations.n_cols) == (Q, C) and fe(cq.evaluations.elements) == felts(pr["cons_q"][0])
    rows = felts(pr["cons_q"][0])
    assert [d.data for d in cq.query_proof.leaves] == [hash_elements(rows[i * C:(i + 1) * C]) for i in range(Q)]
    assert [[d.data for d in layer.nodes] for layer in cq.query_proof.nodes] == batch_vectors(pr["cons_q"][1]) and cq.query_proof.depth == depth
    st, TW = felts(pr["ood_states"]), W + A
    assert fe(sp.ood_frame.main_frame.current) == st[:W] and fe(sp.ood_frame.main_frame.next) == st[TW:TW + W]
    assert sp.ood_frame.HasField("aux_frame") == bool(A)
    if A:
        assert fe(sp.ood_frame.aux_frame.current) == st[W:TW] and fe(sp.ood_frame.aux_frame.next) == st[TW + W:]
    assert fe(sp.ood_frame.evaluations) == felts(pr["ood_evals"])
    Fd, dom = opt[5], N
    assert len(sp.fri_proof.layers) == layers
    for l, layer in enumerate(sp.fri_proof.layers):
        vals = felts(pr["fri"][l][0])
        assert fe(layer.values) == vals
        assert [d.data for d in layer.proofs.leaves] == [hash_elements(vals[i:i + Fd]) for i in range(0, len(vals), Fd)]
        assert [[d.data for d in x.nodes] for x in layer.proofs.nodes] == batch_vectors(pr["fri"][l][1])
        dom //= Fd
        assert layer.proofs.depth == dom.bit_length() - 1
    assert fe(sp.fri_proof.remainder) == felts(pr["remainder"]) and sp.fri_proof.num_partitions == 0
    assert sp.pow_nonce == pr["nonce"]
    return len(data)


def test_protobuf_of_the_golden_proof(golden_dir):
    msgs = sdk_messages()
    inputs, proof = golden(golden_dir)
    size = check_protobuf(proof, msgs)
    assert size > len(proof)                                           # explicit leaves + per-element framing
    data = aero_amd.miden_public_inputs_to_protobuf(inputs)
    pi = msgs["MidenPublicInputs"]()
    pi.ParseFromString(data)
    assert pi.SerializeToString(deterministic=True) == data
    assert pi.program_hash.data == inputs[:32]
    assert fe(pi.stack_inputs) == [1, 0] and fe(pi.outputs.stack)[:2] == [55, 34] and len(pi.outputs.stack) == 16 and len(pi.outputs.overflow_addrs) == 0


@pytest.mark.parametrize("W,log_n,A,R,D,opt", [
    (2, 8, 0, 0, 2, [27, 8, 16, 4, 1, 8, 8]),            # no auxiliary segment: aux_states / aux_frame absent, one query proof
    (4, 7, 3, 2, 2, [12, 4, 0, 4, 1, 2, 4]),             # grinding 0 (a zero scalar is not written), fold 2, several layers
    (72, 7, 9, 16, 8, [27, 8, 16, 4, 1, 4, 8]),          # config-5 shape
])
def test_protobuf_of_oracle_proofs(oracle, W, log_n, A, R, D, opt):
    proof, _, _ = oracle.prove_fib_aux(W, log_n, A, R, opt, D=D)
    check_protobuf(proof, sdk_messages())


def test_protobuf_refuses_what_the_schema_cannot_say(oracle):
    quad, _, _ = oracle.prove_fib(2, 6, [27, 8, 16, 4, 2, 8, 5])
    with pytest.raises(aero_amd.AeroError) as e:
        aero_amd.proof_to_protobuf(quad)
    assert e.value.code == -5                      # FieldExtension::Quadratic => todo!() in convert_proof.rs:140-147
    with pytest.raises(aero_amd.AeroError):
        aero_amd.proof_to_protobuf(b"\x01\x02\x03")


def test_prover_output_message_of_the_golden_proof(golden_dir):
    """What the reference's proving worker posts back to the SDK (proving_worker.rs:205-222): bincode ProverOutput of the three
    protobuf payloads (utils.rs:424-430: three Vec<u8> = u64 length + bytes each)."""
    from aero_amd import messages
    msgs = sdk_messages()
    inputs, proof = golden(golden_dir)
    blob = aero_amd.prover_output(proof, inputs)
    pb_proof, pb_outputs, pb_inputs = messages.decode_prover_output(blob)
    assert blob == b"".join(struct.pack("<Q", len(x)) + x for x in (pb_proof, pb_outputs, pb_inputs))
    assert pb_proof == aero_amd.proof_to_protobuf(proof) and pb_inputs == aero_amd.miden_public_inputs_to_protobuf(inputs)
    po = msgs["MidenProgramOutputs"]()
    po.ParseFromString(pb_outputs)
    assert po.SerializeToString(deterministic=True) == pb_outputs
    pi = msgs["MidenPublicInputs"]()
    pi.ParseFromString(pb_inputs)
    assert fe(po.stack) == fe(pi.outputs.stack) and fe(po.stack)[:2] == [55, 34] and len(po.overflow_addrs) == 0
    with pytest.raises(aero_amd.AeroError):
        aero_amd.prover_output(proof[:-1], inputs)
    with pytest.raises(aero_amd.AeroError):
        aero_amd.prover_output(proof, inputs[:-3])


def test_worker_message_layouts():
    """The bincode layouts aero_amd.messages writes, byte by byte, on the data of the reference's own unit test
    (utils.rs:460-478 `test_work_item_serialization`: rows [[1, 2], [3, 4]], batch 0)."""
    from aero_amd import messages
    item = messages.encode_hashing_work_item([[1, 2], [3, 4]], 0)
    q = lambda *v: b"".join(struct.pack("<Q", x) for x in v)
    assert item == q(2, 2, 1, 2, 2, 3, 4, 0)
    digests = [hash_elements([1, 2]), hash_elements([3, 4])]
    assert digests[0].hex() == "1466784a2149964c3bb5af60fb274365a73ced9e96459ea486fe330a3afa4177"       # SURVEY a5 known answer
    assert messages.decode_hashing_result(q(7, 2) + digests[0] + digests[1]) == (7, digests)
    # ConstraintComputeWorkItem of utils.rs:481-545 (layout (2, [1], [1]), 8 rows, coefficient pairs as in the test; a 2 x 16 LDE here)
    pub = messages.miden_public_inputs([9, 8, 7, 6], [0, 1], [2, 3])
    assert pub == q(9, 8, 7, 6, 2, 0, 1, 2, 2, 3, 0)
    main = [list(range(16)), list(range(100, 116))]
    aux = [[list(range(200, 216))]]
    w = messages.encode_constraint_work_item((2, 1, 1), 8, pub, [27, 8, 17, 4, 1, 16, 7], [[5]], [(1, 2), (3, 4)], [(5, 6), (7, 8), (7, 8)],
                                             main, aux, 2, 0, 8)
    lde = q(3, 2, 16, *main[0], 16, *main[1], 1, 1, 16, *aux[0][0], 2)
    want = (q(3, 3) + bytes([2, 1, 1]) + q(8, 0) + q(1, len(pub)) + pub + q(1, 7) + bytes([27, 8, 17, 4, 1, 16, 7]) + q(1, 1, 5) +
            q(2, 2, 1, 2, 3, 4, 3, 5, 6, 7, 8, 7, 8) + q(len(lde)) + lde + q(0, 8))
    assert w == want
    fi, fn, cols = messages.decode_constraint_result(q(4, 8, 3, 2, 10, 11, 2, 20, 21, 2, 30, 31))
    assert (fi, fn) == (4, 8) and cols.tolist() == [[10, 11], [20, 21], [30, 31]]


def test_stark_parser_command_line_is_the_reference_cli(golden_dir):
    """bin/stark_parser (aero_amd/csrc/stark_parser.cpp): the command line the Cairo side's hints call (src/stark_verifier/utils.py:33-41,
    tests/integration/utils.py:5-24: [parser, path, command, indexes]) - same sub-commands, the JSON array + the newline println! adds."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    aero_amd.lib()                                   # builds the library and the parser when they are missing
    exe = os.path.join(root, "bin", "stark_parser")
    assert os.path.exists(exe)
    path = os.path.join(golden_dir, "fib.bin")
    inputs, proof = golden(golden_dir)
    pos = json.load(open(os.path.join(golden_dir, "fib_kat.json")))["G1"]["positions"]

    def run(*args):
        r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=120)
        return r.returncode, r.stdout, r.stderr

    assert run(path, "proof") == (0, aero_amd.cairo_memory("proof", proof, inputs) + "\n", "")
    assert run(path, "public-inputs") == (0, aero_amd.cairo_memory("public-inputs", b"", inputs) + "\n", "")
    for cmd in ("trace-queries", "constraint-queries", "fri-queries"):
        rc, out, err = run(path, cmd, json.dumps(pos))
        assert (rc, err) == (0, "") and out == aero_amd.cairo_memory(cmd, proof, inputs, pos) + "\n"
        json.loads(out)
    assert run(path, "trace-queries", json.dumps([p + 1 for p in pos]))[0] == 1          # paths that do not reach the commitment
    assert run(path, "nonsense")[0] == 1 and run(os.path.join(golden_dir, "missing.bin"), "proof")[0] == 1
    # interpolate-poly (main.rs:103-109): values as 8-byte little-endian hex, coefficients low to high, folded with ", "
    hx = lambda vals: json.dumps([int(v).to_bytes(8, "little").hex() for v in vals])
    xs = [3, 5, 7, 11, P - 2]
    coeffs = [9, 0, P - 1, 12345678901234567, 4]
    ys = [sum(c * pow(x, i, P) for i, c in enumerate(coeffs)) % P for x in xs]
    rc, out, _ = run(path, "interpolate-poly", hx(xs), hx(ys))
    assert rc == 0 and out == "".join(f", {c}" for c in coeffs) + "\n"


def test_worker_messages_are_validated_on_the_host():
    """aero_worker_message_info: the layouts of the SDK's worker messages checked without a GPU (what aero_worker_* will accept)."""
    from aero_amd import messages
    item = messages.encode_hashing_work_item([[1, 2], [3, 4, 5], []], 7)
    assert aero_amd.worker_message_info("hashing", item) == dict(rows=3, batch_idx=7, min_width=0, max_width=3, elements=5)
    for bad in (item[:-1], item + b"\0" * 8, struct.pack("<Q", 2 ** 61) + item[8:], item[:8] + struct.pack("<Q", 2 ** 30) + item[16:]):
        with pytest.raises(aero_amd.AeroError):
            aero_amd.worker_message_info("hashing", bad)
    pub = messages.miden_public_inputs([9, 8, 7, 6], [0, 1], [2, 3])
    main = [list(range(16)), list(range(100, 116))]
    w = messages.encode_constraint_work_item((2, 1, 1), 8, pub, [27, 2, 17, 4, 1, 16, 7], [[5]], [(1, 2), (3, 4), (5, 6)], [(5, 6), (7, 8), (7, 8), (1, 1)],
                                             main, [[list(range(200, 216))]], 2, 3, 8)
    assert aero_amd.worker_message_info("constraints", w) == dict(main_width=2, aux_width=1, aux_rands=1, trace_len=8, blowup=2, fragment_offset=3,
                                                                   num_fragments=8, coefficient_pairs=7)
    for bad in (w[:-1], w + b"\1", w[:8] + struct.pack("<Q", 4) + w[16:], w.replace(bytes([27, 2, 17, 4, 1, 16, 7]), bytes([27, 2, 17, 4, 1, 16]), 1)):
        with pytest.raises(aero_amd.AeroError):
            aero_amd.worker_message_info("constraints", bad)
