"""One proof sharded over the ranks of ONE process: the library's in-process communicator (stream-ordered device copies between
the ranks' contexts, event hand-shakes, no RCCL / gloo / Python in the exchange) and the host-memory hand-over of the sharded
proof (every rank copies only its width / world columns; coefficients all-gathered; rows exchanged instead of digests where a
row is shorter than its digest). Every rank's bytes must equal the single-GPU proof's (the reference's own gather
steps: proving_worker.rs:302-310,428-437)."""
import json
import os
import subprocess

import numpy as np
import pytest

import aero_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    c = aero_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("width,log_n,aux,opt,worlds", [
    (2, 12, (0, 0, 2), [27, 8, 16, 4, 1, 8, 8], (2, 4, 8)),          # width < world for 4 and 8: whole-trace copy; rows exchanged (16 B < 32 B)
    (8, 13, (0, 0, 2), [27, 8, 8, 4, 1, 8, 6], (2, 4, 8)),           # column-partitioned hand-over
    (16, 12, (4, 3, 5), [27, 8, 8, 4, 1, 4, 7], (2, 4, 8)),          # aux segment of degree 5 (8 composition columns), fold 4
    (8, 12, (0, 0, 2), [20, 8, 8, 4, 2, 8, 6], (2, 8)),              # quadratic extension
    (72, 10, (9, 16, 8), [27, 8, 8, 4, 1, 4, 7], (8,)),              # Miden's shape
    (8, 12, (5, 3, 5), [27, 8, 8, 4, 1, 4, 7], (4, 2)),              # needed main columns straddle W / 2: ranks 0-2 would fetch 3-4 foreign columns, rank 3 five - one decision for all
    (4, 14, (0, 0, 2), [27, 16, 8, 4, 1, 2, 6], (4, 16)),            # blowup 16: world 16
])
def test_local_group_proofs_equal_the_single_gpu_proof(ctx, oracle, width, log_n, aux, opt, worlds):
    trace = aero_amd.fib_trace(width, log_n)
    o = aero_amd.ProofOptions(*opt)
    want, pub = ctx.prove_fib_aux(trace, aux[0], aux[1], o, aux_degree=aux[2])
    ref, _, _ = oracle.prove_fib_aux(width, log_n, aux[0], aux[1], opt, D=aux[2]) if aux[0] else oracle.prove_fib(width, log_n, opt)
    assert want == ref
    for world in worlds:
        proofs, pub2, ms, sent = aero_amd.prove_fib_sharded_local(trace, o, world, aux)
        assert pub2 == pub
        assert all(p == want for p in proofs), f"world {world}"
        assert all(s > 0 for s in sent)


def test_rows_travel_instead_of_digests_when_they_are_shorter(ctx):
    trace = aero_amd.fib_trace(2, 16)
    o = aero_amd.ProofOptions.with_96_bit_security()
    want, _ = ctx.prove_fib(trace, o)
    sent = {}
    for flag in ("1", "0"):
        os.environ["AERO_EXCHANGE_ROWS"] = flag
        try:
            proofs, _, _, s = aero_amd.prove_fib_sharded_local(trace, o, 8)
        finally:
            os.environ.pop("AERO_EXCHANGE_ROWS", None)
        assert all(p == want for p in proofs)
        sent[flag] = s[0]
    # trace and composition commitments: 16-byte rows instead of 32-byte digests
    assert sent["1"] < sent["0"] - 1.5e6, sent           # 2 commitments x 16 B x 2^16 rows x 7/8 less per rank


@pytest.mark.parametrize("chunks", ["4", "16", "3"])
@pytest.mark.parametrize("width,log_n,aux,opt,worlds", [
    (2, 14, (0, 0, 2), [27, 8, 16, 4, 1, 8, 8], (2, 4, 8)),          # rows exchanged, hashed on arrival while the next piece travels
    (8, 13, (0, 0, 2), [27, 8, 8, 4, 1, 8, 6], (2, 4, 8)),           # digests: piece k + 1 hashed while piece k travels
    (16, 12, (4, 3, 5), [27, 8, 8, 4, 2, 4, 7], (8,)),               # auxiliary segment, quadratic extension, fold 4
])
def test_chunked_exchange_gives_the_single_gpu_proof(ctx, oracle, chunks, width, log_n, aux, opt, worlds):
    """AERO_EXCHANGE_CHUNKS: every commitment's all-to-all cut into pieces, each piece's exchange on the proving stream while the row hashing
    of the neighbouring piece runs on the context's second stream. Same leaves in the same slots: same bytes. (3 is not a
    power of two: the library rounds down to 2.) Ranks sharing one GPU can show parity only; whether it pays needs a node with links."""
    trace = aero_amd.fib_trace(width, log_n)
    o = aero_amd.ProofOptions(*opt)
    want, pub = ctx.prove_fib_aux(trace, aux[0], aux[1], o, aux_degree=aux[2])
    ref, _, _ = oracle.prove_fib_aux(width, log_n, aux[0], aux[1], opt, D=aux[2]) if aux[0] else oracle.prove_fib(width, log_n, opt)
    assert want == ref
    os.environ["AERO_EXCHANGE_CHUNKS"] = chunks
    try:
        for world in worlds:
            proofs, pub2, ms, sent = aero_amd.prove_fib_sharded_local(trace, o, world, aux)
            assert pub2 == pub and all(p == want for p in proofs), f"world {world}, chunks {chunks}"
    finally:
        os.environ.pop("AERO_EXCHANGE_CHUNKS", None)


def test_chunked_exchange_of_a_program_air(oracle):
    from aero_amd.shard import LocalGroup
    log_n, pairs, aux, opt = 12, 26, 9, [27, 8, 16, 4, 1, 4, 8]
    program = aero_amd.synth_vm_program(log_n, pairs, aux, 16)
    trace, pub = aero_amd.synth_vm_trace(log_n, pairs)
    air = aero_amd.Air(program)
    want, _ = oracle.prove_air(program, trace, pub, opt)
    os.environ["AERO_EXCHANGE_CHUNKS"] = "8"
    try:
        g = LocalGroup(4, min_peer_digests=64)
        try:
            proofs = g.run(lambda r, c, comm: c.prove_air(air, c.trace_upload(trace), pub, aero_amd.ProofOptions(*opt), comm=comm))
        finally:
            g.close()
    finally:
        os.environ.pop("AERO_EXCHANGE_CHUNKS", None)
    assert all(p == want for p in proofs)


def test_non_canonical_trace_is_refused_by_every_rank():
    trace = aero_amd.fib_trace(8, 12)
    trace[5][100] = 0xFFFFFFFF00000001 + 1           # rank 2 of 4 copies this column: the verdict is all-reduced
    with pytest.raises(aero_amd.AeroError) as e:
        aero_amd.prove_fib_sharded_local(trace, aero_amd.ProofOptions.with_96_bit_security(), 4)
    assert e.value.code == -1 and "non-canonical" in str(e.value)


def test_plain_c_host_shards_a_proof_without_python(tmp_path):
    exe = str(tmp_path / "sharded_local")
    lib_dir = os.path.join(ROOT, "aero_amd")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "sharded_local.c"),
           "-L", lib_dir, "-laero_stark", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for args in (["14", "8", "4"], ["12", "16", "8", "4", "3", "5"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout, r.stderr)
        res = json.loads(r.stdout.strip().splitlines()[-1])
        assert res["identical"] and res["world"] == int(args[2]) and res["bytes_sent_rank0"] > 0


# ---- AIR programs sharded over thread ranks (aero_prove_air with a communicator) -----------------------------------------------
@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("case", ["vm_small", "vm_quadratic", "vm_72_9", "fib_program"])
def test_program_air_sharded_over_thread_ranks_gives_the_single_gpu_bytes(world, case, oracle):
    from aero_amd.shard import LocalGroup
    if case == "fib_program":
        program, (log_n, opt) = aero_amd.fib_program(8, (3, 4, 2)), (12, [27, 8, 8, 4, 1, 4, 7])
        trace = aero_amd.fib_trace(8, log_n)
        pub = [int(trace[2 * k + 1][-1]) for k in range(4)]
    else:
        log_n, pairs, aux, opt = {"vm_small": (10, 2, 3, [27, 8, 8, 4, 1, 8, 6]), "vm_quadratic": (9, 4, 4, [20, 8, 8, 4, 2, 8, 6]),
                                  "vm_72_9": (12, 26, 9, [27, 8, 16, 4, 1, 4, 8])}[case]
        program = aero_amd.synth_vm_program(log_n, pairs, aux, 16 if pairs == 26 else 4)
        trace, pub = aero_amd.synth_vm_trace(log_n, pairs)
    air = aero_amd.Air(program)
    options = aero_amd.ProofOptions(*opt)
    single = aero_amd.Context(0)
    want = single.prove_air(air, single.trace_upload(trace), pub, options)
    single.close()
    ref, _ = oracle.prove_air(program, trace, pub, opt)
    assert want == ref
    g = LocalGroup(world, min_peer_digests=64)
    try:
        proofs = g.run(lambda r, ctx, comm: ctx.prove_air(air, ctx.trace_upload(trace), pub, options, comm=comm))
        stats = g.stats(0)
    finally:
        g.close()
    for r, p in enumerate(proofs):
        assert p == want, f"rank {r} of {world}"
    assert stats["all_to_all"] >= 2 and stats["bytes_sent"] > 0
    # the same from HOST memory: every rank copies its share of the main columns (+ what the auxiliary builders read)
    g = LocalGroup(world, min_peer_digests=64)
    try:
        pinned = aero_amd.PinnedTrace(trace)
        proofs = g.run(lambda r, ctx, comm: ctx.prove_air(air, pinned, pub, options, comm=comm))
    finally:
        g.close()
    for r, p in enumerate(proofs):
        assert p == want, f"host hand-over, rank {r} of {world}"


def test_trace_commitment_alone_is_the_proofs_first_commitment(oracle):
    """aero_commit_trace_sharded (the first half of the fork's commit_to_trace_and_validate, proving_worker.rs:323-332; what the exchange
    stress loop calls): on one GPU and over thread ranks of world 2 / 4 / 8 the root is bytes 24..55 of the oracle's proof, the subtree
    roots are the nodes world .. 2 world - 1 of the single-GPU tree, narrow rows (exchanged as rows) and wide ones (exchanged as digests)."""
    import numpy as np
    from aero_amd.shard import LocalGroup
    DEFAULT = [27, 8, 16, 4, 1, 8, 8]
    opt = aero_amd.ProofOptions(*DEFAULT)
    ctx = aero_amd.Context(0)
    for width, log_n in ((2, 12), (8, 10)):
        trace = aero_amd.fib_trace(width, log_n)
        want = oracle.prove_fib(width, log_n, DEFAULT)[0][24:56]
        dev = ctx.trace_upload(trace)
        root, subs = ctx.commit_trace(dev, opt)
        assert root == want and subs == [want]
        # the single-GPU tree over the same LDE: node 1 = root, nodes [world, 2 world) = the subtree roots
        lde = ctx.evaluate_columns_over(ctx.interpolate_columns(dev), 3)
        tree = ctx.merkle_commit_rows(lde)
        nodes = tree.nodes()
        assert nodes[1].tobytes() == want
        for world in (2, 4, 8):
            g = LocalGroup(world)
            try:
                res = g.run(lambda r, c, comm: c.commit_trace(c.trace_upload(trace), opt, comm=comm))
            finally:
                g.close()
            for r, (rt, sb) in enumerate(res):
                assert rt == want, (width, world, r)
                assert sb == [nodes[world + k].tobytes() for k in range(world)], (width, world, r)
        tree.free(); lde.free(); dev.free()
    ctx.close()
