"""ONE proof sharded over several ranks (aero_prove_fib_sharded) must be byte-identical to the single-GPU proof and to
the CPU oracle's. On a 1-GPU box the ranks are processes sharing the GPU and exchange over gloo (aero_amd/shard.py: TorchComm, on host
copies by default, on the device tensors with AERO_TORCHCOMM_GLOO=device); a multi-GPU node takes the library's native RCCL communicator
(tests/test_gpu_rccl.py). Every rank leaves per-exchange fingerprints of what it sent and received; `diagnose` lines them up across ranks
when bytes differ (profiles/r6_sharded_anomaly.md). Sharded proofs run with prove-then-verify on (aero_ctx_set_self_verify, AUTO)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT = [27, 8, 16, 4, 1, 8, 8]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn_world(world, cases, tmp_path, timeout, script="shard_worker.py", argv=None):
    """One attempt: returns (ok, per-rank output). Worker output goes to files (a pipe can be kept open by helper processes)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world))
    logs = [open(tmp_path / f"rank{r}.log", "wb") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, script), str(tmp_path)] + (argv if argv is not None else [json.dumps(cases)]),
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=logs[r], stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL)
             for r in range(world)]
    import time
    t_end = time.time() + timeout
    try:
        for p in procs:
            p.wait(timeout=max(1.0, t_end - time.time()))
    except subprocess.TimeoutExpired:
        pass
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
        for f in logs:
            f.close()
    outs = [open(tmp_path / f"rank{r}.log", "rb").read().decode(errors="replace") for r in range(world)]
    return all(p.returncode == 0 for p in procs), outs


def run_world(world, cases, tmp_path, timeout=900):
    """Every rank's result of every case: [(single, per_rank_bytes, info)], info = rank 0's exchange counters + "status" (per rank: the
    call's status) + "evidence" (per rank: the exchange fingerprints, shard_worker.py)."""
    ok, outs = _spawn_world(world, cases, tmp_path, timeout)
    if not ok and any(k in o for o in outs for k in ("address already in use", "EADDRINUSE", "Connection refused", "connect() timed out")):
        ok, outs = _spawn_world(world, cases, tmp_path, timeout)     # the rendezvous port was taken between probing and binding
    assert ok, "sharded workers failed:\n" + "\n----\n".join(o[-2500:] for o in outs)
    res = []
    for i in range(len(cases)):
        single = open(tmp_path / f"case{i}.single.bin", "rb").read()
        per_rank = [open(tmp_path / f"case{i}.rank{r}.bin", "rb").read() for r in range(world)]
        info = json.load(open(tmp_path / f"case{i}.comm.json"))
        info["status"] = [json.load(open(tmp_path / f"case{i}.rank{r}.status.json")) for r in range(world)]
        info["evidence"] = [json.load(open(tmp_path / f"case{i}.rank{r}.evidence.json")) for r in range(world)]
        res.append((single, per_rank, info))
    return res


def _where(off):
    # StarkProof::to_bytes: 22 bytes of context, u16 length, then the commitments (trace, [aux], composition, FRI layers ..., remainder)
    if off < 22:
        return "the context header"
    if off < 24:
        return "the commitments' length prefix"
    return f"byte {(off - 24) % 32} of commitment #{(off - 24) // 32} (0 = trace root) or, past the roots, the openings"


def exchange_mismatches(evidence):
    """Line the ranks' exchange fingerprints up: what rank r received from rank q must be what q sent to r. Returns a list of strings."""
    bad = []
    world = len(evidence)
    if any(len(e) != len(evidence[0]) for e in evidence):
        return ["the ranks made different numbers of exchange calls: " + str([len(e) for e in evidence])]
    for k in range(len(evidence[0])):
        calls = [evidence[r][k] for r in range(world)]
        op, nbytes = calls[0]["op"], calls[0]["bytes"]
        if any(c["op"] != op or c["bytes"] != nbytes for c in calls):
            bad.append(f"call {k}: the ranks disagree on the operation: " + str([(c["op"], c["bytes"]) for c in calls]))
            continue
        for r in range(world):
            for q in range(world):
                if op == "all_to_all" and calls[r]["recv"][q] != calls[q]["send"][r]:
                    bad.append(f"call {k} (all_to_all, {nbytes} B per peer): rank {r} did not receive what rank {q} sent it")
                if op == "all_gather" and calls[r]["recv"][q] != calls[q]["send"][0]:
                    bad.append(f"call {k} (all_gather, {nbytes} B per rank): rank {r} holds a piece of rank {q} that {q} did not send")
    return bad


def diagnose(want, per_rank, info, tmp_path=None, label="sharded"):
    """What a wrong sharded proof looks like across ALL ranks: who differs from `want` and where, whether the ranks agree with each other,
    and whether every exchanged piece arrived as it was sent (so: exchange corrupted, or a rank's own data wrong). On a mismatch the
    ranks' evidence files are kept under gpurun_out/shard_evidence/ for the record."""
    lines = []
    wrong = [r for r, p in enumerate(per_rank) if p != want]
    if not wrong:
        return ""
    groups = {}
    for r, p in enumerate(per_rank):
        groups.setdefault(p, []).append(r)
    lines.append(f"{label}: ranks {wrong} of {len(per_rank)} differ from the expected bytes; ranks grouped by identical bytes: {sorted(groups.values())}")
    for r in wrong:
        p = per_rank[r]
        off = next((i for i in range(min(len(p), len(want))) if p[i] != want[i]), min(len(p), len(want)))
        lines.append(f"  rank {r}: {len(p)} bytes (expected {len(want)}), first difference at offset {off}: {_where(off)}; status {info['status'][r]}")
    mism = exchange_mismatches(info["evidence"]) if all(info["evidence"]) else ["no exchange evidence (native communicator)"]
    if mism:
        lines.append("  exchange check: " + "; ".join(mism[:12]) + (f" (+{len(mism) - 12} more)" if len(mism) > 12 else ""))
    else:
        lines.append("  exchange check: every piece arrived as its sender fingerprinted it - the damage is in a rank's OWN data (what it sent "
                     "was already wrong) or behind the exchange (tree build, openings); compare the ranks' send fingerprints with a good run's")
        roots = [next((c["send"][0] for c in e if c["op"] == "all_gather" and c["bytes"] == 32), None) for e in info["evidence"]]
        lines.append(f"  first commitment: subtree-root fingerprints per rank {roots}")
    if tmp_path is not None:
        keep = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(HERE)), "gpurun_out", "shard_evidence", label.replace(" ", "_"))
        try:
            import shutil
            os.makedirs(keep, exist_ok=True)
            for f in os.listdir(tmp_path):
                if f.endswith(".json") or f.endswith(".log"):
                    shutil.copy(os.path.join(tmp_path, f), keep)
            with open(os.path.join(keep, "diagnosis.txt"), "w") as f:
                f.write("\n".join(lines) + "\n")
            lines.append(f"  evidence kept in {keep}")
        except OSError:
            pass
    return "\n".join(lines)


def check(oracle, world, cases, tmp_path):
    for case, (single, per_rank, comm) in zip(cases, run_world(world, cases, tmp_path)):
        # ALL ranks are looked at before anything is asserted: which ranks differ, where, and whether the exchanges carried what was sent
        report = diagnose(single, per_rank, comm, tmp_path, f"world{world}_w{case['width']}_n{case['log_n']}")
        assert not report, f"world {world}: sharded proof differs from the single-GPU proof for {case}\n{report}"
        if comm["backend"].startswith("gloo"):
            assert not exchange_mismatches(comm["evidence"]), "a proof came out right although an exchanged piece did not arrive as sent"
        if case.get("aux"):
            ref = oracle.prove_fib_aux(case["width"], case["log_n"], case["aux"][0], case["aux"][1], case["options"],
                                       D=(case["aux"][2] if len(case["aux"]) > 2 else 2))[0]
        else:
            ref = oracle.prove_fib(case["width"], case["log_n"], case["options"])[0]
        assert single == ref, f"proof differs from the oracle for {case}"
        assert comm["calls"]["all_reduce"] == 1, "the opening phase must need exactly one all-reduce"
        # one digest exchange + one root all-gather per sharded commitment; extra all-gathers: the FRI un-shard, the H
        # evaluations when a shard is smaller than the constraint domain (always with degree-8 constraints), and the OOD
        # frame when the main columns divide evenly among the ranks
        extra = comm["calls"]["all_gather"] - comm["calls"]["all_to_all"]
        assert 0 <= extra <= 3
        if case["width"] % world == 0:
            assert extra >= 1
        if case.get("aux") and len(case["aux"]) > 2 and case["aux"][2] > 4 and case["options"][1] == 8:
            assert extra >= 1
        assert comm["calls"]["all_to_all"] >= (3 if case.get("aux") else 2)


CASES_SMALL = [
    {"width": 2, "log_n": 10, "options": DEFAULT},
    {"width": 2, "log_n": 10, "options": DEFAULT, "min_peer": 1},
    {"width": 4, "log_n": 12, "options": [27, 8, 16, 4, 2, 8, 8], "min_peer": 1},        # quadratic extension
    {"width": 6, "log_n": 11, "options": [20, 8, 8, 4, 1, 4, 6], "min_peer": 2},         # fold 4
    {"width": 2, "log_n": 9, "options": [16, 16, 4, 4, 1, 2, 5], "min_peer": 1},         # blowup 16, fold 2
    {"width": 72, "log_n": 10, "options": DEFAULT, "min_peer": 4},                       # Miden-width rows
    {"width": 2, "log_n": 3, "options": [4, 8, 0, 4, 1, 2, 3], "min_peer": 1},           # smallest trace
    {"width": 2, "log_n": 10, "options": DEFAULT, "min_peer": 1, "aux": [3, 2]},         # auxiliary segment
    {"width": 4, "log_n": 9, "options": [27, 8, 16, 4, 2, 8, 8], "min_peer": 1, "aux": [9, 16]},
    {"width": 2, "log_n": 9, "options": DEFAULT, "min_peer": 1, "aux": [2, 3, 8]},      # degree 8: 8 composition columns
    {"width": 4, "log_n": 8, "options": [27, 8, 16, 4, 2, 8, 8], "min_peer": 1, "aux": [3, 2, 6]},   # same over F_p^2
    {"width": 2, "log_n": 8, "options": [20, 16, 8, 4, 1, 4, 5], "min_peer": 1, "aux": [1, 1, 7]},   # blowup 16 > 8 columns
]


# every case runs at every world size
SUBSETS = {2: range(len(CASES_SMALL)), 4: range(len(CASES_SMALL)), 8: range(len(CASES_SMALL))}


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_proof_identical_small(oracle, world, tmp_path):
    check(oracle, world, [CASES_SMALL[i] for i in SUBSETS[world]], tmp_path)


def test_sharded_proof_identical_config2_shape(oracle, tmp_path):
    """2^16 rows with the DEFAULT sharding threshold (two sharded FRI layers at world 2, then the un-shard all-gather), base
    and quadratic field."""
    cases = [{"width": 2, "log_n": 16, "options": DEFAULT}, {"width": 2, "log_n": 16, "options": [27, 8, 16, 4, 2, 8, 8]},
             {"width": 72, "log_n": 12, "options": [27, 8, 16, 4, 1, 4, 8], "aux": [9, 16, 8]}]     # config-5 shape
    check(oracle, 2, cases, tmp_path)
    check(oracle, 8, [cases[0], cases[2]], tmp_path)


def test_sharded_proof_identical_full_size(oracle, tmp_path):
    """BASELINE configs[1] at full size (2^20 rows) sharded 8 ways: byte-identical to the single-GPU proof and the oracle's."""
    check(oracle, 8, [{"width": 2, "log_n": 20, "options": DEFAULT}], tmp_path)


def test_sharded_rejects_bad_world(tmp_path):
    """world must be a power of two not larger than the blowup factor; a missing callback is refused."""
    import ctypes as C
    import aero_amd
    from aero_amd.shard import CommStruct
    ctx = aero_amd.Context(0)
    trace = ctx.trace_upload(aero_amd.fib_trace(2, 6))
    opts = aero_amd.ProofOptions(*DEFAULT)
    proof, plen = aero_amd.u8p(), C.c_size_t(0)
    for rank, world in [(0, 3), (0, 16), (2, 2), (-1, 2)]:
        cs = CommStruct(rank, world, None)
        rc = aero_amd.lib().aero_prove_fib_sharded(ctx.h, C.byref(cs), trace.h, C.byref(opts), C.byref(proof), C.byref(plen), None)
        assert rc == -1, (rank, world, rc)
    trace.free()
    ctx.close()


def test_exchange_failure_is_reported_and_recoverable():
    """A failing exchange callback surfaces as AERO_E_COMM (-4); the context proves normally afterwards."""
    import ctypes as C
    import aero_amd
    from aero_amd import shard
    ctx = aero_amd.Context(0)
    trace = ctx.trace_upload(aero_amd.fib_trace(2, 8))
    opts = aero_amd.ProofOptions(*DEFAULT)
    cs = shard.CommStruct(0, 2, None, shard._A2A(lambda u, s, r, n: 1), shard._AG(lambda u, s, r, n: 1), shard._AR(lambda u, b, n: 1), 0)
    proof, plen = aero_amd.u8p(), C.c_size_t(0)
    rc = aero_amd.lib().aero_prove_fib_sharded(ctx.h, C.byref(cs), trace.h, C.byref(opts), C.byref(proof), C.byref(plen), None)
    assert rc == -4, rc
    assert b"exchange failed" in aero_amd.lib().aero_last_error(ctx.h)
    ok, _ = ctx.prove_fib(trace, opts)
    assert len(ok) > 1000
    in_use_before = ctx.memory_stats()[0]
    for _ in range(3):
        rc = aero_amd.lib().aero_prove_fib_sharded(ctx.h, C.byref(cs), trace.h, C.byref(opts), C.byref(proof), C.byref(plen), None)
        assert rc == -4
    assert ctx.memory_stats()[0] == in_use_before, "a failed proof leaked pool memory"
    trace.free()
    ctx.close()


# ---- prove-then-verify at the boundary (include/aero_stark.h: aero_ctx_set_self_verify; the reference verifies every proof before it
# leaves the process: miden-proof-generator/src/main.rs:47, aero-sdk/miden-wasm/src/proving_worker.rs:196-203) ----------------------
FAULT = {"op": "all_gather", "call": 0, "chunk": 0, "byte": 5}     # one bit of subtree root 0, in what EVERY rank gathers for the trace commitment


def test_self_verify_turns_a_corrupted_exchange_into_a_status_on_every_rank(tmp_path):
    """A faulty communicator hands every rank a damaged digest (subtree root 0 of the trace commitment). The transcripts stay in step - all
    ranks hash the same wrong top - so the proof completes; its trace root commits to a subtree that does not exist. With the default mode
    (self-verify ON for sharded proofs) every rank gets AERO_E_SELF_VERIFY (-8) and no bytes; with the check switched off the same fault
    leaves the library as AERO_OK with bytes the verifier rejects - the hole this closes."""
    import aero_amd
    base = {"width": 2, "log_n": 10, "options": DEFAULT, "min_peer": 1}
    cases = [dict(base, fault=FAULT, expect_error=-8), dict(base, fault=FAULT, self_verify=0), dict(base)]
    res = run_world(2, cases, tmp_path)
    single, per_rank, info = res[0]
    for r in range(2):
        assert info["status"][r]["code"] == -8, info["status"]
        assert "self-verify" in info["status"][r]["msg"] and "rejected" in info["status"][r]["msg"]
        assert per_rank[r] == b"", "a rejected proof must not leave the library"
    single, per_rank, info = res[1]
    assert [st["code"] for st in info["status"]] == [0, 0]
    assert per_rank[0] == per_rank[1] != single and len(per_rank[0]) > 1000
    pub = [int(aero_amd.fib_trace(2, 10)[1, -1])]
    with pytest.raises(aero_amd.AeroError):
        aero_amd.verify_fib(per_rank[0], pub, (0, 0, 2), expected_log_n=10)
    aero_amd.verify_fib(single, pub, (0, 0, 2), expected_log_n=10)
    single, per_rank, info = res[2]      # the same communicator without the fault: clean, and the check was on (auto) all along
    assert per_rank == [single, single] and [st["code"] for st in info["status"]] == [0, 0]


def test_self_verify_modes_on_one_gpu(oracle):
    """AUTO leaves single-GPU proofs unchecked (the metric's path), ON checks them (same bytes, AERO_OK); a bad mode is refused."""
    import ctypes as C
    import aero_amd
    ctx = aero_amd.Context(0)
    trace = ctx.trace_upload(aero_amd.fib_trace(4, 10))
    opts = aero_amd.ProofOptions(*DEFAULT)
    want = oracle.prove_fib(4, 10, DEFAULT)[0]
    for mode in ("auto", 1, 0):
        ctx.set_self_verify(mode)
        assert ctx.prove_fib(trace, opts)[0] == want
        assert ctx.prove_fib_aux(trace, 3, 2, opts)[0] == oracle.prove_fib_aux(4, 10, 3, 2, DEFAULT)[0]
    assert aero_amd.lib().aero_ctx_set_self_verify(ctx.h, C.c_int32(2)) == -1
    assert aero_amd.lib().aero_ctx_set_self_verify(ctx.h, C.c_int32(-2)) == -1
    # a program AIR under the check
    program = aero_amd.fib_program(4)
    air = aero_amd.Air(program)
    ctx.set_self_verify(1)
    t = aero_amd.fib_trace(4, 10)
    pub = [int(t[1, -1]), int(t[3, -1])]
    got = ctx.prove_air(air, t, pub, opts)
    ctx.set_self_verify(0)
    assert got == ctx.prove_air(air, t, pub, opts)
    trace.free()
    ctx.close()
    # a pool's slots under the check (aero_pool_set_self_verify): same bytes from every slot; a bad mode is refused
    pool = aero_amd.Pool(0, 2)
    pool.set_self_verify(1)
    hosts = [aero_amd.PinnedTrace(aero_amd.fib_trace(4, 10)) for _ in range(2)]
    for p, _ in pool.prove_fib_host(hosts, opts, rounds=2):
        assert p == want
    pool.set_self_verify("auto")
    assert aero_amd.lib().aero_pool_set_self_verify(pool.h, C.c_int32(7)) == -1
    for h in hosts:
        h.release()
    pool.close()
