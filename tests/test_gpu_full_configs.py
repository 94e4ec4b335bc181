"""BASELINE.json configs at FULL size on the GPU (the small shapes of the same configs are in test_gpu_parity.py,
test_gpu_aux.py and test_gpu_sharded.py):

  configs[3]  2^24-row Fibonacci trace — one GPU and sharded 8 ways (ranks share the test box's one GPU and exchange over
              gloo; the exchange code path is the one a multi-GPU node runs over RCCL): proof bytes identical to the CPU
              oracle's on every rank;
  configs[4]  Miden's SHAPE (72 main + 9 auxiliary columns from 16 coin elements, degree-8 constraints => 8 composition
              columns, FRI folding factor 4) at 2^22 rows on the stand-in AIR (the Miden AIR is absent from the reference
              mount): prove -> verify with the OOD constraint check by the oracle's verifier and by the library's own; the
              same shape at 2^18 rows is compared byte for byte with the oracle's prover.

The reference's own counterpart is the prove-then-verify self check of miden-proof-generator/src/main.rs:31-47 and the
`prove` vs `prove_sequential` A/B pair of aero-sdk/miden-wasm/src/proving_worker.rs:124-223 / :441-518.
"""
import os

import pytest

import aero_amd

pytestmark = pytest.mark.gpu

DEFAULT = [27, 8, 16, 4, 1, 8, 8]
MIDEN_SHAPE = [27, 8, 16, 4, 1, 4, 8]        # FRI folding factor 4 (BASELINE configs[4])


@pytest.fixture(scope="module")
def ctx():
    c = aero_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def oracle_2p24(oracle):
    """The oracle's proof of the 2^24 x 2 trace (about half a minute of host time), shared by the tests below."""
    proof, pub, _ = oracle.prove_fib(2, 24, DEFAULT)
    return proof, pub


def test_config4_2p24_single_gpu_bytes_identical(ctx, oracle, oracle_2p24):
    log_n, width = 24, 2
    dev = ctx.trace_upload(aero_amd.fib_trace(width, log_n))
    got, pub = ctx.prove_fib(dev, aero_amd.ProofOptions(*DEFAULT))
    dev.free()
    want, want_pub = oracle_2p24
    assert pub == want_pub
    assert got == want, "2^24-row proof differs from the oracle's"
    oracle.verify(got, pub, air_kind=1, W=width, log_n=log_n)
    aero_amd.verify_fib(got, pub, (0, 0, 2), expected_log_n=log_n, require_options=aero_amd.ProofOptions(*DEFAULT))


def test_config4_2p24_sharded_8_ways_bytes_identical(oracle, oracle_2p24, tmp_path):
    from tests.test_gpu_sharded import diagnose, exchange_mismatches, run_world
    case = {"width": 2, "log_n": 24, "options": DEFAULT}
    (single, per_rank, comm), = run_world(8, [case], tmp_path, timeout=1500)
    want, _ = oracle_2p24
    assert single == want, "single-GPU 2^24 proof differs from the oracle's"
    # all eight ranks are compared (and the exchange fingerprints lined up) BEFORE anything is asserted: a failure names the ranks, the
    # place in the proof and whether an exchanged piece arrived damaged (profiles/r5_sharded_anomaly.md had none of that)
    report = diagnose(want, per_rank, comm, tmp_path, "config4_2p24_world8")
    assert not report, "sharded 2^24 proof differs from the oracle's\n" + report
    if comm["backend"].startswith("gloo"):
        assert not exchange_mismatches(comm["evidence"])
    assert comm["calls"]["all_reduce"] == 1 and comm["calls"]["all_to_all"] >= 2


@pytest.mark.skipif(not os.environ.get("AERO_SHARD_STRESS"), reason="on request: AERO_SHARD_STRESS='log_n:iters:block[:forms[:world]];...'")
def test_config4_exchange_stress_loop(tmp_path):
    """The decisive A/B for the one wrong sharded proof of round 5 (profiles/r6_sharded_anomaly.md): the trace commitment of the 8-way
    proof, hundreds of times per gloo form, inside a test session that has already run the suite's prefix (this module comes third;
    AERO_TEST_STOP_AFTER=test_gpu_full_configs ends the session here). Every rank's record is merged into gpurun_out/r6_stress/."""
    import json
    import shutil
    from tests.test_gpu_sharded import _spawn_world
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "r6_stress")
    os.makedirs(out, exist_ok=True)
    total_wrong = 0
    for n_spec, spec in enumerate(os.environ["AERO_SHARD_STRESS"].split(";")):
        f = spec.split(":")
        log_n, iters, block = int(f[0]), int(f[1]), int(f[2])
        forms = f[3] if len(f) > 3 else "device,host"
        world = int(f[4]) if len(f) > 4 else 8
        d = tmp_path / f"spec{n_spec}"
        d.mkdir()
        ok, outs = _spawn_world(world, None, d, 3600, script="shard_stress_worker.py", argv=[str(log_n), str(iters), str(block), forms])
        assert ok, "stress workers failed:\n" + "\n----\n".join(o[-1500:] for o in outs)
        ranks = [json.load(open(d / f"stress.rank{r}.json")) for r in range(world)]
        summary = {"spec": spec, "world": world, "log_n": log_n, "per_form": {}, "failures": []}
        for form in forms.split(","):
            st = [r["stats"][form] for r in ranks]
            summary["per_form"][form] = {"iterations": st[0]["iters"], "wrong_by_rank": [x["wrong"] for x in st], "secs_rank0": round(st[0]["secs"], 1)}
            total_wrong += sum(x["wrong"] for x in st)
        for r in ranks:
            summary["failures"] += r["failures"]
        with open(os.path.join(out, f"stress_{n_spec}_n{log_n}_w{world}.json"), "w") as fh:
            json.dump(summary, fh, indent=1)
        print("exchange stress:", json.dumps({k: v for k, v in summary.items() if k != "failures"}), "failures:", len(summary["failures"]))
    assert total_wrong == 0, f"{total_wrong} wrong trace commitments - see gpurun_out/r6_stress/"


def test_config5_standin_2p22_verifies(ctx, oracle):
    log_n, W, A, R, D = 22, 72, 9, 16, 8
    dev = ctx.trace_upload(aero_amd.fib_trace(W, log_n))
    got, pub = ctx.prove_fib_aux(dev, A, R, aero_amd.ProofOptions(*MIDEN_SHAPE), aux_degree=D)
    again, _ = ctx.prove_fib_aux(dev, A, R, aero_amd.ProofOptions(*MIDEN_SHAPE), aux_degree=D)
    dev.free()
    assert again == got, "non-deterministic proof bytes"
    # header: 72 main columns, 9 aux columns, 16 aux random elements, 2^22 rows; options incl. fold 4
    assert got[:4] == bytes([W, A, R, log_n]) and got[15:22] == bytes(MIDEN_SHAPE)
    oracle.verify_fib_aux(got, pub, W, log_n, A, R, D=D)      # every check of src/stark_verifier + the OOD constraint check
    aero_amd.verify_fib(got, pub, (A, R, D), expected_log_n=log_n)


def test_config5_standin_2p18_bytes_identical(ctx, oracle):
    log_n, W, A, R, D = 18, 72, 9, 16, 8
    dev = ctx.trace_upload(aero_amd.fib_trace(W, log_n))
    got, pub = ctx.prove_fib_aux(dev, A, R, aero_amd.ProofOptions(*MIDEN_SHAPE), aux_degree=D)
    dev.free()
    want, want_pub, _ = oracle.prove_fib_aux(W, log_n, A, R, MIDEN_SHAPE, D=D)
    assert pub == want_pub
    assert got == want, "config-5-shaped proof differs from the oracle's"


# ---- configs[4] through the AIR-as-data path: a VM-SHAPED program (20 + 2 * 26 = 72 main columns, 9 auxiliary running products
# over 16 random elements, 76 main + 9 auxiliary transition constraints in > 10 degree groups up to degree 8, periodic columns,
# two exemptions, first / last / interior / periodic assertions) instead of the hard-wired stand-in ------------------------------
def test_config5_vm_shaped_program_2p18_bytes_identical(ctx, oracle):
    log_n, pairs, A, R = 18, 26, 9, 16
    program = aero_amd.synth_vm_program(log_n, pairs, A, R)
    trace, pub = aero_amd.synth_vm_trace(log_n, pairs)
    air = aero_amd.Air(program)
    info = air.info()
    assert info["main_width"] == 72 and info["aux_width"] == 9 and info["main_transition"] == 76 and info["aux_transition"] == 9
    got = ctx.prove_air(air, trace, pub, aero_amd.ProofOptions(*MIDEN_SHAPE))
    want, _ = oracle.prove_air(program, trace, pub, MIDEN_SHAPE)
    assert got == want, "VM-shaped program proof differs from the oracle's"
    aero_amd.verify_air(got, pub, air, expected_log_n=log_n)


def test_config5_vm_shaped_program_2p22_verifies(ctx, oracle):
    log_n, pairs, A, R = 22, 26, 9, 16
    program = aero_amd.synth_vm_program(log_n, pairs, A, R)
    trace, pub = aero_amd.synth_vm_trace(log_n, pairs)
    air = aero_amd.Air(program)
    dev = ctx.trace_upload(trace)
    got = ctx.prove_air(air, dev, pub, aero_amd.ProofOptions(*MIDEN_SHAPE))
    dev.free()
    again = ctx.prove_air(air, trace, pub, aero_amd.ProofOptions(*MIDEN_SHAPE))      # handed over in host memory this time
    assert again == got, "resident and host hand-over proofs differ"
    assert got[:4] == bytes([72, A, R, log_n]) and got[15:22] == bytes(MIDEN_SHAPE)
    oracle.verify_air(got, program, pub, log_n)               # every check of src/stark_verifier + the OOD constraint check over the program
    aero_amd.verify_air(got, pub, air, expected_log_n=log_n)
    with pytest.raises(aero_amd.AeroError):
        aero_amd.verify_air(got, pub[:-1] + [pub[-1] ^ 1], air, expected_log_n=log_n)


def test_config5_vm_shaped_program_2p22_sharded_8_ways_from_host(ctx):
    # configs[4] "... 8 x MI355X": ONE proof of the VM-shaped program over 8 ranks (threads of this process sharing the test GPU,
    # exchanges = the library's local group), every rank copying only its 9 of the 72 main columns (+ the ones the auxiliary builders
    # read) from host memory; every rank's bytes = the single-GPU proof (which test_config5_vm_shaped_program_2p22_verifies verifies)
    from aero_amd.shard import LocalGroup
    log_n, pairs, A, R = 22, 26, 9, 16
    program = aero_amd.synth_vm_program(log_n, pairs, A, R)
    trace, pub = aero_amd.synth_vm_trace(log_n, pairs)
    air = aero_amd.Air(program)
    opt = aero_amd.ProofOptions(*MIDEN_SHAPE)
    want = ctx.prove_air(air, trace, pub, opt)
    aero_amd.verify_air(want, pub, air, expected_log_n=log_n)
    pinned = aero_amd.PinnedTrace(trace)
    del trace
    g = LocalGroup(8)
    try:
        proofs = g.run(lambda r, c, comm: c.prove_air(air, pinned, pub, opt, comm=comm))
        sent = g.stats(0)["bytes_sent"]
    finally:
        g.close()
        pinned.release()
    for r, p in enumerate(proofs):
        assert p == want, f"rank {r} of 8"
    assert sent < 4.0e9
