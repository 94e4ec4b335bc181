"""The N > 1 path of bench.py on CPU: world_size-2 `gloo` processes run bench.py's own timing / reduction helpers
(barrier on both sides of exactly K steps, MAX over ranks, whole-job aggregate, rank 0 reports) around a small proof.
The step here is the CPU oracle (tests may use it); on the GPU box the step is aero_amd.Context.prove_fib."""
import json
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys, time
    sys.path.insert(0, %(root)r)
    import torch, torch.distributed as dist
    import bench
    from tests import oracle_lib

    rank, local_rank, world = bench.rank_env()
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    orc = oracle_lib.load()
    orc.set_threads(1)
    opt = [27, 8, 8, 4, 1, 8, 5]
    log_n, width, steps = 8, 2, 3
    proofs = []

    def step():
        # rank r proves its own independent trace shape (weak scaling: no data exchanged between ranks)
        proofs.append(orc.prove_fib(width, log_n, opt)[0])
        if rank == 1:
            time.sleep(0.05)          # make rank 1 the slow one: MAX over ranks must pick it up

    dt_local = bench.timed_steps(step, steps, dist.barrier)
    dt = bench.max_over_ranks(dt_local, dist, "cpu")
    value = bench.aggregate_value((1 << log_n) * width, steps, world, dt)
    gathered = [None] * world
    dist.all_gather_object(gathered, (dt_local, len(proofs), proofs[-1].hex()))
    if rank == 0:
        print(json.dumps({"dt": dt, "value": value, "per_rank": [(g[0], g[1]) for g in gathered],
                          "same_proof": gathered[0][2] == gathered[1][2], "world": world, "steps": steps}))
    dist.barrier()
    dist.destroy_process_group()
''')


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_timing_and_aggregate(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                         capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["world"] == 2 and all(n == r["steps"] for _, n in r["per_rank"])      # exactly K steps on every rank
    slow = max(d for d, _ in r["per_rank"])
    assert abs(r["dt"] - slow) < 1e-6                                               # MAX over ranks
    assert r["dt"] >= 0.15                                                          # includes rank 1's 3 x 50 ms
    assert abs(r["value"] - (256 * 2) * 3 * 2 / r["dt"]) < 1e-6                     # whole-job aggregate over both ranks
    assert r["same_proof"]                                                          # independent replicas, same input -> same bytes


def test_bench_refuses_mismatched_world():
    """Under a launcher (RANK set) a WORLD_SIZE that disagrees with --gpus is an error, not a silent fallback."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         env=dict(os.environ, WORLD_SIZE="1", RANK="0"), timeout=120)
    assert out.returncode != 0 and "torch.distributed.run" in (out.stderr + out.stdout)


CHILD = textwrap.dedent('''
    import json, os, sys
    r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
    if "--fail-rank" in sys.argv and r == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
        sys.exit(7)
    import torch.distributed as dist
    dist.init_process_group(backend="gloo", rank=r, world_size=w)          # the rendezvous the launcher set up works
    got = [None] * w
    dist.all_gather_object(got, r)
    if r == 0:
        print(json.dumps({"ranks": got, "argv": sys.argv[1:]}), flush=True)
    else:
        print("noise from rank", r)                                          # must not reach the job's stdout
    dist.barrier()
    dist.destroy_process_group()
''')


def test_self_launch_of_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher starts its own N ranks (bench.launch_ranks): fresh child processes with the
    torch.distributed.run environment contract, rank 0's stdout is the job's stdout, a failing rank fails the job."""
    sys.path.insert(0, ROOT)
    import bench
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    runner = tmp_path / "runner.py"
    runner.write_text(f"import sys; sys.path.insert(0, {ROOT!r}); import bench; sys.exit(bench.launch_ranks(3, sys.argv[1:], script={str(child)!r}, timeout_s=120))")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, str(runner), "--gpus", "3", "--steps", "5"], capture_output=True, text=True, timeout=200, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip() and not l.startswith("[Gloo]")]     # gloo logs its own line on rank 0
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["ranks"] == [0, 1, 2] and r["argv"] == ["--gpus", "3", "--steps", "5"]
    out = subprocess.run([sys.executable, str(runner), "--fail-rank", "2"], capture_output=True, text=True, timeout=200, env=env)
    assert out.returncode != 0


def test_plain_multi_gpu_invocation_takes_the_launcher_path():
    """bench.py --gpus 2 with no RANK in the environment must not exit with a usage error: it becomes the launcher. Here (no GPU)
    the ranks then fail loudly because the HIP library finds no device - which proves they were started."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], capture_output=True,
                         text=True, env=env, timeout=300)
    text = out.stderr + out.stdout
    assert "must be launched with torch.distributed.run" not in text
    import torch
    if not torch.cuda.is_available():
        assert out.returncode != 0 and ("HIP" in text or "cuda" in text.lower() or "device" in text.lower())
