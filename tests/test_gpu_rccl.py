"""The library's native RCCL communicator (aero_rccl_*, include/aero_stark.h) on the one GPU of the test box: a world of ONE
rank runs the real `ncclCommInitRank`, `ncclSend`/`ncclRecv` group, `ncclAllGather` and `ncclAllReduce` calls, enqueued on
the context's stream (RCCL refuses two ranks on one device, so world > 1 needs a multi-GPU node: bench.py measures that
there; tests/test_gpu_sharded.py checks the sharding itself at world 2/4/8 over gloo). No torch in this file: the
communicator must work in a process that holds only this library and its HIP runtime."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import aero_amd
from aero_amd.shard import RcclComm

pytestmark = pytest.mark.gpu
P = aero_amd.P

# The tests that call `ncclCommInitRank` INSIDE the test process run in a pytest child of their own (AERO_RCCL_INNER=1, started by
# test_world_of_one_in_a_process_of_its_own below). Reason (profiles/r6_suite_abort.md): once in 8 runs of this stretch of the suite a GPU
# memory fault was reported by the runtime's event thread while the main thread sat inside RCCL's initialisation - an abort() that takes the
# whole suite's process and every result with it. In a child the same event is ONE failed test with its log; it also is what a deployment
# looks like (one process per GPU creates its communicator early, not after 180 tests' worth of contexts).
INNER = os.environ.get("AERO_RCCL_INNER") == "1"
in_child = pytest.mark.skipif(not INNER, reason="runs inside test_world_of_one_in_a_process_of_its_own")


@pytest.fixture(scope="module")
def env():
    ctx = aero_amd.Context(0)
    comm = RcclComm(ctx, 0, 1)
    yield ctx, comm
    comm.close()
    ctx.close()


@in_child
def test_exchanges_run_on_the_context_stream(env):
    ctx, comm = env
    cs = comm.struct
    assert cs.rank == 0 and cs.world == 1 and cs.flags == 1          # AERO_COMM_STREAM_ORDERED
    info = comm.info()                                                  # aero_rccl_info: what RCCL itself says (-1 = this librccl lacks the query)
    assert info["world_asked_for"] == 1 and info["ranks_counted_by_rccl"] in (1, -1) and info["rank_as_rccl_numbers_it"] in (0, -1) and info["device_rccl_bound"] in (0, -1)
    rng = np.random.default_rng(5)
    n = 1 << 17
    data = rng.integers(0, P, size=(1, n), dtype=np.uint64)
    src, dst = ctx.trace_upload(data), ctx.trace_upload(np.zeros((1, n), np.uint64))
    assert cs.all_to_all(cs.user, src.device_ptr, dst.device_ptr, 8 * n) == 0, comm.error_text()
    assert (dst.download() == data).all()                               # download waits on the same stream: ordering is the stream's
    dst2 = ctx.trace_upload(np.zeros((1, n), np.uint64))
    assert cs.all_gather(cs.user, src.device_ptr, dst2.device_ptr, 8 * n) == 0, comm.error_text()
    ctx.synchronize()
    assert (dst2.download() == data).all()
    assert cs.all_reduce_sum_u64(cs.user, dst2.device_ptr, n) == 0, comm.error_text()
    assert (dst2.download() == data).all()                              # world of one: the sum is the value itself
    assert comm.calls == {"all_to_all": 1, "all_gather": 1, "all_reduce": 1}
    assert comm.bytes_sent == 8 * n                                     # nothing leaves a world of one except the all-reduce count
    for m in (src, dst, dst2):
        m.free()


@in_child
def test_world_of_one_proof_is_the_single_gpu_proof(env, oracle):
    ctx, comm = env
    opt = aero_amd.ProofOptions.with_96_bit_security()
    dev = ctx.trace_upload(aero_amd.fib_trace(2, 12))
    got, pub = ctx.prove_fib_aux(dev, 0, 0, opt, comm=comm)
    want, want_pub, _ = oracle.prove_fib(2, 12, opt.to_list())
    assert got == want and pub == want_pub
    dev.free()


@in_child
def test_bad_arguments(env):
    ctx, comm = env
    L = aero_amd.lib()
    h = C.c_void_p()
    uid = (C.c_uint8 * 128)()
    assert L.aero_rccl_create(ctx.h, C.c_int32(2), C.c_int32(2), uid, C.byref(h)) == -1      # rank out of range
    assert L.aero_rccl_create(None, C.c_int32(0), C.c_int32(1), uid, C.byref(h)) == -1
    assert L.aero_rccl_unique_id(None) == -1


@in_child
def test_pairwise_exchange_with_itself(env):
    ctx, comm = env
    cs = comm.struct
    n = 1 << 12
    data = np.arange(n, dtype=np.uint64).reshape(1, n)
    src, dst = ctx.trace_upload(data), ctx.trace_upload(np.zeros((1, n), np.uint64))
    assert cs.send_recv(cs.user, src.device_ptr, 0, dst.device_ptr, 0, 8 * n) == 0, comm.error_text()
    assert (dst.download() == data).all()


@pytest.mark.skipif(INNER, reason="this IS the child")
def test_world_of_one_in_a_process_of_its_own():
    """Runs the four in-process communicator tests above in a child pytest. A child that dies of the GPU memory fault described at the top of
    this file is kept as evidence (gpurun_out/rccl_init_fault_<n>.log), reported as a warning and retried ONCE - the fault has never been
    seen twice in a row, and nothing of this library runs between process start and `ncclCommInitRank` in the child; any other failure, and a
    second fault, fail the test with the child's output."""
    import warnings
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    last = None
    for attempt in (1, 2):
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                            "-k", "context_stream or world_of_one_proof_is or bad_arguments or pairwise_exchange"],
                           env=dict(os.environ, AERO_RCCL_INNER="1"), capture_output=True, text=True, timeout=900, cwd=root, stdin=subprocess.DEVNULL)
        last = r.stdout[-4000:] + "\n" + r.stderr[-4000:]
        if r.returncode == 0:
            assert "4 passed" in r.stdout, last
            return
        if attempt == 1 and "Memory access fault" in (r.stdout + r.stderr):
            keep = os.path.join(os.environ.get("GRAFT_REPO_ROOT", root), "gpurun_out")
            try:
                os.makedirs(keep, exist_ok=True)
                with open(os.path.join(keep, f"rccl_init_fault_{os.getpid()}.log"), "w") as f:
                    f.write(r.stdout + "\n" + r.stderr)
            except OSError:
                pass
            warnings.warn("GPU memory fault inside the RCCL child (profiles/r6_suite_abort.md); retrying once:\n" + last[-1500:])
            continue
        break
    raise AssertionError("in-process RCCL tests failed in their child:\n" + last)


RANK_WORKER = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
import aero_amd
from aero_amd.shard import RcclComm
rank, world, idfile, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
def share(uid):
    if rank == 0:
        with open(idfile + ".tmp", "wb") as f: f.write(uid)
        os.replace(idfile + ".tmp", idfile)
        return uid
    for _ in range(600):
        if os.path.exists(idfile): return open(idfile, "rb").read()
        time.sleep(0.1)
    raise SystemExit("no id from rank 0")
ctx = aero_amd.Context(rank)
comm = RcclComm(ctx, rank, world, share_id=share)
opt = aero_amd.ProofOptions.with_96_bit_security()
for (w, log_n, aux) in ((2, 14, (0, 0, 2)), (8, 13, (0, 0, 2)), (16, 12, (4, 3, 5))):
    trace = aero_amd.PinnedTrace(aero_amd.fib_trace(w, log_n))
    proof, pub = ctx.prove_fib_sharded_host(comm, trace, opt, aux)
    open(out + f".{w}.{log_n}", "wb").write(proof)
print("ok", comm.calls, comm.bytes_sent)
'''


@pytest.mark.skipif(aero_amd.device_count() < 2, reason="RCCL refuses two ranks on one device: needs >= 2 GPUs")
def test_two_ranks_over_rccl_give_the_single_gpu_proof(tmp_path):
    """world = 2 through the NATIVE communicator (ncclSend / ncclRecv groups, all-gather, all-reduce, pairwise exchange on the
    contexts' streams, AERO_COMM_STREAM_ORDERED), one process per GPU, trace in host memory."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank.py"
    script.write_text(RANK_WORKER % {"root": root})
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(tmp_path / "id"), str(tmp_path / f"proof{r}")], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, cwd=root) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    ctx = aero_amd.Context(0)
    opt = aero_amd.ProofOptions.with_96_bit_security()
    for (w, log_n, aux) in ((2, 14, (0, 0, 2)), (8, 13, (0, 0, 2)), (16, 12, (4, 3, 5))):
        want, _ = ctx.prove_fib_aux(aero_amd.fib_trace(w, log_n), aux[0], aux[1], opt, aux_degree=aux[2])
        for r in range(2):
            assert (tmp_path / f"proof{r}.{w}.{log_n}").read_bytes() == want
    ctx.close()


NEGATIVE_WORKER = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
import aero_amd
from aero_amd.shard import RcclComm
rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
def share(uid):
    path = os.path.join(d, "id")
    if rank == 0:
        with open(path + ".tmp", "wb") as f: f.write(uid)
        os.replace(path + ".tmp", path)
        return uid
    for _ in range(600):
        if os.path.exists(path): return open(path, "rb").read()
        time.sleep(0.1)
    raise SystemExit("no id from rank 0")
stage = [0]
def agree(status):            # all-gather of one integer through files: every rank reads every rank's status
    stage[0] += 1
    mine = os.path.join(d, f"st{stage[0]}.{rank}")
    with open(mine + ".tmp", "w") as f: f.write(str(int(status)))
    os.replace(mine + ".tmp", mine)
    vals = []
    for r in range(world):
        p = os.path.join(d, f"st{stage[0]}.{r}")
        for _ in range(1200):
            if os.path.exists(p): break
            time.sleep(0.1)
        else:
            raise SystemExit("peer status missing")
        vals.append(int(open(p).read()))
    return min(vals)
ctx = aero_amd.Context(0)                     # BOTH ranks on device 0
try:
    RcclComm(ctx, rank, world, share_id=share, agree=agree)
except aero_amd.AeroError as e:
    print("REFUSED", e.code, str(e)[:300], flush=True)
    ctx.close()
    sys.exit(0)
print("CREATED", flush=True)
sys.exit(3)
'''


def test_two_ranks_on_one_device_are_refused_cleanly_on_both_ranks(tmp_path):
    """RCCL refuses two ranks of one communicator on the same device. First contact with a multi-GPU node must not find out what
    a refusal does: both ranks get an AeroError (status + RCCL's text), nobody hangs, the contexts stay usable and close."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "neg.py"
    script.write_text(NEGATIVE_WORKER % {"root": root})
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              cwd=root, stdin=subprocess.DEVNULL) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240)[0])
    finally:
        for p in procs:                                # exactly the processes started here, by handle
            if p.poll() is None:
                p.kill()
                p.wait()
    assert all(p.returncode == 0 for p in procs), outs
    assert all("REFUSED" in o for o in outs), outs
