"""The library's native RCCL communicator (aero_rccl_*, include/aero_stark.h) on the one GPU of the test box: a world of ONE
rank runs the real `ncclCommInitRank`, `ncclSend`/`ncclRecv` group, `ncclAllGather` and `ncclAllReduce` calls, enqueued on
the context's stream (RCCL refuses two ranks on one device, so world > 1 needs a multi-GPU node: bench.py measures that
there; tests/test_gpu_sharded.py checks the sharding itself at world 2/4/8 over gloo)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch                      # torch first: its HIP runtime and its RCCL are the copies the process keeps
    import aero_amd
    from aero_amd.shard import RcclComm
    ctx = aero_amd.Context(0)
    comm = RcclComm(ctx, 0, 1)
    yield torch, aero_amd, ctx, comm
    comm.close()
    ctx.close()


def test_exchanges_run_on_the_context_stream(env):
    torch, aero_amd, ctx, comm = env
    cs = comm.struct
    assert cs.rank == 0 and cs.world == 1 and cs.flags == 1          # AERO_COMM_STREAM_ORDERED
    n = 1 << 20
    src = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
    dst = torch.zeros_like(src)
    torch.cuda.synchronize()
    assert cs.all_to_all(cs.user, src.data_ptr(), dst.data_ptr(), n) == 0, comm.error_text()
    ctx.synchronize()
    assert torch.equal(src, dst)
    dst.zero_()
    torch.cuda.synchronize()
    assert cs.all_gather(cs.user, src.data_ptr(), dst.data_ptr(), n) == 0, comm.error_text()
    ctx.synchronize()
    assert torch.equal(src, dst)
    vals = torch.arange(1, 1001, dtype=torch.int64, device="cuda") * (1 << 53)      # wraps in u64 when summed over > 1 rank
    want = vals.clone()
    torch.cuda.synchronize()
    assert cs.all_reduce_sum_u64(cs.user, vals.data_ptr(), vals.numel()) == 0, comm.error_text()
    ctx.synchronize()
    assert torch.equal(vals, want)                                     # world of one: the sum is the value itself
    assert comm.calls == {"all_to_all": 1, "all_gather": 1, "all_reduce": 1}
    assert comm.bytes_sent == 8 * 1000                                 # nothing leaves a world of one except the all-reduce count


def test_world_of_one_proof_is_the_single_gpu_proof(env, oracle):
    torch, aero_amd, ctx, comm = env
    opt = aero_amd.ProofOptions.with_96_bit_security()
    dev = ctx.trace_upload(aero_amd.fib_trace(2, 12))
    got, pub = ctx.prove_fib_aux(dev, 0, 0, opt, comm=comm)
    want, want_pub, _ = oracle.prove_fib(2, 12, opt.to_list())
    assert got == want and pub == want_pub
    dev.free()


def test_bad_arguments(env):
    torch, aero_amd, ctx, comm = env
    L = aero_amd.lib()
    h = C.c_void_p()
    uid = (C.c_uint8 * 128)()
    assert L.aero_rccl_create(ctx.h, C.c_int32(2), C.c_int32(2), uid, C.byref(h)) == -1      # rank out of range
    assert L.aero_rccl_create(None, C.c_int32(0), C.c_int32(1), uid, C.byref(h)) == -1
    assert L.aero_rccl_unique_id(None) == -1
