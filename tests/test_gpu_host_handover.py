"""Trace handed over in HOST memory (the reference's hand-over: `Prover::prove(trace)` receives the ExecutionTrace the VM left in
host memory, aero-sdk/miden-wasm/src/proving_worker.rs:140,465-467). Wide traces travel in column groups on a second stream
behind the transforms of the previous group; the canonical-form check rides on the first inverse NTT pass. Bytes must not depend
on how the trace arrived."""
import os

import numpy as np
import pytest

import aero_amd
from tests import air_examples as ex

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def ctx():
    c = aero_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("width,log_n,aux,opt", [
    (72, 14, (0, 0, 2), [27, 8, 16, 4, 1, 8, 8]),       # 18 column groups (2^14 rows: the 32 MiB floor does not bind, 16-group cap does)
    (72, 12, (9, 16, 8), [27, 8, 8, 4, 1, 4, 7]),       # auxiliary segment: a device copy of the main segment is kept for its builders
    (40, 13, (0, 0, 2), [27, 8, 8, 4, 2, 8, 6]),        # quadratic extension
    (16, 10, (0, 0, 2), [27, 8, 8, 4, 1, 8, 6]),
    (2, 12, (1, 1, 2), [27, 8, 8, 4, 1, 8, 6]),         # narrow: one copy on the proving stream
])
def test_host_trace_gives_the_resident_bytes(ctx, oracle, width, log_n, aux, opt):
    trace = aero_amd.fib_trace(width, log_n)
    o = aero_amd.ProofOptions(*opt)
    want, pub = ctx.prove_fib_aux(ctx.trace_upload(trace), aux[0], aux[1], o, aux_degree=aux[2])
    ref, _, _ = oracle.prove_fib_aux(width, log_n, aux[0], aux[1], opt, D=aux[2]) if aux[0] else oracle.prove_fib(width, log_n, opt)
    assert want == ref
    pinned = aero_amd.PinnedTrace(trace)
    for pipeline in ("1", "0"):
        os.environ["AERO_H2D_PIPELINE"] = pipeline
        try:
            for t in (trace, pinned):
                got, got_pub = ctx.prove_fib_aux(t, aux[0], aux[1], o, aux_degree=aux[2])
                assert got == want and got_pub == pub
        finally:
            os.environ.pop("AERO_H2D_PIPELINE", None)
    pinned.release()


def test_program_air_with_aux_from_host_memory(ctx, oracle):
    b, trace, pub = ex.synth_vm(11, 13, 9)               # 46 main columns: pipelined, aux builders read the kept copy
    air = aero_amd.Air(b.to_bytes())
    opt = [27, 8, 8, 4, 1, 4, 7]
    want, _ = oracle.prove_air(b.to_bytes(), trace, pub, opt)
    assert ctx.prove_air(air, trace, pub, aero_amd.ProofOptions(*opt)) == want
    assert ctx.prove_air(air, aero_amd.PinnedTrace(trace), pub, aero_amd.ProofOptions(*opt)) == want


@pytest.mark.parametrize("width,col,row", [(72, 71, 12345), (72, 0, 0), (2, 1, 777)])
def test_non_canonical_element_in_a_host_trace_is_refused(ctx, width, col, row):
    trace = aero_amd.fib_trace(width, 14)
    good, _ = ctx.prove_fib(trace, aero_amd.ProofOptions.with_96_bit_security())
    bad = trace.copy()
    bad[col][row] = P + 3                                # found by the first inverse pass of its column group
    with pytest.raises(aero_amd.AeroError) as e:
        ctx.prove_fib(bad, aero_amd.ProofOptions.with_96_bit_security())
    assert e.value.code == -1 and "non-canonical" in str(e.value)
    again, _ = ctx.prove_fib(trace, aero_amd.ProofOptions.with_96_bit_security())     # the context is still usable
    assert again == good


def test_pinned_views_outlive_release_and_near_allocation_works(ctx):
    """PinnedTrace.array is a view over library-owned pinned memory (aero_host_alloc / aero_host_alloc_near): a view the caller still holds
    keeps the buffer alive after release(); a buffer allocated near the device proves like any other."""
    t = aero_amd.fib_trace(4, 10)
    want, _ = ctx.prove_fib(t, aero_amd.ProofOptions.with_96_bit_security())
    p = aero_amd.PinnedTrace(t, device=0)
    got, _ = ctx.prove_fib(p, aero_amd.ProofOptions.with_96_bit_security())
    assert got == want
    view, row = p.array, p.array[3]
    p.release()
    assert p.array is None and (view == t).all() and (row == t[3]).all()
    del p, view
    assert int(row[5]) == int(t[3][5])
    node = aero_amd.C.c_int32(-9)
    assert aero_amd.lib().aero_numa_device_node(aero_amd.C.c_int32(0), aero_amd.C.byref(node)) == 0 and node.value >= -1
    pool = aero_amd.Pool(0, 2)
    n, pinned = pool.placement()
    assert n == node.value and (pinned == 0 if n < 0 else pinned <= 2)
    pool.close()
