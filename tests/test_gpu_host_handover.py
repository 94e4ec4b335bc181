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
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


@pytest.mark.parametrize("prefetch_mb", ["32", "0.01"])
def test_pool_queue_of_different_traces(oracle, prefetch_mb, tmp_path):
    """aero_pool_prove_*_queue: a queue of DIFFERENT traces dealt to the slots round-robin (pool.rs:105-124), every proof back in queue
    order - fewer traces than slots, more traces than slots, with the prefetch of the next trace (threshold lowered) and without; a
    constraint program with one statement per trace; a bad trace fails the whole call and leaves the pool usable."""
    import json, subprocess, sys
    body = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
import aero_amd
from aero_amd import air as A
from tests import oracle_lib
P = 0xFFFFFFFF00000001
orc = oracle_lib.load(); orc.set_threads(16)
opt = [27, 8, 8, 4, 1, 8, 6]
o = aero_amd.ProofOptions(*opt)
pool = aero_amd.Pool(0, 3)
W, log_n = 4, 12
base = aero_amd.fib_trace(W, log_n)
rng = np.random.default_rng(5)
# different traces: the valid one, and copies with one cell changed (they violate the AIR: a prover does not care, the bytes are still a
# pure function of the trace, and the oracle is handed the same trace)
traces = [base.copy() for _ in range(8)]
for t in range(1, 8):
    traces[t][t %% W][100 + t] = int(rng.integers(1, 1 << 62))
want = [orc.prove_fib(W, log_n, opt, trace=tr)[0] for tr in traces]
assert len(set(want)) == 8
for count in (2, 3, 8):
    got = pool.prove_fib_queue([aero_amd.PinnedTrace(tr) for tr in traces[:count]], o)
    assert [g[0] for g in got] == want[:count], count
    assert all(g[1] == [int(tr[2 * k + 1][-1]) for k in range(W // 2)] for g, tr in zip(got, traces))
bad = [tr.copy() for tr in traces[:5]]
bad[3][1][7] = P + 1
try:
    pool.prove_fib_queue(bad, o)
    raise SystemExit("non-canonical element went unnoticed")
except aero_amd.AeroError as e:
    assert "non-canonical" in str(e)
assert [g[0] for g in pool.prove_fib_queue(traces[:4], o)] == want[:4]
# a program whose statement differs per trace: one doubling column started at pub(0)
n = 1 << 10
b = A.AirBuilder(1, num_pub=1)
b.transition(b.main_next(0) - 2 * b.main(0), 1)
b.assert_single(0, 0, b.pub(0))
air = aero_amd.Air(b.to_bytes())
starts = [3, 5, 7, 11, 13]
ptraces = [np.array([[pow(2, i, P) * s0 %% P for i in range(n)]], dtype=np.uint64) for s0 in starts]
pwant = [orc.prove_air(b.to_bytes(), tr, [s0], opt)[0] for tr, s0 in zip(ptraces, starts)]
assert pool.prove_air_queue(air, ptraces, [[s0] for s0 in starts], o) == pwant
pool.close()
print("ok")
''' % ROOT
    script = tmp_path / "queue.py"
    script.write_text(body)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=dict(os.environ, AERO_POOL_PREFETCH_MIN_MB=prefetch_mb), cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-2000:])
