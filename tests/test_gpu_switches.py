"""Every switch of the library that selects a kernel or a plan, and every plan the library selects by SHAPE, against the oracle.

Two kinds of selection exist inside a prover whose contract is bit-exactness:
 * by shape - which NTT pass kernels a transform of a given size and width runs through (two-phase contiguous passes, one- and
   two-lane register passes, buffer-descriptor or pointer addressing, 256- or 512-thread LDS tiles, the two-launch inverse plan):
   `test_every_ntt_plan_is_reached` runs shapes chosen to reach each of them, checks by NAME (AERO_NTT_NAMES=1 labels the launches
   per plan) that each was reached, and compares the values with the oracle;
 * by environment variable - the fallbacks and alternatives listed in tools/README.md ("switches"): one test id per switch here
   or in tests/test_gpu_fallback_paths.py / test_gpu_host_handover.py / test_gpu_sharded_local.py / test_gpu_air.py.
Each case runs in its own process: the switches are read once, when the library meets them."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HEAD = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import aero_amd
from tests import oracle_lib
from tests import air_examples as ex
P = 0xFFFFFFFF00000001
orc = oracle_lib.load()
orc.set_threads(16)
ctx = aero_amd.Context(0)
'''


def run(tmp_path, body, env, timeout=900):
    script = tmp_path / "worker.py"
    script.write_text(HEAD % {"root": ROOT} + body)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **env), cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (env, r.stdout[-800:], r.stderr[-2000:])
    return r.stdout


# ---- plans selected by shape ------------------------------------------------------------------------------------------------------
PLANS = r'''
SHAPES = json.loads(os.environ.get("AERO_TEST_SHAPES", "null"))
# (log_n, columns, log_blowup): interpolate + LDE; columns beyond the first two repeat them, so that the oracle transforms two columns
# whatever the width and every other column is checked against its twin on the device
shapes = SHAPES or [(3, 2, 3), (8, 2, 1), (10, 4, 3), (12, 2, 3), (13, 2, 3), (16, 2, 3), (18, 1, 3), (20, 2, 3), (20, 72, 3), (21, 1, 3), (22, 1, 3), (22, 2, 1), (14, 128, 1)]
seen = {}
for log_n, cols, lb in shapes:
    rng = np.random.default_rng(77 + log_n + cols)
    base = (rng.integers(0, 1 << 63, (min(cols, 2), 1 << log_n), dtype=np.uint64) % np.uint64(P)).astype(np.uint64)
    trace = np.concatenate([base[c % len(base)][None, :] for c in range(cols)])
    dev = ctx.trace_upload(trace)
    ctx.set_kernel_timing(True)
    polys = ctx.interpolate_columns(dev)
    lde = ctx.evaluate_columns_over(polys, lb)
    names = sorted(ctx.kernel_timing_report())
    ctx.set_kernel_timing(False)
    seen["%dx%d/%d" % (log_n, cols, lb)] = names
    got = lde.download()
    for c in range(len(base)):
        want = orc.lde(orc.intt(base[c]), 1 << lb)
        assert (got[c] == want).all(), ("LDE differs from the oracle", log_n, cols, lb, c)
    for c in range(len(base), cols):
        assert (got[c] == got[c % len(base)]).all(), ("column differs from its twin", log_n, cols, lb, c)
    lde.free(); polys.free(); dev.free()
print(json.dumps(seen))
print("ok")
'''

# every per-plan launch label of ntt.hip that the default configuration can produce (the LDS-only forward passes run under
# AERO_NTT_REG=0 and the 12-bit forward first pass of a blowup-8 LDE under AERO_NTT_2PHASE=0: tests/test_gpu_fallback_paths.py)
EXPECTED = {
    "ntt_fwd_first8",            # two-phase contiguous first pass of a blowup-8 LDE
    "ntt_fwd_first8w",           # the same from 16 columns on: one tile and 4 K columns per workgroup, boundary factors generated once per tile
    "ntt_fwd_first_512", "ntt_fwd_first_256",      # LDS-round first pass (small transforms, other blowups), 512 / 256 threads per tile
    "ntt_fwd_reg6_mid_buf", "ntt_fwd_reg6_last_buf", "ntt_fwd_reg6_last",      # two-lane radix 64: buffer form / pointer form from 16 columns on
    "ntt_fwd_reg7",              # two-lane radix 128 (2^24- and 2^25-point transforms)
    "ntt_fwd_reg5", "ntt_fwd_reg4", "ntt_fwd_reg123",
    "ntt_inv_last11",            # contiguous last pass, 2048-point tiles (launches of >= 2^21 elements)
    "ntt_inv_last12",            # contiguous last pass, 4096-point tiles (smaller launches of transforms of >= 2^13 points)
    "ntt_inv_lds_512", "ntt_inv_lds_strided_512",          # LDS rounds (transforms of <= 2^12 points); the strided half of the two-launch plan of 2^18..2^20-point small launches
    "ntt_inv_reg6", "ntt_inv_reg5", "ntt_inv_reg4", "ntt_inv_reg123",
}


def test_every_ntt_plan_is_reached(tmp_path):
    out = run(tmp_path, PLANS, {"AERO_NTT_NAMES": "1"}, timeout=1500)
    seen = json.loads(out.strip().split("\n")[-2])
    reached = set(n for names in seen.values() for n in names if n.startswith("ntt_"))
    assert EXPECTED <= reached, ("plans never reached by the shapes above", sorted(EXPECTED - reached), seen)
    assert reached <= EXPECTED, ("launch labels this test does not know", sorted(reached - EXPECTED), seen)


# the LDS-only plans (AERO_NTT_REG=0: no register passes, no two-phase passes) on launches large enough for their 256-thread tiles,
# which no default plan reaches any more: 2^22 and 2^23 elements, 72 columns
def test_ntt_lds_only_plans_on_large_launches(tmp_path):
    out = run(tmp_path, PLANS, {"AERO_NTT_NAMES": "1", "AERO_NTT_REG": "0", "AERO_TEST_SHAPES": "[[20, 4, 3], [14, 72, 3], [21, 1, 3], [16, 2, 3]]"}, timeout=1500)
    seen = json.loads(out.strip().split("\n")[-2])
    reached = set(n for names in seen.values() for n in names if n.startswith("ntt_"))
    assert reached == {"ntt_fwd_lds_256", "ntt_fwd_lds_512", "ntt_inv_lds_256", "ntt_inv_lds_512", "ntt_inv_lds_strided_256", "ntt_inv_lds_strided_512"}, seen


def test_ntt_without_two_phase_passes_on_large_launches(tmp_path):
    """AERO_NTT_2PHASE=0: the 12-bit LDS-round first pass in front of the register passes (the forward direction's fallback)."""
    out = run(tmp_path, PLANS, {"AERO_NTT_NAMES": "1", "AERO_NTT_2PHASE": "0", "AERO_TEST_SHAPES": "[[20, 4, 3], [14, 72, 3], [16, 2, 3]]"}, timeout=1500)
    seen = json.loads(out.strip().split("\n")[-2])
    reached = set(n for names in seen.values() for n in names if n.startswith("ntt_fwd"))
    assert "ntt_fwd_first8" not in reached and {"ntt_fwd_first_256", "ntt_fwd_first_512"} <= reached, seen


@pytest.mark.parametrize("flag", ["1", "0"])
def test_ntt_first_pass_shared_boundary_factors(tmp_path, flag):
    """AERO_NTT_F8W=0: wide launches of the two-phase first pass keep one column per wave with the two-multiplication progression instead of
    sharing the tile's boundary factors through LDS (ntt_fwd_first_pass_8w). Widths with K = 4, 5, 9, 9 columns per wave, one that does not
    split (44 = 4 x 11: no K <= 9) and one below the threshold; every column against the oracle or its twin."""
    shapes = "[[10, 16, 3], [13, 20, 3], [14, 36, 3], [12, 44, 3], [20, 72, 3], [17, 12, 3]]"
    out = run(tmp_path, PLANS, {"AERO_NTT_NAMES": "1", "AERO_NTT_F8W": flag, "AERO_TEST_SHAPES": shapes}, timeout=1500)
    seen = json.loads(out.strip().split("\n")[-2])
    for shape, names in seen.items():
        wide = shape.split("/")[0] in ("10x16", "13x20", "14x36", "20x72") and flag == "1"
        assert ("ntt_fwd_first8w" in names) == wide and ("ntt_fwd_first8" in names) == (not wide), (shape, names)


# ---- switches -----------------------------------------------------------------------------------------------------------------
HASH_FORMS = r'''
import hashlib
rng = np.random.default_rng(11)
for width in (1, 2, 7, 8, 9, 15, 16, 17, 23, 24, 25, 31, 32, 33, 47, 48, 49, 72, 81):
    rows = (rng.integers(0, 1 << 63, (777, width), dtype=np.uint64) % np.uint64(P)).astype(np.uint64)
    got = ctx.hash_rows(rows)
    assert (got == orc.hash_rows(np.ascontiguousarray(rows.T))).all(), width
    blob = b"".join(int(v).to_bytes(8, "little") + bytes(24) for v in rows[776])
    assert got[776].tobytes() == hashlib.blake2s(blob).digest(), width
# a launch of 2^16 rows: a 9-column matrix through the hashing seam ...
big = (rng.integers(0, 1 << 63, (9, 1 << 16), dtype=np.uint64) % np.uint64(P)).astype(np.uint64)
m = ctx.trace_upload(big)
assert (ctx.hash_matrix_rows(m) == orc.hash_rows(big)).all()
m.free()
# ... and FRI layers of 2^17 rows, fold 8 and fold 4, inside whole proofs
for width, log_n, opt in ((2, 17, [27, 8, 16, 4, 1, 8, 8]), (2, 16, [27, 8, 16, 4, 1, 4, 8])):
    want = orc.prove_fib(width, log_n, opt)[0]
    dev = ctx.trace_upload(aero_amd.fib_trace(width, log_n))
    assert ctx.prove_fib(dev, aero_amd.ProofOptions(*opt))[0] == want, (width, log_n)
    dev.free()
print("ok")
'''


def test_row_hashes_at_many_widths_and_sizes(tmp_path):
    """The row-hash kernels by shape: widths around every even / odd / chunk boundary against the oracle and hashlib, a 2^16-row matrix through
    the hashing seam, FRI layers of 2^17 rows (fold 8) and 2^16 (fold 4) inside whole proofs. (Round 6 tried three other forms of these kernels -
    pipelined column loads, two rows per lane, R rows per thread - on this test; none was faster, none is kept: profiles/r6_hash_forms.md.)"""
    run(tmp_path, HASH_FORMS, {})


R128 = r'''
for log_n in (21, 22):
    rng = np.random.default_rng(log_n)
    a = (rng.integers(0, 1 << 63, (1, 1 << log_n), dtype=np.uint64) % np.uint64(P)).astype(np.uint64)
    ctx.set_kernel_timing(True)
    lde = ctx.evaluate_columns_over(ctx.interpolate_columns(ctx.trace_upload(a)), 3)
    names = set(ctx.kernel_timing_report())
    ctx.set_kernel_timing(False)
    assert ("ntt_fwd_reg7" in names) == (os.environ.get("AERO_NTT_R128") != "0"), names
    assert (lde.download()[0] == orc.lde(orc.intt(a[0]), 8)).all()
    lde.free()
print("ok")
'''


@pytest.mark.parametrize("flag", ["1", "0"])
def test_ntt_r128(tmp_path, flag):
    """AERO_NTT_R128=0: 2^24 / 2^25-point transforms take radix <= 64 passes (one pass more) instead of the two-lane radix-128 pass."""
    run(tmp_path, R128, {"AERO_NTT_R128": flag, "AERO_NTT_NAMES": "1"})


POOL = r'''
opt = [27, 8, 8, 4, 1, 8, 6]
o = aero_amd.ProofOptions(*opt)
pool = aero_amd.Pool(0, 3)
for width, log_n, aux in ((8, 14, (0, 0, 2)), (8, 12, (3, 4, 4)), (72, 13, (0, 0, 2))):
    trace = aero_amd.fib_trace(width, log_n)
    want = orc.prove_fib_aux(width, log_n, aux[0], aux[1], opt, D=aux[2])[0] if aux[0] else orc.prove_fib(width, log_n, opt)[0]
    hosts = [aero_amd.PinnedTrace(trace.copy()) for _ in range(3)]
    for rounds in (1, 2, 5):
        for p, _ in pool.prove_fib_host(hosts, o, aux, rounds=rounds):
            assert p == want, ("pool proof differs from the oracle", width, log_n, aux, rounds)
    # a non-canonical element in ONE slot's trace: that batch fails, the pool stays usable
    hosts[1].array[width - 1][123] = P + 5
    try:
        pool.prove_fib_host(hosts, o, aux, rounds=3)
        raise SystemExit("non-canonical element went unnoticed")
    except aero_amd.AeroError as e:
        assert "non-canonical" in str(e), str(e)
    hosts[1].array[...] = trace
    for p, _ in pool.prove_fib_host(hosts, o, aux, rounds=3):
        assert p == want
    for h in hosts: h.release()
# a program AIR with auxiliary builders through the pool (the builders read main columns of the landing buffer again)
b, trace, pub = ex.synth_vm(10, 6, 5)
air = aero_amd.Air(b.to_bytes())
vopt = [27, 8, 8, 4, 1, 4, 6]
want, _ = orc.prove_air(b.to_bytes(), trace, pub, vopt)
hosts = [aero_amd.PinnedTrace(trace.copy()) for _ in range(3)]
for rounds in (1, 4):
    for p in pool.prove_air(air, hosts, pub, aero_amd.ProofOptions(*vopt), rounds=rounds):
        assert p == want, "pool program proof differs from the oracle"
for h in hosts: h.release()
pool.close()
print("ok")
'''


@pytest.mark.parametrize("min_mb", ["0", "32", "1"])
def test_pool_prefetch(tmp_path, min_mb):
    """AERO_POOL_PREFETCH_MIN_MB: traces of at least that many MiB are copied one round ahead into a second landing buffer while the
    slot proves the current one (the default, 32, leaves these small traces on the in-stream copy; 1 sends them through the prefetch,
    0 switches it off). Bytes against the oracle, with and without an auxiliary segment, for a constraint program, and the
    canonical-form check of a landed trace."""
    run(tmp_path, POOL, {"AERO_POOL_PREFETCH_MIN_MB": min_mb})


TUNE = r'''
b, trace, pub = ex.synth_vm(10, 6, 5)
air = aero_amd.Air(b.to_bytes())
for vopt in ([27, 8, 8, 4, 1, 4, 6], [20, 8, 4, 4, 2, 8, 5]):
    want, _ = orc.prove_air(b.to_bytes(), trace, pub, vopt)
    ctx.set_kernel_timing(True)
    got = ctx.prove_air(air, trace, pub, aero_amd.ProofOptions(*vopt))
    assert "air_jit_kernel" in set(ctx.kernel_timing_report()), "the run-time compiled kernel did not run"
    ctx.set_kernel_timing(False)
    assert got == want
print("ok")
'''


@pytest.mark.parametrize("tune", ["", "wide=0", "early=0,rows=1", "barrier=1,rows=4,minblocks=1", "rows=2,wide=0,early=0"])
def test_air_jit_tune(tmp_path, tune):
    """AERO_AIR_JIT_TUNE: the code-generation variants of the run-time compiled constraint kernel (air_jit.hip: Tune)."""
    env = {"AERO_AIR_JIT_TUNE": tune, "AERO_AIR_JIT_CACHE": str(tmp_path / "cache")} if tune else {"AERO_AIR_JIT_CACHE": str(tmp_path / "cache")}
    run(tmp_path, TUNE, env)


GUARD = r'''
for width, log_n, aux, opt in ((2, 12, (0, 0, 2), [27, 8, 8, 4, 1, 8, 6]), (6, 10, (2, 3, 4), [20, 8, 4, 4, 2, 4, 5])):
    want = orc.prove_fib_aux(width, log_n, aux[0], aux[1], opt, D=aux[2])[0] if aux[0] else orc.prove_fib(width, log_n, opt)[0]
    for src in (aero_amd.fib_trace(width, log_n), ctx.trace_upload(aero_amd.fib_trace(width, log_n))):
        got, _ = ctx.prove_fib_aux(src, aux[0], aux[1], aero_amd.ProofOptions(*opt), aux_degree=aux[2])
        assert got == want
print("ok")
'''


def test_pool_guard_allocator(tmp_path):
    """AERO_POOL_GUARD=1: every device block of the context's allocator between unmapped guard pages (diagnosis mode of round 4)."""
    run(tmp_path, GUARD, {"AERO_POOL_GUARD": "1"})


GENERAL = r'''
log_n = int(os.environ["AERO_TEST_LOG_N"])
for name, (b, trace, pub), nrand in (("v2", ex.v2_air(log_n), 4), ("chain", ex.general_chain_air(log_n), 2)):
    program = b.to_bytes()
    air = aero_amd.Air(program)
    info = air.info()
    n, A_ = 1 << log_n, info["aux_width"]
    for ext in ((1, 2) if log_n < 16 else (1,)):      # the long case once per program: the oracle's proof is what takes the time
        opt = [27, 8, 8, 4, ext, 8, 6]
        want, _ = orc.prove_air(program, trace, pub, opt, keep_artifacts=True)
        rands = orc.artifact("aux_rands", ext * nrand)
        dev = ctx.trace_upload(trace)
        ctx.set_kernel_timing(True)
        auxm = ctx.aux_columns_program(air, dev, pub, rands, ext)
        names = set(ctx.kernel_timing_report())
        ctx.set_kernel_timing(False)
        assert ("air_general_column_kernel" in names) == (os.environ.get("AERO_AIR_GENERAL_DEVICE") == "1"), names
        assert (auxm.download() == orc.artifact("aux_cols", A_ * ext * n).reshape(A_ * ext, n)).all(), (name, ext, "auxiliary columns differ from the oracle")
        for src in ((dev, trace) if log_n < 16 else (dev,)):      # resident, and from host memory (the builders read the kept copy)
            assert ctx.prove_air(air, src, pub, aero_amd.ProofOptions(*opt)) == want, (name, ext)
        auxm.free(); dev.free()
print("ok")
'''


@pytest.mark.parametrize("log_n,device", [(6, "1"), (10, "1"), (14, "1"), (10, "0"), (18, "0")])
def test_general_aux_recurrences(tmp_path, log_n, device):
    """General (non-scannable) auxiliary recurrences (the reference builds its auxiliary columns inside commit_to_trace_and_validate,
    proving_worker.rs:323-332): the serial host evaluation in the middle of a device proof (the default: a dependent chain costs a host
    core nanoseconds per link and a lone wavefront a microsecond, profiles/r5_general_recurrence.md), at 2^18 rows too, and
    AERO_AIR_GENERAL_DEVICE=1, one wavefront per column on the device (air_general_column_kernel) - against the oracle's auxiliary columns
    and proof bytes, in both fields, for two programs (one with three general columns that read each other)."""
    run(tmp_path, GENERAL, {"AERO_TEST_LOG_N": str(log_n), "AERO_AIR_GENERAL_DEVICE": device}, timeout=1500)


POLL = r'''
for width, log_n, aux, opt in ((2, 16, (0, 0, 2), [27, 8, 16, 4, 1, 8, 8]), (6, 12, (2, 3, 4), [20, 8, 4, 4, 2, 4, 5]), (72, 12, (0, 0, 2), [27, 8, 8, 4, 1, 8, 6])):
    want = orc.prove_fib_aux(width, log_n, aux[0], aux[1], opt, D=aux[2])[0] if aux[0] else orc.prove_fib(width, log_n, opt)[0]
    dev = ctx.trace_upload(aero_amd.fib_trace(width, log_n))
    for _ in range(3):
        got, _ = ctx.prove_fib_aux(dev, aux[0], aux[1], aero_amd.ProofOptions(*opt), aux_degree=aux[2])
        assert got == want
pool = aero_amd.Pool(0, 4)
node, pinned = pool.placement()
assert (node, pinned) == (-1, 0) if os.environ.get("AERO_NUMA") == "0" else (node >= -1 and pinned <= 4), (node, pinned)
trace = aero_amd.fib_trace(2, 14)
want = orc.prove_fib(2, 14, [27, 8, 16, 4, 1, 8, 8])[0]
for p, _ in pool.prove_fib([pool.ctx(i).trace_upload(trace) for i in range(4)], aero_amd.ProofOptions(27, 8, 16, 4, 1, 8, 8), rounds=3):
    assert p == want
pool.close()
print("ok")
'''


@pytest.mark.parametrize("env", [{"AERO_POLL_FLAGS": "0"}, {"AERO_POLL_FLAGS": "1"}, {"AERO_NUMA": "0"}])
def test_completion_word_and_placement_switches(tmp_path, env):
    """AERO_POLL_FLAGS=0: the host waits for the stream at the tree roots and the FRI tail instead of polling the completion word the
    producing launch stores in mapped pinned memory; AERO_NUMA=0: no thread binding, no node preference for pinned buffers."""
    run(tmp_path, POLL, env)


CONS_TABLE = r'''
# four shapes on one context (it keeps the two used last), each proved three times (first proof builds the table)
for width, log_n, aux, opt in ((2, 16, (0, 0, 2), [27, 8, 16, 4, 1, 8, 8]), (6, 12, (2, 3, 4), [20, 8, 4, 4, 2, 4, 5]), (2, 13, (0, 0, 2), [27, 4, 8, 4, 1, 8, 6]),
                              (8, 10, (0, 0, 2), [16, 16, 0, 4, 2, 8, 7])):
    want = orc.prove_fib_aux(width, log_n, aux[0], aux[1], opt, D=aux[2])[0] if aux[0] else orc.prove_fib(width, log_n, opt)[0]
    dev = ctx.trace_upload(aero_amd.fib_trace(width, log_n))
    ctx.set_kernel_timing(True)
    for _ in range(3):
        got, _ = ctx.prove_fib_aux(dev, aux[0], aux[1], aero_amd.ProofOptions(*opt), aux_degree=aux[2])
        assert got == want, (width, log_n)
    names = set(ctx.kernel_timing_report())
    ctx.set_kernel_timing(False)
    if os.environ.get("AERO_CONS_INV_TABLE") == "0" or os.environ.get("AERO_TEST_TABLE_OOM"): assert "fib_inverse_table_kernel" not in names, names
# the tables are shared by the contexts of a device and counted by every holder: a pool's slots and a second context build nothing new
if os.environ.get("AERO_CONS_INV_TABLE") != "0" and not os.environ.get("AERO_TEST_TABLE_OOM"):
    width, log_n, opt = 2, 16, [27, 8, 16, 4, 1, 8, 8]
    want = orc.prove_fib(width, log_n, opt)[0]
    other = aero_amd.Context(0)
    dev = other.trace_upload(aero_amd.fib_trace(width, log_n))
    before = other.memory_stats()[0]
    other.set_kernel_timing(True)
    first = aero_amd.Context(0)       # a third context builds the table (the main one has long evicted the shape) ...
    d1 = first.trace_upload(aero_amd.fib_trace(width, log_n))
    first.set_kernel_timing(True)
    assert first.prove_fib(d1, aero_amd.ProofOptions(*opt))[0] == want
    assert "fib_inverse_table_kernel" in set(first.kernel_timing_report())
    assert other.prove_fib(dev, aero_amd.ProofOptions(*opt))[0] == want      # ... and this one finds it
    assert "fib_inverse_table_kernel" not in set(other.kernel_timing_report())
    assert other.memory_stats()[0] - before == 5 * 8 * (2 << log_n), (other.memory_stats(), before)     # counted by the holder: 5 words per constraint-domain row
    d1.free(); first.close()
    assert other.prove_fib(dev, aero_amd.ProofOptions(*opt))[0] == want      # the builder is gone, the table is not
    dev.free(); other.close()
print("ok")
'''


@pytest.mark.parametrize("env", [{"AERO_CONS_INV_TABLE": "1"}, {"AERO_CONS_INV_TABLE": "0"}, {"AERO_TEST_TABLE_OOM": "1"}])
def test_constraint_divisor_inverse_table(tmp_path, env):
    """AERO_CONS_INV_TABLE=0: the FibAir constraint kernel inverts its two boundary divisors per thread (batched over four rows) instead of
    reading them from the per-shape table the first proof of a shape builds (stark.hip: fib_inverse_table_kernel); proof bytes against the
    oracle in both fields, with and without an auxiliary segment, more shapes than a context keeps (least recently used out). The table is
    one per device and shape for all contexts; AERO_TEST_TABLE_OOM=1 makes its allocation fail: the proofs fall back to per-row inversion."""
    run(tmp_path, CONS_TABLE, env)


DEEP_FORM = r'''
for width, log_n, aux, opt in ((2, 16, (0, 0, 2), [27, 8, 16, 4, 1, 8, 8]), (4, 12, (2, 3, 4), [20, 8, 4, 4, 1, 4, 5]), (6, 11, (2, 3, 4), [20, 8, 4, 4, 1, 4, 5]), (2, 13, (0, 0, 2), [27, 4, 8, 4, 1, 8, 6]),
                              (8, 10, (0, 0, 2), [16, 16, 0, 4, 2, 8, 7]), (72, 12, (0, 0, 2), [27, 8, 8, 4, 1, 8, 6]), (2, 3, (0, 0, 2), [4, 8, 0, 4, 1, 2, 3]),
                              (4, 7, (0, 0, 2), [8, 2, 0, 4, 1, 4, 4])):
    want = orc.prove_fib_aux(width, log_n, aux[0], aux[1], opt, D=aux[2])[0] if aux[0] else orc.prove_fib(width, log_n, opt)[0]
    dev = ctx.trace_upload(aero_amd.fib_trace(width, log_n))
    ctx.set_kernel_timing(True)
    got, _ = ctx.prove_fib_aux(dev, aux[0], aux[1], aero_amd.ProofOptions(*opt), aux_degree=aux[2])
    names = set(ctx.kernel_timing_report())
    ctx.set_kernel_timing(False)
    assert got == want, (width, log_n)
    base_field = opt[4] == 1
    narrow = width + aux[0] < 8           # from 8 columns on the evaluation form is the default too
    assert ("deep_coeff_quotient_kernel" in names) == (base_field and narrow and os.environ.get("AERO_DEEP_COEFF") != "0"), (names, opt)
    assert ("deep_kernel" in names) != ("deep_coeff_quotient_kernel" in names), names
print("ok")
'''


@pytest.mark.parametrize("flag", ["1", "0"])
def test_deep_composition_form(tmp_path, flag):
    """AERO_DEEP_COEFF=0: the DEEP composition evaluated on the trace-length coset (per-row inversions) and interpolated, instead of the
    synthetic divisions of the coefficient vectors (base field, one GPU, fewer than 8 columns; F_p^2 and wide proofs take the evaluation form either way): proof bytes
    against the oracle for narrow, wide, auxiliary-segment and tiny traces, and the launch labels say which form ran."""
    run(tmp_path, DEEP_FORM, {"AERO_DEEP_COEFF": flag})
