"""The library picks its fast paths by shape (two-phase LDE pass, register NTT passes, compact row copies, the FRI tail launch,
quad-lane tree tops); each has a general fallback that other shapes run through. Here the switches that force the fallbacks are
flipped one at a time (environment variables read at context creation) and whole proofs are compared with the oracle's bytes, so
that a fallback that is rarely taken on the default workload cannot rot."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, sys
sys.path.insert(0, %(root)r)
import aero_amd
from tests import oracle_lib
orc = oracle_lib.load()
orc.set_threads(16)
ctx = aero_amd.Context(0)
cases = json.loads(sys.argv[1])
for W, log_n, A, R, D, opt in cases:
    dev = ctx.trace_upload(aero_amd.fib_trace(W, log_n))
    got, pub = ctx.prove_fib_aux(dev, A, R, aero_amd.ProofOptions(*opt), aux_degree=D)
    want, want_pub, _ = orc.prove_fib_aux(W, log_n, A, R, opt, D=D)
    assert pub == want_pub and got == want, ("proof differs", W, log_n, A, R, D, opt)
    host, _ = ctx.prove_fib_aux(aero_amd.fib_trace(W, log_n), A, R, aero_amd.ProofOptions(*opt), aux_degree=D) if A == 0 else (got, None)
    assert host == got
    dev.free()
ctx.selftest(4096, 7)
print("ok")
'''

CASES = [
    (2, 16, 0, 0, 2, [27, 8, 16, 4, 1, 8, 8]),        # two-phase pass, register passes, compact rows, tail: all active by default
    (2, 14, 0, 0, 2, [27, 8, 16, 4, 2, 8, 8]),        # quadratic extension
    (4, 13, 2, 3, 5, [20, 8, 8, 4, 1, 4, 6]),         # aux segment, 8 composition columns, fold 4
    (6, 12, 0, 0, 2, [16, 16, 4, 4, 1, 2, 5]),        # blowup 16 (no two-phase pass), fold 2
]


@pytest.mark.parametrize("env", [{}, {"AERO_NTT_2PHASE": "0"}, {"AERO_NTT_REG": "0"}, {"AERO_COMPACT_ROWS": "0"}, {"AERO_FRI_TAIL": "0"},
                                 {"AERO_QUAD_TOPS": "0"}, {"AERO_NTT_2PHASE": "0", "AERO_NTT_REG": "0", "AERO_COMPACT_ROWS": "0", "AERO_FRI_TAIL": "0"}])
def test_proofs_identical_with_fast_paths_switched_off(env, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    r = subprocess.run([sys.executable, str(script), json.dumps(CASES)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **env), cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (env, r.stdout[-500:], r.stderr[-1500:])
