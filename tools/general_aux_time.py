"""Wall-clock of the auxiliary segment of the two programs with GENERAL (non-scannable) recurrences (tests/air_examples.py: v2_air - one general
column next to four scanned ones; general_chain_air - three general columns that read each other) through the stage entry point
aero_aux_columns_program: the device scans, the copies down, the host chains, the copies up. usage: general_aux_time.py [log_n ...]
(AERO_AIR_GENERAL_DEVICE=1 times the device form instead; profiles/r5_general_recurrence.md has round 5's 22 ms at 2^20 for v2.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aero_amd
from tests import air_examples as ex

ctx = aero_amd.Context(0)
for log_n in [int(v) for v in sys.argv[1:]] or [14, 18, 20]:
    for name, (b, trace, pub), nrand in (("v2_air", ex.v2_air(log_n), 4), ("general_chain_air", ex.general_chain_air(log_n), 2)):
        air = aero_amd.Air(b.to_bytes())
        dev = ctx.trace_upload(trace)
        rands = np.arange(11, 11 + nrand, dtype=np.uint64)
        ts = []
        for it in range(7):
            t0 = time.perf_counter()
            m = ctx.aux_columns_program(air, dev, pub, rands, 1)
            ts.append((time.perf_counter() - t0) * 1e3)
            m.free()
        ts = sorted(ts[2:])
        print(f"{name} 2^{log_n}: auxiliary segment {ts[len(ts) // 2]:.2f} ms (min {ts[0]:.2f}) = {1e6 * ts[len(ts) // 2] / (1 << log_n):.1f} ns per row")
        dev.free()
