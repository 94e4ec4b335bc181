#!/bin/bash
# round 5 session k: general auxiliary recurrences on the device - parity, then their cost at 2^18 rows (device kernel against the host step)
mkdir -p gpurun_out/r5k
timeout 2400 python -m pytest tests/test_gpu_switches.py -q -m gpu -k "general" 2>&1 | tail -30 | tee gpurun_out/r5k/tests.txt
timeout 1200 python -m pytest tests/test_gpu_air.py -q -m gpu -x 2>&1 | tail -5 | tee -a gpurun_out/r5k/tests.txt
cat > /tmp/gen_time.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, aero_amd
from tests import air_examples as ex
ctx = aero_amd.Context(0)
for log_n in (14, 18, 20):
    b, trace, pub = ex.v2_air(log_n)
    air = aero_amd.Air(b.to_bytes())
    dev = ctx.trace_upload(trace)
    rands = np.arange(1, 5, dtype=np.uint64)
    for _ in range(2): ctx.aux_columns_program(air, dev, pub, rands, 1).free()
    t = time.perf_counter()
    for _ in range(3): ctx.aux_columns_program(air, dev, pub, rands, 1).free()
    print("general_host=%s v2_air 2^%d: aux columns in %.2f ms" % (os.environ.get("AERO_AIR_GENERAL_HOST", "0"), log_n, (time.perf_counter() - t) / 3 * 1e3))
PY
python3 /tmp/gen_time.py | tee -a gpurun_out/r5k/tests.txt
AERO_AIR_GENERAL_HOST=1 python3 /tmp/gen_time.py | tee -a gpurun_out/r5k/tests.txt
