"""Stage times of the CPU oracle against its thread count on this host (how far the OpenMP port scales), optionally with the environment of
bench.py's CPU leg around it (torch imported, a context and a pool alive).   usage: oracle_thread_sweep.py [plain|torch|ctx|pool] [threads ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
threads = [int(v) for v in sys.argv[2:]] or [16, 32, 64, 128]
if mode in ("torch", "ctx", "pool"):
    import torch          # noqa: F401
if mode in ("ctx", "pool"):
    import aero_amd
    keep = aero_amd.Pool(0, 8) if mode == "pool" else aero_amd.Context(0)
from tests import oracle_lib

orc = oracle_lib.load()
opt = [27, 8, 16, 4, 1, 8, 8]
print(mode, "affinity:", len(os.sched_getaffinity(0)), "cpus")
for th in threads:
    orc.set_threads(th)
    orc.prove_fib(2, 16, opt)
    p, pub, t = orc.prove_fib(2, 20, opt)
    print(th, {k: round(v * 1e3) for k, v in t.items()})
