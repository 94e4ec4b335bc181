"""Rate of the message-level hashing seam (aero_worker_hash_rows): bincode HashingWorkItem in -> HashingResult out, host parsing,
transposition and both copies included; the CPU figure beside it is hashlib's BLAKE2s on one core (what a hashing web worker does).
usage: python tools/worker_seam_rate.py [width] [rows ...]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aero_amd
from aero_amd import messages
width = int(sys.argv[1]) if len(sys.argv) > 1 else 72
sizes = [int(a) for a in sys.argv[2:]] or [1024, 16384, 262144]
ctx = aero_amd.Context(0)
rng = np.random.default_rng(1)
for rows in sizes:
    data = rng.integers(0, aero_amd.P, size=(rows, width), dtype=np.uint64)
    item = messages.encode_hashing_work_item(data, 1)
    ctx.worker_hash_rows(item)
    reps = max(3, min(50, (1 << 22) // rows))
    t0 = time.perf_counter()
    for _ in range(reps):
        res = ctx.worker_hash_rows(item)
    dt = (time.perf_counter() - t0) / reps
    k = min(rows, 2000)
    t1 = time.perf_counter()
    for r in data[:k]:
        hashlib.blake2s(b"".join(int(e).to_bytes(8, "little") + bytes(24) for e in r)).digest()
    cpu = (time.perf_counter() - t1) / k
    print(f"{rows:8d} rows x {width}: {dt * 1e3:9.3f} ms per item = {rows / dt / 1e6:7.2f} M rows/s ({len(item) / dt / 1e9:5.2f} GB/s of message); "
          f"hashlib on one core {1 / cpu / 1e6:6.3f} M rows/s")
