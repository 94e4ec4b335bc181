"""Which rows of aero_hash_rows differ under the guard-page allocator (diagnosis of the round-4 finding)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aero_amd
P = aero_amd.P
ctx = aero_amd.Context(0)
for width in (8, 9, 10, 11, 72, 81):
    for n in (300, 256, 512, 1000):
        rng = np.random.default_rng(width * 1000 + n)
        rows = rng.integers(0, P, size=(n, width), dtype=np.uint64)
        got = ctx.hash_rows(rows)
        bad = []
        for r in range(n):
            blob = b"".join(int(v).to_bytes(8, "little") + bytes(24) for v in rows[r])
            if got[r].tobytes() != hashlib.blake2s(blob).digest():
                bad.append(r)
        print(f"width {width} rows {n}: {len(bad)} wrong", bad[:12], ("... last " + str(bad[-1])) if bad else "", flush=True)
ctx.close()
