"""Seeded random shapes through the sharded prover (ranks sharing the one GPU of the test box, exchanges over gloo): every rank's
proof must equal the single-GPU proof of the same case. usage: python tools/fuzz_sharded.py [cases_per_world] [seed]"""
import os, pathlib, random, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_sharded import run_world

per_world = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
total = 0
for world in (2, 4, 8):
    cases = []
    while len(cases) < per_world:
        log_n = rng.randint(6, 14)
        W = 2 * rng.choice([1, 1, 2, 3, 4, 8, 36])
        B = rng.choice([8, 8, 16, 32])
        F = rng.choice([2, 4, 8, 8])
        ext = rng.choice([1, 1, 2])
        lr = rng.choice([3, 5, 6, 8])
        aux = rng.choice([None, None, [3, 2], [9, 16, 8], [2, 3, 4], [1, 1, 7]])
        D = aux[2] if aux and len(aux) > 2 else 2
        C = 2 if not aux or D <= 2 else (4 if D <= 4 else 8)
        N = B << log_n
        dom = N
        while dom > (1 << lr):
            dom //= F
        if world > B or C > B or dom < F or dom * 8 * ext > 0xFFFF or (W + (aux[0] if aux else 0)) * N > (1 << 23) or N // world < world:
            continue
        case = {"width": W, "log_n": log_n, "options": [rng.choice([8, 27, 40]), B, rng.choice([0, 8, 16]), 4, ext, F, lr], "min_peer": rng.choice([1, 2, 16, 2048])}
        if aux:
            case["aux"] = aux
        cases.append(case)
    with tempfile.TemporaryDirectory() as d:
        for (single, per_rank, comm), case in zip(run_world(world, cases, pathlib.Path(d), timeout=1200), cases):
            for r, p in enumerate(per_rank):
                assert p == single, f"world {world} rank {r}: sharded proof differs for {case}"
            total += 1
    print(f"world {world}: {len(cases)} random shapes identical on every rank", flush=True)
print(f"{total} sharded cases ok")
