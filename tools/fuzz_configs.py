"""Extended seeded sweep over the option space (the generator of tests/test_gpu_random_configs.py with larger traces and more
draws): GPU proof bytes against the oracle's, oracle verification with the OOD check. usage: python tools/fuzz_configs.py [count] [seed] [max_log_n]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aero_amd
from tests import oracle_lib
from tests.test_gpu_random_configs import composition_columns

count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_log_n = int(sys.argv[3]) if len(sys.argv) > 3 else 17
rng = random.Random(seed)
orc = oracle_lib.load()
orc.set_threads(min(32, os.cpu_count() or 1))
ctx = aero_amd.Context(0)
done, t0 = 0, time.time()
while done < count:
    log_n = rng.randint(3, max_log_n)
    W = 2 * rng.choice([1, 1, 1, 2, 3, 4, 8, 17, 36])
    A = rng.choice([0, 0, 1, 2, 3, 9])
    R = rng.choice([1, 2, 5, 16])
    D = rng.choice([2, 2, 3, 4, 5, 7, 8])
    o = [rng.choice([1, 4, 16, 27, 40, 64]), rng.choice([2, 4, 8, 8, 8, 16, 32, 64, 128]), rng.choice([0, 4, 8, 12, 16]), 4,
         rng.choice([1, 1, 2]), rng.choice([2, 4, 8, 8, 16]), rng.choice([3, 4, 5, 6, 7, 8, 10])]
    # the validity rule of the test, with the size cap lifted to 2^25 LDE cells per segment
    q, B, g, _, ext, F, lr = o
    N = B << log_n
    dom = N
    while dom > (1 << lr):
        dom //= F
    if composition_columns(A, D) > B or N > (1 << 22) or (W + A) * N > (1 << 25) or dom < F or dom * 8 * (2 if ext == 2 else 1) > 0xFFFF or q > N // 4:
        continue
    dev = ctx.trace_upload(aero_amd.fib_trace(W, log_n))
    got, pub = ctx.prove_fib_aux(dev, A, R, aero_amd.ProofOptions(*o), aux_degree=D)
    want, want_pub, _ = orc.prove_fib_aux(W, log_n, A, R, o, D=D)
    assert pub == want_pub and got == want, f"MISMATCH for log_n={log_n} W={W} A={A} R={R} D={D} options={o}"
    orc.verify_fib_aux(got, pub, W, log_n, A, R, D=D)
    dev.free()
    done += 1
print(f"{done} random configurations up to 2^{max_log_n} rows: proof bytes identical to the oracle's, verified ({time.time() - t0:.0f} s)")
