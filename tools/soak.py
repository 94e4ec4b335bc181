"""Soak: many proofs of several shapes on one context; checks determinism and that pool memory does not grow."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aero_amd
ctx = aero_amd.Context(0)
shapes = [(2, 16, (0, 0, 2), {}), (4, 12, (3, 2, 8), {"field_extension": 2}), (2, 20, (0, 0, 2), {}), (8, 14, (2, 2, 3), {"fri_folding_factor": 4})]
ref = {}
t0 = time.time()
peak0 = None
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    for (w, ln, aux, kw) in shapes:
        o = aero_amd.ProofOptions.with_96_bit_security()
        for k, v in kw.items(): setattr(o, k, v)
        dev = ctx.trace_upload(aero_amd.fib_trace(w, ln)) if it < 2 or (w, ln) not in ref else None
        if (w, ln) not in ref:
            ref[(w, ln)] = {"dev": dev}
        d = ref[(w, ln)]["dev"]
        p, _ = ctx.prove_fib_aux(d, aux[0], aux[1], o, aux_degree=aux[2])
        h = hashlib.sha256(p).hexdigest()
        assert ref[(w, ln)].setdefault("h", h) == h, "non-deterministic proof"
        if dev is not None and dev is not d: dev.free()
    if it == 5: peak0 = ctx.memory_stats()
print("iterations done in", round(time.time() - t0, 1), "s; memory (in_use, peak) after 5:", peak0, "at end:", ctx.memory_stats())
assert ctx.memory_stats()[1] == peak0[1], "pool peak grew during the soak"
print("soak ok")
