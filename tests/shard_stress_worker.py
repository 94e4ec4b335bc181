"""One rank of the exchange stress loop (profiles/r6_sharded_anomaly.md): the TRACE COMMITMENT of a sharded proof - interpolate, extend this
rank's coset, pack, all-to-all, hash on arrival, subtree, root all-gather (aero_commit_trace_sharded) - over and over against the known
root, ranks = processes sharing one GPU, exchanges over gloo in its two forms (aero_amd/shard.py: TorchComm "device" = the device tensors
handed to gloo, what rounds 1-5 ran and what met one wrong proof; "host" = host copies made here). The forms alternate in blocks inside
the same processes, so both see the same machine state. The reference's counterpart is its A/B pair `prove` vs `prove_sequential`
(aero-sdk/miden-wasm/src/proving_worker.rs:124-223 / :441-518): same statement through two data planes, results compared.

argv: out_dir log_n iters block forms(comma separated). env: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT.
Writes <out_dir>/stress.rank<r>.json: per form {iters, wrong, secs}, and for every wrong iteration the root, the subtree roots, which
fingerprints differ from the form's first good iteration (sent pieces = this rank's LDE rows; received pieces; subtree root; gathered top).
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out_dir, log_n, iters, block = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    forms = sys.argv[5].split(",")
    width = int(sys.argv[6]) if len(sys.argv) > 6 else 2
    import torch
    import torch.distributed as dist
    import aero_amd
    from aero_amd.shard import TorchComm

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = aero_amd.Context(dev)
    opts = aero_amd.ProofOptions(27, 8, 16, 4, 1, 8, 8)
    trace = ctx.trace_upload(aero_amd.fib_trace(width, log_n))
    box = [None]
    if rank == 0:
        box[0] = ctx.commit_trace(trace, opts)[0]          # the single-GPU tree's root
    dist.broadcast_object_list(box, src=0)
    want = box[0]
    comms = {f: TorchComm(device=dev, gloo_tensors=f, evidence=True) for f in forms}
    stats = {f: {"iters": 0, "wrong": 0, "secs": 0.0} for f in forms}
    ref = {}
    failures = []
    for it in range(iters):
        form = forms[(it // block) % len(forms)]
        comm = comms[form]
        comm.evidence = []
        t0 = time.time()
        root, subs = ctx.commit_trace(trace, opts, comm)
        stats[form]["secs"] += time.time() - t0
        stats[form]["iters"] += 1
        if root == want:
            ref.setdefault(form, comm.evidence)
            continue
        stats[form]["wrong"] += 1
        rec = {"iteration": it, "form": form, "rank": rank, "root": root.hex(), "subtree_roots": [s.hex() for s in subs], "evidence": comm.evidence}
        good = ref.get(form) or next(iter(ref.values()), None)
        if good is not None and len(good) == len(comm.evidence):
            diff = []
            for k, (a, b) in enumerate(zip(good, comm.evidence)):
                for side in ("send", "recv"):
                    for q, (x, y) in enumerate(zip(a[side], b[side])):
                        if x != y:
                            diff.append(f"call {k} {b['op']} {side}[{q}]")
            rec["differs_from_good_iteration"] = diff
        failures.append(rec)
    trace.free()
    dist.barrier()
    with open(os.path.join(out_dir, f"stress.rank{rank}.json"), "w") as f:
        json.dump({"rank": rank, "world": world, "log_n": log_n, "width": width, "want": want.hex(), "stats": stats, "failures": failures}, f)
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
