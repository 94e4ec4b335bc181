"""One rank of a sharded proof (spawned by test_gpu_sharded.py; several ranks may share one GPU over gloo).

argv: out_dir cases_json. env: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT. Every rank proves every case cooperatively and
writes <out_dir>/case<i>.rank<r>.bin; rank 0 also writes the single-GPU proof of the same case as case<i>.single.bin.
Beside the bytes every rank leaves what the harness needs to say WHERE a wrong proof went wrong (profiles/r5_sharded_anomaly.md had
only rank 0's bytes to look at): case<i>.rank<r>.evidence.json = per exchange, fingerprints of every piece this rank sent and received
(aero_amd/shard.py: TorchComm evidence; for a commitment: rows / leaf digests per peer, its subtree root, the gathered top), and
case<i>.rank<r>.status.json = the call's status (0, or the library's error code and text - a case may expect one: "expect_error").
Case keys: width, log_n, options, min_peer, aux, fault (TorchComm fault, tests only), self_verify (Context.set_self_verify mode).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out_dir, cases = sys.argv[1], json.loads(sys.argv[2])
    import torch
    import torch.distributed as dist
    import aero_amd
    from aero_amd.shard import RcclComm, TorchComm

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    dev = rank % ndev
    torch.cuda.set_device(dev)
    # control plane over gloo; data plane: the native RCCL communicator when every rank has its own GPU, else (ranks sharing
    # the test box's one GPU) torch.distributed collectives (aero_amd/shard.py: TorchComm, host or device form)
    backend = "rccl-native" if ndev >= world else "gloo"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = aero_amd.Context(dev)
    for i, case in enumerate(cases):
        opts = aero_amd.ProofOptions(*case["options"])
        if ndev >= world:
            comm = RcclComm(ctx, rank, world, min_peer_digests=case.get("min_peer", 0))
        else:
            comm = TorchComm(device=dev, min_peer_digests=case.get("min_peer", 0), evidence=True, fault=case.get("fault"))
            backend = "gloo-" + comm.gloo_tensors
        ctx.set_self_verify(case.get("self_verify", "auto"))
        trace = ctx.trace_upload(aero_amd.fib_trace(case["width"], case["log_n"]))
        aux = (case.get("aux") or [0, 0]) + [2]
        status = {"code": 0, "msg": ""}
        proof, pub = b"", None
        try:
            proof, pub = ctx.prove_fib_aux(trace, aux[0], aux[1], opts, comm=comm, aux_degree=aux[2])
        except aero_amd.AeroError as e:
            status = {"code": e.code, "msg": str(e)}
            if "expect_error" not in case:
                raise
        with open(os.path.join(out_dir, f"case{i}.rank{rank}.bin"), "wb") as f:
            f.write(proof)
        with open(os.path.join(out_dir, f"case{i}.rank{rank}.status.json"), "w") as f:
            json.dump(status, f)
        with open(os.path.join(out_dir, f"case{i}.rank{rank}.evidence.json"), "w") as f:
            json.dump(getattr(comm, "evidence", []), f)
        if rank == 0:
            ctx.set_self_verify("auto")
            single, pub1 = ctx.prove_fib_aux(trace, aux[0], aux[1], opts, aux_degree=aux[2])
            assert pub is None or pub1 == pub
            with open(os.path.join(out_dir, f"case{i}.single.bin"), "wb") as f:
                f.write(single)
            with open(os.path.join(out_dir, f"case{i}.comm.json"), "w") as f:
                json.dump({"calls": comm.calls, "bytes_sent": comm.bytes_sent, "backend": backend}, f)
        trace.free()
        if ndev >= world:
            comm.close()
        dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
