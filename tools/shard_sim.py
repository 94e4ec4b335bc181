"""Per-rank compute time of a sharded proof, measured on ONE GPU with loopback exchanges (aero_amd.shard.LoopbackComm).
usage: python tools/shard_sim.py [log_n ...]   -> JSON lines (one per log_n / world)
       AERO_SIM_SHAPE=width,aux_width,aux_rands,aux_degree,fold selects another trace shape (default 2,0,0,2,8)
       AERO_SIM_HOST=1: the trace is handed over in pinned HOST memory (aero_prove_fib_sharded_host): the rank's share of the
       host-to-device copy is inside the clock (width / world columns when world divides the width, else the whole trace)
       AERO_SIM_PROGRAM=pairs,aux,rands,fold: the VM-shaped constraint PROGRAM (aero_air_synth_vm_*; 26,9,16,4 = BASELINE configs[4]'s
       shape) through aero_prove_air / aero_prove_air_sharded_host instead of the built-in AIR"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()   # torch bundles its own HIP runtime: it must initialise before libaero_stark's
import aero_amd
from aero_amd.shard import LoopbackComm


def main():
    logs = [int(a) for a in sys.argv[1:]] or [20]
    W, A, R, D, fold = [int(v) for v in os.environ.get("AERO_SIM_SHAPE", "2,0,0,2,8").split(",")]
    ctx = aero_amd.Context(0)
    opts = aero_amd.ProofOptions.with_96_bit_security()
    opts.fri_folding_factor = fold
    host = os.environ.get("AERO_SIM_HOST", "0") == "1"
    program = os.environ.get("AERO_SIM_PROGRAM")
    for log_n in logs:
        if program:
            pairs, A, R, fold = [int(v) for v in program.split(",")]
            opts.fri_folding_factor = fold
            W, D = 20 + 2 * pairs, 8
            air = aero_amd.Air(aero_amd.synth_vm_program(log_n, pairs, A, R))
            host_trace, pub = aero_amd.synth_vm_trace(log_n, pairs)
            trace = aero_amd.PinnedTrace(host_trace) if host else ctx.trace_upload(host_trace)
            del host_trace
        else:
            trace = aero_amd.PinnedTrace(aero_amd.fib_trace(W, log_n)) if host else ctx.trace_upload(aero_amd.fib_trace(W, log_n))

        def prove(comm):
            if program:
                return ctx.prove_air(air, trace, pub, opts, comm=comm if (comm.world > 1 or not host) else None)
            if host:
                return ctx.prove_fib_sharded_host(comm, trace, opts, (A, R, D))
            return ctx.prove_fib_aux(trace, A, R, opts, comm=comm, aux_degree=D)
        reps = 7 if log_n <= 20 else 3
        base = None
        for world in (1, 2, 4, 8):
            for rank in sorted({0, world - 1}):
                comm = LoopbackComm(rank, world)
                ts = []
                for i in range(reps + 1):
                    t0 = time.perf_counter()
                    prove(comm)
                    ts.append((time.perf_counter() - t0) * 1e3)
                ts = sorted(ts[1:])
                ms = ts[len(ts) // 2]
                ctx.set_stage_timing(True)
                prove(comm)
                st = ctx.last_stage_ms()
                ctx.set_stage_timing(False)
                if world == 1:
                    base = ms
                print(json.dumps({"log_n": log_n, "shape": [W, A, R, D, fold], "air": "vm_shaped_program" if program else "built_in", "host_handover": host, "world": world, "rank": rank, "ms": round(ms, 3), "speedup_vs_1": round(base / ms, 2),
                                  "exchanges": dict(comm.calls), "bytes_sent_per_proof": comm.bytes_sent // (reps + 2),
                                  "stages_ms": {k: round(v, 3) for k, v in st.items()}}), flush=True)
        trace.release() if host else trace.free()
    ctx.close()


if __name__ == "__main__":
    main()
