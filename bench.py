#!/usr/bin/env python3
"""bench.py — throughput of the MI355X proving hot path on BASELINE.json's headline workload.

    python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input: a batch of independent traces (64 per GPU for the
2^20 x 2 workload: `traces_per_step_per_gpu`), each proven completely, --concurrent (default 8) of them in flight at a time,
each on its own HIP stream so that the latency-bound tree tops and host round trips of one proof overlap the throughput-bound
kernels of another. Every trace is the 2^20-row x 2-column Fibonacci trace (BASELINE configs[1]: Goldilocks base field,
blowup 8, blake2s, 27 queries, grinding 16, FRI fold 8) — interpolate, LDE, row hashing + Merkle, constraint evaluation,
composition commit, OOD, DEEP, FRI, grinding, openings, proof bytes.

Hand-over: the trace is in (pinned) HOST memory and the proof bytes come back to host
memory; the host-to-device copy of every trace is INSIDE the timed region (`config.h2d_included: true`), enqueued on the
proving stream so that it overlaps the other streams' kernels. The same measurement with the traces already resident in HBM
is reported next to it as `hbm_resident_value` (`--resident` swaps the two).
metric = trace cells/sec = n * W * proofs / wall-clock (max over ranks); `single_proof_ms` is the wall-clock of ONE
proof with nothing else in flight (the "proof-gen wall-clock" half of BASELINE's metric), host trace -> proof bytes.

Multi-GPU (N > 1): one process per GPU, each proving its own independent traces (the path shards by independent
proofs; no data-path collective) -> "scaling": "weak". torch.distributed (RCCL) is used only for the barriers and the
max-over-ranks of the timing.

ONE proof sharded over the N GPUs (BASELINE configs[3]: LDE cosets + Merkle subtrees per GPU, RCCL all-to-all of leaf
digests and all-gather of subtree roots per commitment) is the other multi-GPU mode:
  * `--mode sharded` makes it the timed step ("scaling": "strong": the total work is one trace whatever N is);
  * in the default mode with N > 1, rank 0 additionally measures it AFTER the timed region in a separate group of N
    worker processes under a hard timeout (so that a problem there can never take the headline line down) and reports
    it as "sharded_proof" on the same JSON line; its proof bytes are asserted identical to the single-GPU proof.

Extra objects on the JSON line:
  roofline     — for the dominant kernel (largest share of HIP-event time): achieved = algorithmic bytes of its
                 launches / their summed duration, measured with HIP events on the launch stream inside the timed
                 region (only that kernel's launches are bracketed, so the timed region is perturbed by 2 event
                 records per launch of one kernel). traffic = PMC-measured HBM bytes per launch from
                 profiles/pmc_traffic.json when present, else null. `limiter` names what actually bounds the kernel.
  cpu_baseline — the CPU oracle (oracle/, kind "port") timed on this box's host cores on a bounded sample (1 warm-up +
                 median of 5 runs), rank 0, N == 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)

WORKLOADS = {
    # name: (log_n, width, option overrides)
    "fib_2^20x2_blowup8_blake2s_base": (20, 2, {}),
    "fib_2^20x2_blowup8_blake2s_quadratic": (20, 2, {"field_extension": 2}),
    "fib_2^24x2_blowup8_blake2s_base": (24, 2, {}),
    "fib_2^20x72_blowup8_blake2s_base": (20, 72, {}),
    "fib_2^16x2_blowup8_blake2s_base": (16, 2, {}),
    # BASELINE configs[4] stand-in (the Miden AIR is absent from the reference mount): Miden's SHAPE — 72 main columns, one
    # auxiliary segment of 9 columns built from 16 coin elements, constraints of degree 8 (=> constraint-evaluation blowup 8
    # and 8 composition columns, as in the golden proof fib.bin), 2^22 rows, FRI folding factor 4 — on the synthetic AIR
    # (Fibonacci pairs + prefix-product aux columns). Labelled a stand-in wherever it is reported. aux = (columns, random
    # elements, constraint degree).
    "standin_miden_shape_2^22x(72+9aux)_deg8_fold4": (22, 72, {"fri_folding_factor": 4, "aux": (9, 16, 8)}),
    "standin_miden_shape_2^18x(72+9aux)_deg8_fold4": (18, 72, {"fri_folding_factor": 4, "aux": (9, 16, 8)}),
    "standin_miden_shape_2^22x(72+9aux)_fold4": (22, 72, {"fri_folding_factor": 4, "aux": (9, 16, 2)}),
    # the same shape through the AIR-as-data path (include/aero_air.h): the VM-shaped constraint PROGRAM of aero_air_synth_vm_* -
    # 76 + 9 transition constraints up to degree 8, periodic columns, interior / periodic assertions, two exemptions - proven by
    # aero_pool_prove_air*. program = (Fibonacci pairs, aux columns, random elements); aux = (columns, -, -) only counts the cells.
    "program_vm_shape_2^22x(72+9aux)_fold4": (22, 72, {"fri_folding_factor": 4, "aux": (9, 16, 8), "program": (26, 9, 16)}),
    "program_vm_shape_2^20x(72+9aux)_fold4": (20, 72, {"fri_folding_factor": 4, "aux": (9, 16, 8), "program": (26, 9, 16)}),
    "program_vm_shape_2^14x(24+3aux)_fold4": (14, 24, {"fri_folding_factor": 4, "aux": (3, 4, 8), "program": (2, 3, 4)}),      # small: tests
    "standin_miden_shape_2^18x(72+9aux)_fold4": (18, 72, {"fri_folding_factor": 4, "aux": (9, 16, 2)}),
}


def rank_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def timed_steps(step, steps, barrier):
    """Exactly `steps` calls of step() bracketed by barrier() on both sides; returns this rank's wall-clock seconds."""
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0


def max_over_ranks(dt, dist, device):
    """MAX-reduce of the per-rank timing (the slowest rank defines the job's wall-clock)."""
    if dist is None:
        return dt
    import torch
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_value(cells_per_proof, steps, world, dt):
    """Whole-job throughput: every rank proves `steps` independent traces (weak scaling)."""
    return cells_per_proof * steps * world / dt


def cpu_thread_candidates(cores):
    """Thread counts tried for the CPU baseline: the oracle's OpenMP loops stop scaling well before 256 threads."""
    c = sorted({max(1, cores // 16), max(1, cores // 8), max(1, cores // 4), max(1, cores // 2), cores})
    return [t for t in c if t >= 1]


def make_options(aero_amd, over):
    opt = aero_amd.ProofOptions.with_96_bit_security()
    for k, v in over.items():
        if k not in ("aux", "program"):
            setattr(opt, k, v)
    return opt


def trace_cols(width, over):
    """Trace columns that count as cells: main columns plus auxiliary-segment columns."""
    return width + (over["aux"][0] if over.get("aux") else 0)


def prove_call(ctx, dev, opt, over, comm=None):
    """One proof of the workload: plain FibAir, FibAir + auxiliary segment when the workload names one, or a constraint program.
    Returns (proof bytes, public inputs)."""
    if over.get("_air") is not None:
        return ctx.prove_air(over["_air"], dev, over["_pub"], opt, comm=comm), over["_pub"]
    aux = over.get("aux")
    if aux or comm is not None:
        return ctx.prove_fib_aux(dev, aux[0] if aux else 0, aux[1] if aux else 0, opt, comm=comm, aux_degree=aux[2] if aux else 2)
    return ctx.prove_fib(dev, opt)


def kernel_times(ctx, fn, names, reps):
    """Per-kernel HIP-event ms per proof over `reps` calls of fn (one untimed call first) + the wall-clock per call."""
    fn()
    ctx.set_kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    wall = (time.perf_counter() - t0) / reps * 1e3
    rep = ctx.kernel_timing_report()
    ctx.set_kernel_timing(False)
    out = {k: round(rep[k][1] / reps, 4) for k in names if k in rep}
    out["proof_wall_ms"] = round(wall, 3)
    return out


def air_program_leg(aero_amd, ctx, device=0, log_n=20, reps=3):
    """The AIR-as-data path (include/aero_air.h) beside the hard-wired kernels: (1) FibAir(72) as a program vs fib_constraints_kernel
    on the same trace, same proof bytes; (2) a VM-shaped program (72 + 9 columns, 76 + 9 transition constraints, degree <= 8) proven
    and verified. Resident traces, one proof at a time."""
    opt = aero_amd.ProofOptions(27, 8, 16, 4, 1, 8, 8)
    out = {}
    dev = ctx.trace_upload(aero_amd.fib_trace(72, log_n))
    air = aero_amd.Air(aero_amd.fib_program(72))
    want, pub = ctx.prove_fib(dev, opt)
    assert ctx.prove_air(air, dev, pub, opt) == want, "program proof differs from the hard-wired proof"
    hard = kernel_times(ctx, lambda: ctx.prove_fib(dev, opt), ["fib_constraints_kernel"], reps)
    prog = kernel_times(ctx, lambda: ctx.prove_air(air, dev, pub, opt), ["air_jit_kernel", "air_constraints_kernel"], reps)
    pk = "air_jit_kernel" if "air_jit_kernel" in prog else "air_constraints_kernel"
    dev.free()
    out["fibair_72_as_program"] = {"workload": f"fib_2^{log_n}x72", "hard_wired_ms": hard, "program_ms": prog,
                                   "evaluator": "compiled at run time (hiprtc)" if pk == "air_jit_kernel" else "interpreter",
                                   "constraint_kernel_ratio": round(prog[pk] / hard["fib_constraints_kernel"], 3)}
    pairs, A, R = 26, 9, 16
    fold4 = aero_amd.ProofOptions(27, 8, 16, 4, 1, 4, 8)
    program = aero_amd.synth_vm_program(log_n, pairs, A, R)
    trace, vpub = aero_amd.synth_vm_trace(log_n, pairs)
    vdev = ctx.trace_upload(trace)
    # cold start: the first proof of a program pays the hiprtc compilation of its evaluation kernel unless the host prepared it
    # (aero_air_prepare) or a code-object cache directory holds it (AERO_AIR_JIT_CACHE); each figure from a fresh program handle
    import shutil
    import tempfile
    cache_dir = tempfile.mkdtemp(prefix="aero_jit_cache_")
    os.chmod(cache_dir, 0o700)
    had = os.environ.get("AERO_AIR_JIT_CACHE")
    cold = {}
    try:
        def first_proof(handle):
            t0 = time.perf_counter()
            p_ = ctx.prove_air(handle, vdev, vpub, fold4)
            return p_, round((time.perf_counter() - t0) * 1e3, 2)
        os.environ["AERO_AIR_JIT_CACHE"] = cache_dir
        vair = aero_amd.Air(program)
        proof, cold["first_proof_ms"] = first_proof(vair)                                # compiles, writes the cache entry
        p2, cold["first_proof_ms_cached"] = first_proof(aero_amd.Air(program))            # fresh handle: code object read from the directory
        os.environ.pop("AERO_AIR_JIT_CACHE")
        prepared = aero_amd.Air(program)
        t0 = time.perf_counter()
        prepared.prepare(log_n, fold4)
        cold["prepare_ms"] = round((time.perf_counter() - t0) * 1e3, 2)                   # the compilation, off the proof's clock
        p3, cold["first_proof_ms_prepared"] = first_proof(prepared)
        assert p2 == proof and p3 == proof, "cold-start proofs differ"
    finally:
        if had is None:
            os.environ.pop("AERO_AIR_JIT_CACHE", None)
        else:
            os.environ["AERO_AIR_JIT_CACHE"] = had
        shutil.rmtree(cache_dir, ignore_errors=True)
    aero_amd.verify_air(proof, vpub, vair, expected_log_n=log_n)
    vm = kernel_times(ctx, lambda: ctx.prove_air(vair, vdev, vpub, fold4), ["air_jit_kernel", "air_constraints_kernel", "air_aux_factors_kernel"], reps)
    vk = "air_jit_kernel" if "air_jit_kernel" in vm else "air_constraints_kernel"
    vdev.free()
    # the same program with 3 proofs in flight: one C call, a context + stream + worker thread per slot inside the library
    vpool = aero_amd.Pool(device, 3)
    vdevs = [vpool.ctx(i).trace_upload(trace) for i in range(3)]
    rounds = 3
    proofs3 = vpool.prove_air(vair, vdevs, vpub, fold4, rounds=1)          # warm: modules loaded on every context
    assert all(p_ == proof for p_ in proofs3), "pool proofs differ from the single proof"
    t0 = time.perf_counter()
    vpool.prove_air(vair, vdevs, vpub, fold4, rounds=rounds)
    dt3 = time.perf_counter() - t0
    for d_ in vdevs:
        d_.free()
    vpool.close()
    info = vair.info()
    out["vm_shaped_program"] = {"workload": f"synth_vm_2^{log_n}x(72+9aux)_fold4", "ms": vm, "verified": True, "proof_bytes": len(proof),
                                "transition_constraints": info["main_transition"] + info["aux_transition"],
                                "cells_per_s_single_proof": round((81 << log_n) / (vm["proof_wall_ms"] * 1e-3)),
                                "cells_per_s_3_in_flight": round(3 * rounds * (81 << log_n) / dt3),
                                "constraint_stage_share": round(vm[vk] / vm["proof_wall_ms"], 3), "cold_start": cold}
    return out


def sharded_measure(workload, steps, warmup, rank, world, device, dist, torch, comm_kind="auto"):
    """ONE proof of `workload` proven cooperatively by all ranks of the default process group: `warmup` untimed proofs, then
    exactly `steps` timed ones (barrier + synchronize on both sides, max over ranks). Every rank checks its proof bytes
    against the single-GPU proof it computes itself. Returns the result dict (same on every rank)."""
    import aero_amd
    from aero_amd.shard import RcclComm, TorchComm
    log_n, width, over = WORKLOADS[workload]
    opt = make_options(aero_amd, over)
    ctx = aero_amd.Context(device)
    # the metric's hand-over: the trace lies in (pinned) HOST memory; a rank copies its width / world columns inside the clock
    # (aero_prove_fib_sharded_host). One process per GPU: every rank holds the synthetic trace in its own host memory.
    host = aero_amd.PinnedTrace(aero_amd.fib_trace(width, log_n))
    aux = over.get("aux") or (0, 0, 2)
    dev = ctx.trace_upload(host.array)
    # data plane: the library's native RCCL communicator when every rank has its own GPU; torch.distributed collectives on the
    # device buffers (gloo) when the ranks share one GPU (RCCL refuses two ranks on one device)
    native = comm_kind == "rccl" or (comm_kind == "auto" and torch.cuda.device_count() >= world)
    # the rank's host side next to its GPU (numa.hip): bound before the pinned trace below would ideally be allocated; reported per rank
    import ctypes as C
    node, ncpu = C.c_int32(-1), C.c_uint32(0)
    aero_amd.lib().aero_numa_device_node(C.c_int32(device), C.byref(node))
    aero_amd.lib().aero_numa_bind_thread(C.c_int32(device), C.byref(ncpu))
    comm = RcclComm(ctx, rank, world) if native else TorchComm(device=device)
    me = {"rank": rank, "pid": os.getpid(), "device": device, "gpu_numa_node": node.value, "cpus_bound_to": ncpu.value,
          "rccl": comm.info() if native else None}
    placement = [None] * world
    dist.all_gather_object(placement, me)

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    def sharded_proof():
        return ctx.prove_fib_sharded_host(comm, host, opt, aux)

    proof = None
    for _ in range(max(1, warmup)):
        proof, _ = sharded_proof()
    # the single-GPU proof every rank compares with. When the ranks SHARE one GPU (--share-gpu) only rank 0 computes it and the others compare
    # against its SHA-256: eight whole config-5 proofs (46 GB each, and every context keeps its blocks cached) do not fit one device
    import hashlib
    sharing = torch.cuda.device_count() < world
    if sharing:
        box = [None]
        if rank == 0:
            single, _ = prove_call(ctx, dev, opt, over)
            box[0] = hashlib.sha256(single).hexdigest()
        dist.broadcast_object_list(box, src=0)
        identical = hashlib.sha256(proof).hexdigest() == box[0]
    else:
        single, _ = prove_call(ctx, dev, opt, over)
        identical = proof == single
    calls0, sent0 = dict(comm.calls), comm.bytes_sent
    dt = timed_steps(sharded_proof, steps, barrier)
    ctl = torch.device("cuda", device) if dist.get_backend() == "nccl" else "cpu"     # control-plane tensors
    dt = max_over_ranks(dt, dist, ctl)
    calls1, sent1 = dict(comm.calls), comm.bytes_sent
    flag = torch.tensor([1 if identical else 0], dtype=torch.int32, device=ctl)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    ctx.set_stage_timing(True)
    sharded_proof()
    stages = ctx.last_stage_ms()
    ctx.set_stage_timing(False)
    single_ms = 0.0
    if not sharing or rank == 0:
        t1 = time.perf_counter()
        for _ in range(3):
            ctx.prove_fib_aux(host, aux[0], aux[1], opt, aux_degree=aux[2])      # one GPU, same hand-over (whole trace copied in)
        single_ms = (time.perf_counter() - t1) * 1e3 / 3
    if sharing:
        box = [single_ms]
        dist.broadcast_object_list(box, src=0)
        single_ms = box[0]
    res = {
        "workload": workload, "world": world, "exchange": ("native RCCL (aero_rccl_*), stream-ordered" if native else f"torch.distributed {dist.get_backend()} on device buffers"),
        "control_plane": dist.get_backend(), "gpus_visible": torch.cuda.device_count(), "ranks": placement,
        "self_verify": "on (default for proofs made by more than one rank: every rank's bytes pass the library's verifier before they are returned)",
        "h2d_included": True, "hand_over": "trace in pinned host memory; every rank copies its share of the columns inside the timed region",
        "steps": steps, "ms_per_proof": 1e3 * dt / steps, "value": (1 << log_n) * trace_cols(width, over) * steps / dt, "unit": "cells/s",
        "single_gpu_ms_same_process": single_ms, "speedup_vs_single_gpu": single_ms / (1e3 * dt / steps),
        "proof_identical_to_single_gpu_on_every_rank": bool(flag.item() == 1), "proof_bytes": len(proof),
        "exchanges_per_proof": {k: (calls1[k] - calls0[k]) // steps for k in calls1},
        "bytes_sent_per_rank_per_proof": (sent1 - sent0) // steps,
        "rank0_stage_ms": {k: round(v, 3) for k, v in stages.items()},
    }
    dev.free()
    host.release()
    if native:
        comm.close()
    ctx.close()
    return res


def shard_worker_main(args):
    """Hidden mode: one rank of the side measurement spawned by rank 0 of a default-mode run (see spawn_sharded_check)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = rank_env()
    ndev = torch.cuda.device_count()
    device = local_rank % max(ndev, 1)
    torch.cuda.set_device(device)
    # control plane (barriers, the max-reduce of the timing, the 128-byte RCCL id): gloo. The data plane is chosen in sharded_measure.
    dist.init_process_group("gloo", rank=rank, world_size=world)
    results = []
    for wl in args.sharded_workloads.split(","):
        try:
            results.append(sharded_measure(wl, args.steps, args.warmup, rank, world, device, dist, torch, args.shard_comm))
        except Exception as e:  # keep the other workloads; every rank fails the same way on a deterministic error
            results.append({"workload": wl, "error": f"{type(e).__name__}: {e}"[:400]})
            break
    if rank == 0:
        print("SHARDED_RESULT " + json.dumps(results), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def spawn_sharded_check(world, workloads, steps, warmup, timeout_s, comm_kind="auto"):
    """Run the sharded-proof measurement in `world` fresh worker processes (one per GPU) under a hard timeout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = {k: v for k, v in os.environ.items() if not k.startswith(("TORCHELASTIC", "GROUP_", "ROLE_", "LOCAL_WORLD", "TORCH_NCCL_ASYNC"))}
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    cmd = [sys.executable, os.path.abspath(__file__), "--shard-worker", "--gpus", str(world), "--steps", str(steps), "--warmup", str(warmup),
           "--sharded-workloads", workloads, "--shard-comm", comm_kind]
    import tempfile
    logdir = tempfile.mkdtemp(prefix="aero_shard_")
    logs = [open(os.path.join(logdir, f"rank{r}.log"), "wb") for r in range(world)]
    procs = [subprocess.Popen(cmd, env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=logs[r], stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL)
             for r in range(world)]
    t_end = time.time() + timeout_s
    timed_out = False
    for p in procs:
        try:
            p.wait(timeout=max(1.0, t_end - time.time()))
        except subprocess.TimeoutExpired:
            timed_out = True
            break
    for p in procs:
        if p.poll() is None:
            p.kill()
            p.wait()
    for f in logs:
        f.close()

    def tail(r, nbytes=600):
        try:
            with open(os.path.join(logdir, f"rank{r}.log"), "rb") as f:
                return f.read()[-nbytes:].decode(errors="replace")
        except OSError:
            return ""

    text = ""
    try:
        with open(os.path.join(logdir, "rank0.log"), "rb") as f:
            text = f.read().decode(errors="replace")
    except OSError:
        pass
    for line in text.splitlines():
        if line.startswith("SHARDED_RESULT "):
            return {"results": json.loads(line[len("SHARDED_RESULT "):])}
    rc = [p.returncode for p in procs]
    why = f"timeout after {timeout_s} s" if timed_out else f"no result (exit codes {rc})"
    return {"error": why, "rank0_log_tail": tail(0), "last_rank_log_tail": tail(world - 1)}


def launch_ranks(n, argv, script=None, timeout_s=None):
    """`python bench.py --gpus N` started WITHOUT a launcher (no RANK in the environment): start the N rank processes here —
    fresh children (never a re-exec: this process has not touched the GPU and does not after this point), one per GPU, with
    the same RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* contract torch.distributed.run provides. Rank 0 inherits stdout (its
    JSON line is the job's line), the other ranks log to files. Returns the exit code (non-zero if any rank failed)."""
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    logdir = tempfile.mkdtemp(prefix="aero_bench_ranks_")
    cmd = [sys.executable, script or os.path.abspath(__file__)] + list(argv)
    procs, logs = [], []
    for r in range(n):
        out = None if r == 0 else open(os.path.join(logdir, f"rank{r}.log"), "wb")
        logs.append(out)
        procs.append(subprocess.Popen(cmd, env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=out,
                                      stderr=None if r == 0 else subprocess.STDOUT, stdin=subprocess.DEVNULL))
    t_end = None if timeout_s is None else time.time() + timeout_s
    rc = 0
    pending = list(range(n))
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is not None:
                pending.remove(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    # one rank died: the others would wait for it in a collective for ever
                    for q in pending:
                        procs[q].terminate()
        if pending:
            if t_end is not None and time.time() > t_end:
                for q in pending:
                    procs[q].kill()
                rc = rc or 124
            time.sleep(0.05)
    for f in logs:
        if f is not None:
            f.close()
    if rc != 0:
        for r in range(1, n):
            try:
                with open(os.path.join(logdir, f"rank{r}.log"), "rb") as f:
                    tail = f.read()[-1500:].decode(errors="replace")
                if tail.strip():
                    print(f"---- rank {r} log tail ----\n{tail}", file=sys.stderr)
            except OSError:
                pass
    return rc


def median(xs):
    xs = sorted(xs)
    m = len(xs) // 2
    return xs[m] if len(xs) % 2 else 0.5 * (xs[m - 1] + xs[m])


CPU_CHILD = r'''
import hashlib, json, os, sys, time
sys.path.insert(0, %(root)r)
from tests import oracle_lib
a = json.loads(sys.argv[1])
orc = oracle_lib.load()
width, log_n, aux, opt, probe_log, cands = a["width"], a["log_n"], a["aux"], a["opt"], a["probe_log"], a["candidates"]
def prove(ln):
    return orc.prove_fib_aux(width, ln, aux[0], aux[1], opt, D=aux[2])
best = None
for t in cands:
    orc.set_threads(t)
    prove(min(12, log_n))
    probe = prove(probe_log)[2]
    if best is None or probe["total"] < best[1]:
        best = (t, probe["total"])
cores = best[0]
orc.set_threads(cores)
s_log_n = a["sample_log_n"]
if not s_log_n:
    projected, s_log_n = best[1] * (1 << max(log_n - probe_log, 0)) * 1.3, log_n
    while s_log_n > probe_log and projected > 5.0:
        s_log_n -= 1
        projected /= 2
runs, walls, proof = [], [], None
for i in range(6):                                # run 0 = warm-up
    t1 = time.perf_counter()
    proof, _, times = prove(s_log_n)
    if i:
        runs.append(times["total"]); walls.append(time.perf_counter() - t1)
orc.set_threads(1)
st = prove(a["st_log"])[2]
print(json.dumps({"cores": cores, "s_log_n": s_log_n, "runs": runs, "walls": walls, "sha256": hashlib.sha256(proof).hexdigest(), "single_total": st["total"]}))
'''


def cpu_baseline_leg(args, log_n, width, over, opt, first_proof):
    """The CPU oracle (oracle/, kind "port") on this box's host cores: thread count picked on a probe, then 1 warm-up + the
    MEDIAN of 5 complete proofs of the sample (SURVEY 8d / BASELINE.md section 2). The built-in AIR's oracle runs in a process of its own
    (no torch, no HIP runtime, no pool threads next to its OpenMP team: inside this process the same runs took 1.57 s instead of 1.15 s);
    the proof's SHA-256 comes back and is compared with the GPU's."""
    if not over.get("program"):
        import hashlib
        import subprocess
        ncpu = os.cpu_count() or 1
        aux = over.get("aux") or (0, 0, 2)
        cols = trace_cols(width, over)
        probe_log = min(14 if width > 8 else 18, log_n)
        st_log = min(log_n, 16 if width <= 8 else 13)
        job = {"width": width, "log_n": log_n, "aux": list(aux), "opt": opt.to_list(), "probe_log": probe_log, "candidates": cpu_thread_candidates(ncpu),
               "sample_log_n": min(args.cpu_sample_log_n, log_n) if args.cpu_sample_log_n else 0, "st_log": st_log}
        env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS",)}
        # threads pinned to neighbouring cores: 16 % faster than an unbound team at the best thread count on this host (2^20 x 2: 1.09 s against
        # 1.29 s at 32 threads; more threads are slower bound or not: profiles/r6_oracle_thread_binding.txt)
        env.setdefault("OMP_PROC_BIND", "close")
        env.setdefault("OMP_PLACES", "cores")
        r = subprocess.run([sys.executable, "-c", CPU_CHILD % {"root": ROOT}, json.dumps(job)], capture_output=True, text=True, timeout=1800, env=env, cwd=ROOT)
        if r.returncode != 0:
            raise RuntimeError("cpu baseline child failed: " + r.stderr[-400:])
        c = json.loads(r.stdout.strip().splitlines()[-1])
        if c["s_log_n"] == log_n:
            assert c["sha256"] == hashlib.sha256(first_proof).hexdigest(), "GPU and CPU proofs differ"
        med = median(c["runs"])
        return {
            "value": (1 << c["s_log_n"]) * cols / med, "unit": "cells/s", "cores": c["cores"], "kind": "port",
            "sample": f"complete proofs of a 2^{c['s_log_n']} x {cols} trace of the same AIR with the same options: 1 warm-up + median of 5 runs "
                      f"(prover time per run {', '.join(f'{x:.2f}' for x in c['runs'])} s; median {med:.2f} s; wall incl. trace generation {median(c['walls']):.2f} s); "
                      f"OpenMP port in a process of its own, single-run thread sweep over {cpu_thread_candidates(ncpu)} on a 2^{probe_log} probe picked {c['cores']}; "
                      f"host has {ncpu} logical CPUs" + ("; proof bytes identical to the GPU's (SHA-256)" if c["s_log_n"] == log_n else ""),
            "runs_s": c["runs"],
            "single_thread": {"value": (1 << st_log) * cols / c["single_total"], "unit": "cells/s",
                              "sample": f"one complete proof of a 2^{st_log} x {cols} trace, 1 thread (the reference binary runs Winterfell single-threaded)"},
        }
    from tests import oracle_lib
    orc = oracle_lib.load()
    ncpu = os.cpu_count() or 1
    aux = over.get("aux") or (0, 0, 2)
    cols = trace_cols(width, over)
    if over.get("program"):
        import aero_amd
        pairs, pa, pr = over["program"]

        real = orc

        class _ProgramOracle:                         # the oracle's ProgramAir prover behind the call shape used below
            def __init__(self):
                self.cache = {}

            def prove_fib_aux(self, width_, ln, a_, r_, o_, D=2):
                if ln not in self.cache:
                    self.cache = {ln: (aero_amd.synth_vm_program(ln, pairs, pa, pr),) + aero_amd.synth_vm_trace(ln, pairs)}
                prog_, tr_, pub_ = self.cache[ln]
                proof_, times_ = real.prove_air(prog_, tr_, pub_, o_)
                return proof_, pub_, times_

            def set_threads(self, t):
                real.set_threads(t)

        orc = _ProgramOracle()
    probe_log = min(14 if width > 8 else 18, log_n)      # large enough for the transforms' inner parallelism to show (a 2^16 probe picked 16 threads of 256)
    best = None
    for t in cpu_thread_candidates(ncpu):
        orc.set_threads(t)
        orc.prove_fib_aux(width, min(12, log_n), aux[0], aux[1], opt.to_list(), D=aux[2])
        _, _, probe = orc.prove_fib_aux(width, probe_log, aux[0], aux[1], opt.to_list(), D=aux[2])
        if best is None or probe["total"] < best[1]:
            best = (t, probe["total"])
    cores = best[0]
    orc.set_threads(cores)
    if args.cpu_sample_log_n:
        s_log_n = min(args.cpu_sample_log_n, log_n)
    else:
        # 1 warm-up + 5 timed runs must stay within about half a minute of host time
        projected = best[1] * (1 << max(log_n - probe_log, 0)) * 1.3
        s_log_n = log_n
        while s_log_n > probe_log and projected > 5.0:
            s_log_n -= 1
            projected /= 2
    runs, walls = [], []
    cproof = None
    for i in range(6):                                # run 0 = warm-up
        t1 = time.perf_counter()
        cproof, _, ctimes = orc.prove_fib_aux(width, s_log_n, aux[0], aux[1], opt.to_list(), D=aux[2])
        if i:
            runs.append(ctimes["total"])
            walls.append(time.perf_counter() - t1)
    if s_log_n == log_n:
        assert cproof == first_proof, "GPU and CPU proofs differ"
    med = median(runs)
    # the reference binary itself runs Winterfell single-threaded: one thread on a smaller sample
    orc.set_threads(1)
    st_log = min(log_n, 16 if width <= 8 else 13)
    _, _, st = orc.prove_fib_aux(width, st_log, aux[0], aux[1], opt.to_list(), D=aux[2])
    orc.set_threads(cores)
    return {
        "value": (1 << s_log_n) * cols / med, "unit": "cells/s", "cores": cores, "kind": "port",
        "sample": f"complete proofs of a 2^{s_log_n} x {cols} trace of the same AIR with the same options: 1 warm-up + median of 5 runs "
                  f"(prover time per run {', '.join(f'{r:.2f}' for r in runs)} s; median {med:.2f} s; wall incl. trace generation {median(walls):.2f} s); "
                  f"OpenMP port, single-run thread sweep over {cpu_thread_candidates(ncpu)} on a 2^{probe_log} probe picked {cores}; "
                  f"host has {ncpu} logical CPUs" + ("; proof bytes identical to the GPU's" if s_log_n == log_n else ""),
        "runs_s": runs,
        "single_thread": {"value": (1 << st_log) * cols / st["total"], "unit": "cells/s",
                          "sample": f"one complete proof of a 2^{st_log} x {cols} trace, 1 thread (the reference binary runs Winterfell single-threaded)"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="fib_2^20x2_blowup8_blake2s_base", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-log-n", type=int, default=0,
                    help="trace size of the bounded CPU-baseline sample (0 = the full workload if six runs of it project to <= 30 s, else smaller)")
    ap.add_argument("--concurrent", type=int, default=0,
                    help="proofs IN FLIGHT per GPU: each on its own HIP stream (context) driven by its own host thread inside the library; "
                         "1 = strictly one proof at a time; 0 (default) = 8, fewer when 8 working sets would not fit comfortably in HBM "
                         "(throughput plateaus at 8 on the 2^20 x 2 workload)")
    ap.add_argument("--batch", type=int, default=0,
                    help="traces per step per GPU (a step = one batch of independent traces, each proven completely); 0 (default) = "
                         "8 x the proofs in flight for the small workloads (so that 20 steps are > 2 s of GPU work), 1 x otherwise")
    ap.add_argument("--resident", action="store_true",
                    help="time the HBM-resident variant as `value` (traces uploaded before the clock) instead of the host-memory hand-over")
    ap.add_argument("--stages", action="store_true", help="also print per-stage ms and the per-kernel table to stderr")
    ap.add_argument("--mode", default="replicas", choices=["replicas", "sharded"],
                    help="N > 1: replicas = every GPU proves its own traces (weak scaling, default); sharded = ONE proof per step "
                         "proven cooperatively by all GPUs (strong scaling)")
    ap.add_argument("--sharded-check-world", type=int, default=-1,
                    help="ranks of the sharded-proof side measurement run after the timed region (default: N when N > 1, else "
                         "off; on a 1-GPU box the ranks share the GPU over gloo, which checks the path but not its speed)")
    ap.add_argument("--no-air-program", action="store_true", help="skip the AIR-as-data leg (interpreter vs hard-wired constraint kernel)")
    ap.add_argument("--sharded-workloads", default="fib_2^20x2_blowup8_blake2s_base,fib_2^24x2_blowup8_blake2s_base,"
                                                     "standin_miden_shape_2^22x(72+9aux)_deg8_fold4")
    ap.add_argument("--sharded-timeout", type=float, default=300.0)
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing aid for a 1-GPU box: every rank uses cuda:0 and the ranks talk over gloo (exercises the whole "
                         "N > 1 control flow; the numbers then describe N processes sharing one GPU)")
    ap.add_argument("--shard-comm", default="auto", choices=["auto", "rccl", "torch"],
                    help="exchanges of a sharded proof: rccl = the library's native communicator (aero_rccl_*), torch = torch.distributed "
                         "collectives on the device buffers, auto = rccl when every rank has its own GPU")
    ap.add_argument("--exchange-chunks", type=int, default=0,
                    help="sharded proofs: cut every commitment's exchange into this many pieces per peer, overlapped with the row hashing "
                         "(AERO_EXCHANGE_CHUNKS; 0 = the library's default, one exchange) - for an A/B on a node with links")
    ap.add_argument("--shard-worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.exchange_chunks > 0:
        os.environ["AERO_EXCHANGE_CHUNKS"] = str(args.exchange_chunks)      # read by every rank's prover (children inherit it)
    if args.shard_worker:
        return shard_worker_main(args)

    if "RANK" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: be the launcher (nothing in this process has touched the GPU; torch is not even imported)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank, local_rank, world = rank_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world} (launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                         "or run `python bench.py --gpus N` without RANK/WORLD_SIZE in the environment and it starts its own ranks)")

    import numpy as np
    import torch
    import aero_amd

    dist = None
    if args.share_gpu:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.share_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    log_n, width, over = WORKLOADS[args.workload]
    opt = make_options(aero_amd, over)

    if args.mode == "sharded":
        if world < 2:
            raise SystemExit("--mode sharded needs --gpus N with N > 1")
        res = sharded_measure(args.workload, args.steps, args.warmup, rank, world, local_rank, dist, torch,
                              "torch" if args.share_gpu else args.shard_comm)
        if rank == 0:
            E = 2
            bpc = 168 + 8 * E + (1259 + 176 * E) / width + ((162 + 176 * E) / width if opt.field_extension == 2 else 0)
            print(json.dumps({
                "metric": "trace_cells_per_sec", "value": res["value"], "unit": "cells/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": res["ms_per_proof"], "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": args.workload, "trace_rows": 1 << log_n, "trace_cols": width, "blowup": opt.blowup_factor,
                           "h2d_included": True,
                           "hand_over": "trace in pinned host memory; every rank copies its share of the columns inside the timed region (aero_prove_fib_sharded_host)",
                           "parallelism": f"ONE proof sharded over {world} GPUs by LDE coset; every rank copies and interpolates W / {world} columns, all-gather of the "
                                          "coefficients; per commitment: all-to-all of rows or leaf digests + all-gather of subtree roots; one all-reduce for the openings"},
                "sharded_proof": res, "exchange_chunks": int(os.environ.get("AERO_EXCHANGE_CHUNKS", "1")),
                "path_roofline": {"bytes_per_cell": bpc, "achieved_GBps": res["value"] * bpc / 1e9 / world,
                                  "frac_of_hbm_peak": res["value"] * bpc / 1e9 / world / HBM_PEAK_GBS},
            }), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    cols = trace_cols(width, over)
    if args.concurrent > 0:
        S = args.concurrent
    else:
        # working set of one proof ~ 1.5 x the LDE matrices (trace + aux + composition + DEEP columns) over N = blowup * n rows
        est = 1.5 * 8 * (opt.blowup_factor << log_n) * (cols + 4)
        S = int(max(1, min(8, (120 << 30) // est)))
    # traces per step: a multiple of the proofs in flight; small workloads take 8 rounds per step so that the driver's 20 steps are > 2 s
    per_slot = (max(1, args.batch // S) if args.batch > 0 else (8 if (cols << log_n) <= (1 << 22) else 1))
    batch = per_slot * S
    # S contexts on this GPU, each driven by its own worker thread INSIDE the library (aero_pool_*): the whole timed region is one C call
    pool = aero_amd.Pool(local_rank, S)
    ctxs = [pool.ctx(i) for i in range(S)]
    program = over.get("program")
    if program:
        over["_air"] = aero_amd.Air(aero_amd.synth_vm_program(log_n, *program))
        trace, over["_pub"] = aero_amd.synth_vm_trace(log_n, program[0])
        assert trace.shape[0] == width
    else:
        trace = aero_amd.fib_trace(width, log_n)      # synthetic data, pure function of (width, log_n)
    # the hand-over the metric is defined on: trace in (pinned) HOST memory -> proof bytes in
    # host memory; every proof starts with the host-to-device copy of its trace on its own stream. One pinned buffer per slot.
    hosts = [aero_amd.PinnedTrace(trace, device=local_rank) for _ in range(S)]      # on the GPU's NUMA node where the box names one
    devs = [c.trace_upload(trace) for c in ctxs]      # HBM-resident copies for the secondary figure and the traced pass
    del trace
    ctx, dev = ctxs[0], devs[0]
    aux = over.get("aux") or (0, 0, 2)
    h2d = not args.resident

    def run_rounds(rounds, from_host):
        if program:
            return [(p_, over["_pub"]) for p_ in pool.prove_air(over["_air"], hosts if from_host else devs, over["_pub"], opt, rounds=rounds)]
        if from_host:
            return pool.prove_fib_host(hosts, opt, aux, rounds=rounds)
        return pool.prove_fib(devs, opt, aux, rounds=rounds)

    # ---- warmup (untimed): W steps of the timed path, every stream ----
    proof = None
    for _ in range(max(1, args.warmup)):
        proof, pub = run_rounds(per_slot, h2d)[0]
    # one more untimed pass with every launch bracketed by HIP events: per-kernel table, picks the dominant kernel
    # (steady state: tables and code objects are already resident after the warmup)
    TRACED = 3
    prove_call(ctx, dev, opt, over)                   # settle after the all-streams warmup
    ctx.set_kernel_timing(True)
    for _ in range(TRACED):
        first_proof, pub = prove_call(ctx, dev, opt, over)
    table = {k: (c / TRACED, ms / TRACED, b / TRACED) for k, (c, ms, b) in ctx.kernel_timing_report().items()}
    ctx.set_kernel_timing(False)
    assert proof == first_proof, "host-trace and resident-trace proofs differ"
    proof_len = len(first_proof)
    dominant = max(table.items(), key=lambda kv: kv[1][1])[0]

    # single-proof wall-clock (one stream, nothing else on the GPU): the "proof-gen wall-clock" half of the metric, trace in
    # pinned host memory -> proof bytes (H2D included), and the same with the trace already resident
    def one_proof_ms(src, reps=0):
        # median of enough repetitions to spend about 0.2 s (7 at least, 60 at most): seven runs of a 2.3 ms proof right behind a batch
        # read 2 - 3 % high and noisy against tools/single_latency.py's 300
        if not reps:
            t1 = time.perf_counter()
            prove_call(ctx, src, opt, over) if program else ctx.prove_fib_aux(src, aux[0], aux[1], opt, aux_degree=aux[2])
            first = time.perf_counter() - t1
            reps = int(max(7, min(60, 0.2 / max(first, 1e-4))))
        ts = []
        for _ in range(reps):
            t1 = time.perf_counter()
            prove_call(ctx, src, opt, over) if program else ctx.prove_fib_aux(src, aux[0], aux[1], opt, aux_degree=aux[2])
            ts.append((time.perf_counter() - t1) * 1e3)
        return median(ts)

    # the link the hand-over crosses: this box's pinned host-to-device rate on THIS trace buffer (a roofline of every H2D-inclusive figure:
    # cells/s <= rate / 8 B; tools/ubench_h2d.hip measures the same thing stand-alone -> profiles/ceilings.json)
    def h2d_rate_GBps(reps=5):
        flat = torch.from_numpy(hosts[0].array.reshape(-1).view(np.int64))
        d = torch.empty(flat.shape, dtype=torch.int64, device=torch.device("cuda", local_rank))
        best = 0.0
        for _ in range(reps):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            d.copy_(flat)
            torch.cuda.synchronize()
            best = max(best, flat.numel() * 8 / (time.perf_counter() - t1) / 1e9)
        del d
        return best

    barrier()
    link_GBps = h2d_rate_GBps()
    single_ms = one_proof_ms(hosts[0])
    single_resident_ms = one_proof_ms(dev)

    # ---- timed region: exactly K steps, barrier + synchronize on both sides. One step = one batch of `batch` independent
    # traces; the S streams run their K * batch / S proofs back to back inside the library (no artificial join between steps). ----
    ctx.set_kernel_timing(True, only_kernel=dominant)
    last = [None] * S

    def all_steps():
        for i, (p, _) in enumerate(run_rounds(args.steps * per_slot, h2d)):
            last[i] = p

    dt = timed_steps(all_steps, 1, barrier)
    dom_rep = ctx.kernel_timing_report().get(dominant, (0, 0.0, 0.0))
    ctx.set_kernel_timing(False)
    assert all(p == first_proof for p in last), "non-deterministic proof bytes"
    dt = max_over_ranks(dt, dist, "cpu" if args.share_gpu else "cuda")
    # secondary figure: the other hand-over (HBM-resident when `value` includes H2D and vice versa), a quarter of the steps
    other_steps = max(1, args.steps // 4)
    dt_other = timed_steps(lambda: run_rounds(other_steps * per_slot, not h2d), 1, barrier)
    dt_other = max_over_ranks(dt_other, dist, "cpu" if args.share_gpu else "cuda")

    cells = (1 << log_n) * cols * batch                # cells per step (one batch)
    value = aggregate_value(cells, args.steps, world, dt)
    other_value = aggregate_value(cells, other_steps, world, dt_other)
    out = {
        "metric": "trace_cells_per_sec",
        "value": value,
        "unit": "cells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": args.workload, "trace_rows": 1 << log_n, "trace_cols": width, "blowup": opt.blowup_factor,
                   "num_queries": opt.num_queries, "grinding": opt.grinding_factor, "fri_fold": opt.fri_folding_factor,
                   "field_extension": "quadratic" if opt.field_extension == 2 else "none", "hash": "blake2s_256",
                   "aux_segment": ({"columns": over["aux"][0], "random_elements": over["aux"][1], "constraint_degree": over["aux"][2],
                                    "composition_columns": 2 if over["aux"][2] <= 2 else (4 if over["aux"][2] <= 4 else 8),
                                    "air": ("VM-shaped constraint PROGRAM (aero_air_synth_vm_*: 76 + 9 transition constraints up to degree 8, periodic columns, interior / periodic assertions, auxiliary running products with denominators), evaluated by a kernel compiled from it at run time; the Miden AIR is not in the reference mount"
                                            if program else "synthetic stand-in (prefix-product columns); the Miden AIR is not in the reference mount")}
                                   if over.get("aux") else None),
                   "h2d_included": h2d,
                   "hand_over": ("trace in pinned host memory -> proof bytes in host memory; the host-to-device copy of every trace is inside the timed region"
                                 if h2d else "trace resident in HBM before the timed region"),
                   "traces_per_step_per_gpu": batch, "proofs_in_flight_per_gpu": S, "proof_bytes": proof_len,
                   "parallelism": f"{world} GPU(s) x {S} independent proofs in flight per GPU (one HIP stream + one library worker thread each), no data-path collective"},
        "timed_region_s": dt,
        "ms_per_proof": 1e3 * dt / (args.steps * batch),
        ("hbm_resident_value" if h2d else "h2d_inclusive_value"): other_value,
        "device_bytes_peak_per_proof_in_flight": ctx.memory_stats()[1],
        "single_proof_ms": single_ms if h2d else single_resident_ms,
        "single_proof_ms_hbm_resident": single_resident_ms,
        "single_proof_ms_h2d_included": single_ms,
        "single_proof_value": (1 << log_n) * cols / ((single_ms if h2d else single_resident_ms) * 1e-3),
    }

    # host link: bytes of trace handed over per second against the measured pinned rate of the same buffer
    h2d_value = value if h2d else other_value
    link_used = h2d_value * 8 / 1e9 / world
    out["pcie"] = {"h2d_pinned_GBps_measured": link_GBps, "achieved_GBps": link_used, "frac": link_used / link_GBps if link_GBps else None,
                   "what": "8 B per trace cell of the H2D-inclusive figure / this box's pinned host-to-device rate on the same buffer (one copy at a time, best of 5)",
                   "binds": bool(link_GBps and link_used / link_GBps >= 0.9)}
    # prove-then-verify (aero_ctx_set_self_verify; both reference callers verify before a proof leaves the process: main.rs:47,
    # proving_worker.rs:196-203): host time of the library's verifier on this proof - BESIDE the metric, which SURVEY 8(d) defines without
    # verification; the check is off for single-GPU proofs by default and on for every sharded one
    def self_verify_ms(reps=5):
        ts = []
        for _ in range(reps):
            t1 = time.perf_counter()
            if program:
                aero_amd.verify_air(first_proof, over["_pub"], over["_air"], expected_log_n=log_n)
            else:
                aero_amd.verify_fib(first_proof, pub, aux, expected_log_n=log_n)
            ts.append((time.perf_counter() - t1) * 1e3)
        return median(ts)
    try:
        sv = self_verify_ms()
        out["self_verify"] = {"ms_per_proof": sv, "frac_of_single_proof": sv / single_resident_ms if single_resident_ms else None,
                              "in_timed_region": False,
                              "what": "host time of aero_verify_* (every check of src/stark_verifier + the OOD constraint check) on one proof of this workload; "
                                      "AERO_SELF_VERIFY=1 / aero_ctx_set_self_verify(ON) adds it to every proof, sharded proofs have it on by default"}
    except Exception as e:      # a verifier rejection here is a failed run, not a missing field
        raise SystemExit(f"self-verify of the bench proof failed: {e}")
    numa_node, workers_pinned = pool.placement()
    out["host_placement"] = {"gpu_numa_node": numa_node, "worker_threads_bound": workers_pinned, "of": S,
                             "what": "pool worker threads bound to the CPUs of the GPU's NUMA node, pinned trace buffers allocated with that node preferred (inside the rank process: no numactl, no re-exec); -1 = the box names no node"}
    if rank == 0:
        calls, ms, abytes = dom_rep
        achieved = (abytes / (ms * 1e-3)) / 1e9 if ms > 0 else 0.0
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)
            traffic = pmc.get(args.workload, {}).get(dominant, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
        total_ms = sum(v[1] for v in table.values())
        # Two measurements of the dominant kernel: inside the timed region (the contract's numbers; contended when several
        # proofs are in flight) and alone on the GPU (the single-stream traced pass; the kernel's own quality).
        s_calls, s_ms, s_bytes = table[dominant]
        s_achieved = (s_bytes / (s_ms * 1e-3)) / 1e9 if s_ms > 0 else 0.0
        comp_per_launch = (1 << log_n) * opt.blowup_factor * ((width + 1) // 2 + 7 / 8)
        is_hash = dominant.startswith("merkle_") or dominant.startswith("hash_")
        # measured ceilings of this chip, from the tracked raw outputs of the microbenchmarks (tools/ceilings.sh -> profiles/)
        ceilings = {}
        try:
            with open(os.path.join(ROOT, "profiles", "ceilings.json")) as f:
                ceilings = json.load(f)
        except Exception:
            ceilings = {}
        blake_ceiling = ceilings.get("blake2s_in_register_ceiling_Gcomp_per_s")
        # nominal VALU rate: 256 CUs x 128 lanes x 2.4 GHz; one compression = 957 VALU instructions (profiles/r2_isa_merkle_leaf8_rows2.json)
        blake_nominal = 256 * 128 * 2.4e9 / 957 / 1e9
        out["roofline"] = {
            # contract numbers: HIP events on the launch stream INSIDE the timed region (stream 0's launches of this kernel; with
            # several proofs in flight a launch shares the CUs with the other streams, so this duration is a property of the
            # mix, not of the kernel - the kernel alone is reported under "one_proof_in_flight")
            # `bound` names what limits the kernel; achieved / peak / frac stay the HBM figures the contract defines
            "bound": "valu" if is_hash else "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "limiter": ("valu: BLAKE2s is 32-bit integer issue-bound (957 VALU instructions = about 1208 full-rate issue slots per 64-byte block, 16 of which bytes come from HBM; profiles/r2_isa_merkle_leaf8_rows2.json); "
                        "the HBM fraction is priced as the contract asks but is not what limits this kernel - see valu_view") if is_hash
                       else "valu (64-bit modular arithmetic on the 32-bit VALU) unless the achieved fraction says otherwise",
            "avg_launch_us": 1e3 * ms / max(calls, 1), "launches_timed": calls, "proofs_in_flight": S,
            "algorithmic_bytes_per_launch": s_bytes / max(s_calls, 1),
            "launches_per_proof": s_calls, "share_of_kernel_time": s_ms / total_ms if total_ms else None,
            "measured": "HIP events on the launch stream over the timed region (stream 0)",
            "one_proof_in_flight": {"achieved": s_achieved, "frac": s_achieved / HBM_PEAK_GBS, "avg_launch_us": 1e3 * s_ms / max(s_calls, 1),
                                    "measured": "HIP events, 3 traced proofs with nothing else on the GPU (right before the timed region)"},
            "valu_view": ({"what": "BLAKE2s compressions/s of this kernel (8 leaves + 7 nodes per thread)",
                           "achieved_Gcomp_per_s": comp_per_launch / (1e-3 * s_ms / max(s_calls, 1)) / 1e9,
                           "in_register_ceiling_Gcomp_per_s": blake_ceiling,
                           "frac_of_in_register_ceiling": (comp_per_launch / (1e-3 * s_ms / max(s_calls, 1)) / 1e9 / blake_ceiling) if blake_ceiling else None,
                           "nominal_valu_ceiling_Gcomp_per_s": blake_nominal,
                           "frac_of_nominal_valu_rate": comp_per_launch / (1e-3 * s_ms / max(s_calls, 1)) / 1e9 / blake_nominal,
                           "ceiling_source": "profiles/ceilings.json + profiles/r3_ubench_valu.txt (tools/ceilings.sh: the register-only BLAKE2s loops of tools/ubench_valu.hip on this chip); nominal = 256 CUs x 128 lanes x 2.4 GHz / 957 VALU instructions per compression (profiles/r2_isa_merkle_leaf8_rows2.json)"}
                          if dominant == "merkle_leaf8_kernel" else None),
            "next_kernels": [{"kernel": k, "share_of_kernel_time": v[1] / total_ms if total_ms else None,
                              "achieved": (v[2] / (v[1] * 1e-3)) / 1e9 if v[1] > 0 else 0.0,
                              "frac": (v[2] / (v[1] * 1e-3)) / 1e9 / HBM_PEAK_GBS if v[1] > 0 else 0.0,
                              "avg_launch_us": 1e3 * v[1] / max(v[0], 1)}
                             for k, v in sorted(table.items(), key=lambda kv: -kv[1][1])[1:4]],
        }
        # the NTT family is VALU-bound as well: its register transforms against their in-register rate on this chip
        # (tools/ubench_dft.hip, raw output tracked under profiles/): a contiguous first pass = 3 shift stages + phase B, a strided
        # radix-64 pass = 6 shift stages + one multiplication; the forward passes of an LDE are 1 first + 2 strided per element
        try:
            dft = {}
            variant = None
            for line in open(os.path.join(ROOT, "profiles", "r3_ubench_dft.txt")):
                if line.startswith("variant:"):
                    variant = "lazy" if "WITHOUT" in line else "canonical"
                elif variant == "canonical" and "Gelem/s" in line:
                    dft[line.split("  ")[0].strip()] = float(line.split()[-2])
            mul_rate = ceilings.get("valu_Gop_per_s", {}).get("goldilocks mul")
            pb, d32, d8 = dft["phase B (mul + dft32 + 2 mul) per element"], dft["dft32 shift-twiddle stages only"], dft["dft8 (radix-8 butterfly of the LDS rounds)"]
            first_pass = 1.0 / (1.0 / pb + 1.0 / d8)
            strided = 1.0 / (6.0 / (5.0 * d32) + (1.0 / mul_rate if mul_rate else 0.0))
            lde_mix = 3.0 / (1.0 / first_pass + 2.0 / strided)
            if "ntt_fwd_pass" in table and table["ntt_fwd_pass"][1] > 0:
                c_, ms_, b_ = table["ntt_fwd_pass"]
                passes = (b_ * 3.0 / 41.0) / (ms_ * 1e-3) / 1e9          # 9 + 16 + 16 algorithmic bytes per element over the three passes
                out["roofline"]["ntt_valu_view"] = {
                    "what": "forward NTT passes (one proof in flight): element-passes per second against the register transforms' own rate",
                    "achieved_Gelem_passes_per_s": passes, "in_register_ceiling_Gelem_passes_per_s": lde_mix, "frac_of_in_register_ceiling": passes / lde_mix,
                    "ceilings_Gelem_per_s": {"contiguous_first_pass": first_pass, "strided_radix64_pass": strided},
                    "source": "profiles/r3_ubench_dft.txt (tools/ubench_dft.hip, canonical variant) + profiles/ceilings.json (goldilocks mul)"}
        except Exception:
            pass
        # whole-proof view: SURVEY 8d algorithmic bytes per cell, E = constraint-evaluation blowup = composition columns (2 for the degree-1
        # FibAir, 2 / 4 / 8 by the auxiliary constraint's degree, the program's own for a constraint program; until round 6 this said E = 2
        # for every workload, which priced the degree-8 shapes at 206 instead of 269 B per cell)
        E = 2
        if over.get("aux"):
            E = 2 if over["aux"][2] <= 2 else (4 if over["aux"][2] <= 4 else 8)
        if program:
            E = int(over["_air"].info().get("ce_blowup", E)) or E
        bpc = 168 + 8 * E + (1259 + 176 * E) / width
        if opt.field_extension == 2:
            bpc += (162 + 176 * E) / width
        out["path_roofline"] = {"bytes_per_cell": bpc, "achieved_GBps": value * bpc / 1e9 / world,
                                "frac_of_hbm_peak": value * bpc / 1e9 / world / HBM_PEAK_GBS}
        if args.stages:
            ctx.set_stage_timing(True)
            prove_call(ctx, dev, opt, over)
            print("stage ms:", json.dumps(ctx.last_stage_ms()), file=sys.stderr)
            ctx.set_stage_timing(False)
            for name, (c, m, b) in sorted(table.items(), key=lambda kv: -kv[1][1]):
                gbs = (b / (m * 1e-3)) / 1e9 if m > 0 else 0.0
                print(f"  {name:28s} calls/proof {c:7.1f}  ms/proof {m:8.3f}  alg GB/s {gbs:8.1f}", file=sys.stderr)

        if world == 1 and not args.no_cpu_baseline:
            sys.stdout.flush()
            out["cpu_baseline"] = cpu_baseline_leg(args, log_n, width, over, opt, first_proof)
        if world == 1 and not args.no_air_program:
            try:
                out["air_program"] = air_program_leg(aero_amd, ctx, local_rank)
            except Exception as e:
                out["air_program"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        check_world = args.sharded_check_world if args.sharded_check_world >= 0 else (world if world > 1 else 0)
        if check_world > 1:
            sys.stdout.flush()
            try:
                out["sharded_proof"] = spawn_sharded_check(check_world, args.sharded_workloads, 10, 2, args.sharded_timeout, args.shard_comm)
            except Exception as e:
                out["sharded_proof"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        print(json.dumps(out), flush=True)
    if dist is not None:
        # ranks > 0 wait on the rendezvous store (CPU) rather than in a collective, so that no RCCL kernel spins on their
        # GPUs while rank 0's worker group measures the sharded proof
        store = torch.distributed.distributed_c10d._get_default_store()
        if rank == 0:
            store.set("aero_bench_done", "1")
        else:
            import datetime
            store.wait(["aero_bench_done"], datetime.timedelta(seconds=args.sharded_timeout + 900))

    for d in devs:
        d.free()
    for h in hosts:
        h.release()
    pool.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
