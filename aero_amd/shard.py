"""Exchange steps of a sharded proof (include/aero_stark.h: aero_comm) carried by torch.distributed.

One process per GPU. On a multi-GPU node the product path is the library's NATIVE communicator (RcclComm below: RCCL over xGMI, exchanges
enqueued on the prover's stream) or, for ranks that are threads of one process, the in-library local group (LocalGroup). TorchComm is the
TEST data plane for ranks that are processes sharing one GPU (a 1-GPU box): the library hands the callbacks raw DEVICE pointers, which are
wrapped as torch tensors through `__cuda_array_interface__` (no copy); with backend "nccl" (= RCCL) the collectives run directly on those
tensors, with backend "gloo" they run by default on HOST copies made and written back here ("host" form), or - AERO_TORCHCOMM_GLOO=device,
the form rounds 1-5 ran - on the device tensors themselves, which gloo stages through host memory on streams of its own. Both gloo forms
are kept because one wrong proof was met under the device form in round 5 (profiles/r5_sharded_anomaly.md, r6_sharded_anomaly.md): the
stress loop (tests/shard_stress_worker.py) runs them side by side. This file is plumbing: the sharding itself (coset geometry, what is
exchanged and when) lives in aero_amd/csrc/prover.hip.
"""
import ctypes as C
import os

_A2A = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
_AG = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
_AR = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64)
_SR = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_uint64)


class CommStruct(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("user", C.c_void_p), ("all_to_all", _A2A), ("all_gather", _AG),
                ("all_reduce_sum_u64", _AR), ("min_peer_digests", C.c_uint32), ("flags", C.c_uint32), ("send_recv", _SR)]   # send_recv: NULL = all_gather fallback


class _DevPtr:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3, "strides": None}


def chunk_digests(torch, t, chunks):
    """32-byte fingerprints of the `chunks` equal pieces of a device byte tensor, computed ON the device (four wrapping 64-bit sums:
    plain, index-weighted, index-squared-weighted, and of a xor-shifted copy with odd weights) - cheap enough to take on every exchange of
    a stress loop (a 256 MiB buffer costs a few milliseconds), strong enough to tell any bit flip, any moved or missing piece. Not a hash."""
    w = t.view(torch.int64).view(chunks, -1)
    n = w.shape[1]
    i = torch.arange(1, n + 1, dtype=torch.int64, device=w.device)
    out = torch.stack([w.sum(1), (w * i).sum(1), (w * (i * i)).sum(1), ((w ^ (w >> 29)) * (2 * i + 1)).sum(1)], 1).cpu()
    return [out[c].numpy().tobytes().hex() for c in range(chunks)]


class TorchComm:
    """aero_comm whose callbacks run torch.distributed collectives on the default (or given) process group.

    gloo_tensors: "host" (default; env AERO_TORCHCOMM_GLOO) or "device" - see the module text; ignored for backend nccl.
    evidence (env AERO_SHARD_EVIDENCE=1): every callback appends {"op", "bytes", "send": [...], "recv": [...]} to self.evidence - one
        fingerprint (chunk_digests) per peer piece of what this rank sent and of what it received, taken after the collective. For a
        commitment that is: the rows or leaf digests sent to / received from every peer (all_to_all), this rank's subtree root (the
        all_gather's send) and the gathered top (its recv). tests/test_gpu_sharded.py: diagnose() lines them up across ranks.
    fault (tests only): {"op": "all_gather", "call": k, "chunk": q, "byte": b} flips one bit of byte b of piece q of what the k-th call of
        that kind received - the same on every rank that is given the fault."""

    def __init__(self, device=0, group=None, min_peer_digests=0, gloo_tensors=None, evidence=None, fault=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.device = torch.device("cuda", device)
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.calls = {"all_to_all": 0, "all_gather": 0, "all_reduce": 0}
        self.bytes_sent = 0
        self.last_error = None
        # one collective per exchange, chosen ONCE by backend (a fallback taken by one rank only would desynchronise the ranks):
        # gloo has no all_gather_into_tensor for device tensors, nccl (= RCCL) has
        self.gather_into_tensor = dist.get_backend(group) == "nccl"
        self.gloo_tensors = gloo_tensors or os.environ.get("AERO_TORCHCOMM_GLOO", "host")
        assert self.gloo_tensors in ("host", "device")
        self.keep_evidence = (os.environ.get("AERO_SHARD_EVIDENCE", "0") != "0") if evidence is None else bool(evidence)
        self.evidence = []
        self.fault = fault
        # keep the CFUNCTYPE objects alive for as long as the struct is in use
        self._a2a, self._ag, self._ar = _A2A(self._all_to_all), _AG(self._all_gather), _AR(self._all_reduce)
        self.struct = CommStruct(self.rank, self.world, None, self._a2a, self._ag, self._ar, min_peer_digests)

    def _t(self, ptr, nbytes):
        return self.torch.as_tensor(_DevPtr(ptr, nbytes), device=self.device)

    def _guard(self, fn):
        try:
            fn()
            self.torch.cuda.synchronize(self.device)
            return 0
        except Exception as e:  # the C side turns a non-zero status into AERO_E_COMM
            self.last_error = e
            return 1

    def _after(self, op, nbytes, send_t, send_chunks, recv_t, recv_chunks):
        """fault injection and evidence, behind the collective and its synchronisation"""
        f = self.fault
        if f and f["op"] == op and f["call"] == self.calls[op]:
            self.torch.cuda.synchronize(self.device)
            recv_t[f["chunk"] * (recv_t.numel() // recv_chunks) + f.get("byte", 0)] ^= 1
        if self.keep_evidence:
            self.torch.cuda.synchronize(self.device)
            self.evidence.append({"op": op, "bytes": int(nbytes), "send": chunk_digests(self.torch, send_t, send_chunks) if send_t is not None else [],
                                  "recv": chunk_digests(self.torch, recv_t, recv_chunks)})

    def _all_to_all(self, _user, send, recv, nbytes):
        def run():
            n = int(nbytes) * self.world
            s, r = self._t(send, n), self._t(recv, n)
            if self.gather_into_tensor or self.gloo_tensors == "device":
                self.dist.all_to_all_single(r, s, group=self.group)
            else:
                hs = s.cpu()
                hr = self.torch.empty_like(hs)
                self.dist.all_to_all_single(hr, hs, group=self.group)
                r.copy_(hr)
            self._after("all_to_all", nbytes, s, self.world, r, self.world)
            self.calls["all_to_all"] += 1
            self.bytes_sent += int(nbytes) * (self.world - 1)
        return self._guard(run)

    def _all_gather(self, _user, send, recv, nbytes):
        def run():
            out, inp = self._t(recv, int(nbytes) * self.world), self._t(send, int(nbytes))
            if self.gather_into_tensor:
                self.dist.all_gather_into_tensor(out, inp, group=self.group)
            elif self.gloo_tensors == "device":
                self.dist.all_gather(list(out.chunk(self.world)), inp, group=self.group)
            else:
                h = inp.cpu()
                parts = [self.torch.empty_like(h) for _ in range(self.world)]
                self.dist.all_gather(parts, h, group=self.group)
                out.copy_(self.torch.cat(parts))
            self._after("all_gather", nbytes, inp, 1, out, self.world)
            self.calls["all_gather"] += 1
            self.bytes_sent += int(nbytes) * (self.world - 1)
        return self._guard(run)

    def _all_reduce(self, _user, buf, count):
        def run():
            t = self._t(buf, int(count) * 8).view(self.torch.int64)   # wrapping two's-complement sum == u64 sum
            if self.gather_into_tensor or self.gloo_tensors == "device":
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            else:
                h = t.cpu()
                self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            self._after("all_reduce", int(count) * 8, None, 1, t.view(self.torch.uint8), 1)
            self.calls["all_reduce"] += 1
            self.bytes_sent += int(count) * 8
        return self._guard(run)


class RcclComm:
    """The library's NATIVE communicator (aero_rccl_*: RCCL over xGMI, exchanges enqueued on the context's stream). Python only
    moves the 128-byte id from rank 0 to the other ranks - `share_id` is any callable(bytes_or_None) -> bytes that broadcasts
    rank 0's value (default: torch.distributed.broadcast_object_list on the default group, any backend)."""

    def __init__(self, ctx, rank, world, share_id=None, min_peer_digests=0, agree=None):
        """share_id(bytes_or_None) -> bytes broadcasts rank 0's value; agree(int) -> int returns the most negative status over all
        ranks (0 = every rank is fine). Defaults: torch.distributed on the default group. With `agree`, a rank that cannot take
        part (no librccl, a refused `ncclCommInitRank`) makes EVERY rank raise - before the collective initialisation where that
        is knowable, after it otherwise - instead of leaving its peers inside RCCL's bootstrap for ever."""
        import aero_amd
        self.lib = aero_amd.lib()
        self.lib.aero_rccl_last_error.restype = C.c_char_p
        self.lib.aero_rccl_last_error.argtypes = [C.c_void_p]
        self.rank, self.world = rank, world
        self.h = None
        if share_id is None:
            share_id = _share_over_torch_dist if world > 1 else (lambda b: b)
            if agree is None and world > 1:
                agree = _agree_over_torch_dist
        # pre-flight on every rank: a rank without a usable RCCL is known BEFORE anybody enters ncclCommInitRank. Rank 0 asks for the id
        # (which binds librccl on the way); the others only bind it (aero_rccl_available creates nothing)
        buf = (C.c_uint8 * 128)()
        rc0 = self.lib.aero_rccl_unique_id(buf) if rank == 0 else self.lib.aero_rccl_available()
        text0 = self.lib.aero_rccl_last_error(None).decode() if rc0 != 0 else ""
        if agree is not None:
            worst = agree(rc0)
            if worst != 0:
                raise aero_amd.AeroError(rc0 if rc0 != 0 else worst, "RCCL communicator not created: " + (text0 or f"a peer rank reported status {worst} before initialisation"))
        uid, failure = None, None
        if rank == 0:
            if rc0 != 0:
                # the peers are waiting for the id: hand them the failure instead of leaving them in the broadcast for ever
                failure = (rc0, text0)
                uid = b"FAIL" + repr(failure).encode()
            else:
                uid = bytes(buf)
        uid = share_id(uid)
        if isinstance(uid, (bytes, bytearray)) and bytes(uid[:4]) == b"FAIL":
            raise aero_amd.AeroError(failure[0] if failure else -4, "rank 0 could not create the RCCL id: " + bytes(uid[4:]).decode(errors="replace"))
        if rc0 != 0:
            raise aero_amd.AeroError(rc0, text0)
        assert isinstance(uid, (bytes, bytearray)) and len(uid) == 128
        h = C.c_void_p()
        rc = self.lib.aero_rccl_create(ctx.h, C.c_int32(rank), C.c_int32(world), (C.c_uint8 * 128)(*uid), C.byref(h))
        text = self.lib.aero_rccl_last_error(None).decode() if rc != 0 else ""
        if agree is not None:
            worst = agree(rc)
            if worst != 0:
                if rc == 0:
                    self.lib.aero_rccl_destroy(h)
                raise aero_amd.AeroError(rc if rc != 0 else worst, "RCCL communicator not created: " + (text or f"a peer rank failed in aero_rccl_create (status {worst})"))
        elif rc != 0:
            raise aero_amd.AeroError(rc, text)
        self.h = h
        self.struct = CommStruct()
        rc = self.lib.aero_rccl_comm(self.h, C.c_uint32(min_peer_digests), C.byref(self.struct))
        assert rc == 0
        self._base = self._stats()

    def _stats(self):
        out = (C.c_uint64 * 4)()
        self.lib.aero_rccl_stats(self.h, out)
        return list(out)

    @property
    def calls(self):
        s = self._stats()
        return {"all_to_all": s[0], "all_gather": s[1], "all_reduce": s[2]}

    @property
    def bytes_sent(self):
        return self._stats()[3]

    def info(self):
        """aero_rccl_info: what RCCL itself reports about this communicator."""
        out = (C.c_int32 * 4)()
        self.lib.aero_rccl_info(self.h, out)
        return {"ranks_counted_by_rccl": out[0], "rank_as_rccl_numbers_it": out[1], "device_rccl_bound": out[2], "world_asked_for": out[3]}

    def error_text(self):
        return self.lib.aero_rccl_last_error(self.h).decode() if self.h else ""

    @property
    def last_error(self):
        return self.error_text() or None

    def close(self):
        if self.h:
            self.lib.aero_rccl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _agree_over_torch_dist(status):
    import torch.distributed as dist
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, int(status))
    return min(box)


def _share_over_torch_dist(uid):
    import torch.distributed as dist
    box = [uid]
    dist.broadcast_object_list(box, src=0)
    return box[0]


class LoopbackComm:
    """Timing aid for a 1-GPU box: an aero_comm for rank `rank` of `world` whose exchanges stay on the device (all_to_all
    copies send -> recv, all_gather replicates the local piece, all_reduce is a no-op). The proof it yields is NOT valid
    (peers' digests are replaced by this rank's own) but the rank performs exactly the kernels, transfers sizes and host
    round trips of a real sharded run, so its wall-clock is the per-rank compute time of a `world`-GPU proof."""

    def __init__(self, rank, world, device=0, min_peer_digests=0):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.rank, self.world = rank, world
        self.calls = {"all_to_all": 0, "all_gather": 0, "all_reduce": 0}
        self.bytes_sent = 0
        self.last_error = None
        self._a2a, self._ag, self._ar, self._sr = _A2A(self._all_to_all), _AG(self._all_gather), _AR(self._all_reduce), _SR(self._send_recv)
        self.struct = CommStruct(rank, world, None, self._a2a, self._ag, self._ar, min_peer_digests, 0, self._sr)

    def _t(self, ptr, nbytes):
        return self.torch.as_tensor(_DevPtr(ptr, nbytes), device=self.device)

    def _send_recv(self, _user, send, _to, recv, _from, nbytes):
        n = int(nbytes)
        self._t(recv, n).copy_(self._t(send, n))
        self.torch.cuda.synchronize(self.device)
        self.calls["all_to_all"] += 1
        self.bytes_sent += n
        return 0

    def _all_to_all(self, _user, send, recv, nbytes):
        n = int(nbytes) * self.world
        self._t(recv, n).copy_(self._t(send, n))
        self.torch.cuda.synchronize(self.device)
        self.calls["all_to_all"] += 1
        self.bytes_sent += int(nbytes) * (self.world - 1)
        return 0

    def _all_gather(self, _user, send, recv, nbytes):
        n = int(nbytes)
        self._t(recv, n * self.world).view(self.world, n).copy_(self._t(send, n).unsqueeze(0).expand(self.world, n))
        self.torch.cuda.synchronize(self.device)
        self.calls["all_gather"] += 1
        self.bytes_sent += n * (self.world - 1)
        return 0

    def _all_reduce(self, _user, buf, count):
        self.calls["all_reduce"] += 1
        self.bytes_sent += int(count) * 8
        return 0


class LocalGroup:
    """aero_local_group_* (comm_local.hip): the ranks of ONE sharded proof as threads of this process, exchanges = stream-ordered peer
    copies inside the library (no torch, no RCCL). `run(fn)` calls fn(rank, ctx, comm) on one Python thread per rank (the C calls
    release the GIL) and returns the per-rank results; a failing rank aborts the group so that nobody waits at a rendezvous."""

    class _Comm:
        def __init__(self, struct):
            self.struct = struct
            self.rank, self.world = struct.rank, struct.world
            self.last_error = None

    def __init__(self, world, devices=None, min_peer_digests=0):
        import aero_amd
        self.aero, self.world = aero_amd, world
        self.h = C.c_void_p()
        rc = aero_amd.lib().aero_local_group_create(C.c_uint32(world), C.byref(self.h))
        if rc != 0:
            raise aero_amd.AeroError(rc, "aero_local_group_create")
        self.ctxs = [aero_amd.Context((devices or [0] * world)[r]) for r in range(world)]
        self.comms = []
        for r in range(world):
            s = CommStruct()
            rc = aero_amd.lib().aero_local_group_comm(self.h, self.ctxs[r].h, C.c_int32(r), C.c_uint32(min_peer_digests), C.byref(s))
            if rc != 0:
                raise aero_amd.AeroError(rc, "aero_local_group_comm")
            self.comms.append(LocalGroup._Comm(s))

    def stats(self, rank):
        out = (C.c_uint64 * 4)()
        self.aero.lib().aero_local_group_stats(self.h, C.c_int32(rank), out)
        return {"all_to_all": out[0], "all_gather": out[1], "all_reduce": out[2], "bytes_sent": out[3]}

    def run(self, fn):
        from concurrent.futures import ThreadPoolExecutor

        def one(r):
            try:
                return fn(r, self.ctxs[r], self.comms[r])
            except BaseException:
                self.aero.lib().aero_local_group_abort(self.h)
                raise

        with ThreadPoolExecutor(self.world) as ex:
            futs = [ex.submit(one, r) for r in range(self.world)]
            return [f.result() for f in futs]

    def close(self):
        for c in self.ctxs:
            c.close()
        self.ctxs = []
        if self.h:
            self.aero.lib().aero_local_group_destroy(self.h)
            self.h = None
