"""aero_amd — Python harness binding of libaero_stark.so (the MI355X-native Winterfell proving backend).

This module is plumbing for tests, bench.py and the multi-GPU launcher: it loads the C-ABI library
(include/aero_stark.h) with ctypes and mirrors the reference's operator surface by name
(ProofOptions, Matrix, MerkleTree, Prover.prove ... — see aero_amd/csrc/prover.hpp for the file:line map).
There is NO CPU fallback: if the HIP library is missing or no GPU is visible, construction fails loudly.
Nothing here imports oracle/ or tests/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AERO_LIB_PATH") or os.path.join(_HERE, "libaero_stark.so")   # override: A/B runs of another build
CSRC = os.path.join(_HERE, "csrc")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "aero_stark.h")

u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)

P = 0xFFFFFFFF00000001

STAGE_NAMES = ["interpolate", "lde", "trace_commit", "constraints", "composition", "comp_commit", "ood", "deep", "fri",
               "grind", "queries", "total"]


class AeroError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"aero_stark error {code}: {msg}")
        self.code = code


def build(force=False):
    """Compile every HIP translation unit for gfx950 into aero_amd/libaero_stark.so (hipcc cross-compiles
    without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j4"])
    return LIB_PATH


class ProofOptions(C.Structure):
    """Mirror of winter_air::ProofOptions (convert_inputs.rs:54-66); byte order = proof-context order."""
    _fields_ = [("num_queries", C.c_uint8), ("blowup_factor", C.c_uint8), ("grinding_factor", C.c_uint8),
                ("hash_fn", C.c_uint8), ("field_extension", C.c_uint8), ("fri_folding_factor", C.c_uint8),
                ("fri_log_max_remainder", C.c_uint8)]

    @classmethod
    def with_96_bit_security(cls):
        return cls(27, 8, 16, 4, 1, 8, 8)

    def to_list(self):
        return [self.num_queries, self.blowup_factor, self.grinding_factor, self.hash_fn, self.field_extension,
                self.fri_folding_factor, self.fri_log_max_remainder]


class FibAirDesc(C.Structure):
    """aero_fib_air: the optional auxiliary segment of the built-in AIR."""
    _fields_ = [("aux_width", C.c_uint32), ("aux_rands", C.c_uint32), ("aux_degree", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the HIP library is required; there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.aero_last_error.restype = C.c_char_p
        L.aero_last_error.argtypes = [C.c_void_p]
        L.aero_device_count.restype = C.c_int32
        _lib = L
    return _lib


def _p8(a):
    return a.ctypes.data_as(u8p)


def _p64(a):
    return a.ctypes.data_as(u64p)


def fib_trace(width, log_n):
    """Synthetic Fibonacci trace, column-major (width, 2^log_n) uint64 (host)."""
    out = np.zeros((width, 1 << log_n), np.uint64)
    rc = lib().aero_fib_trace(C.c_uint32(width), C.c_uint32(log_n), _p64(out))
    if rc != 0:
        raise AeroError(rc, "fib_trace: width must be even and >= 2")
    return out


class PinnedTrace:
    """A host trace in pinned memory for asynchronous DMA; `.array` is the (width, n) uint64 matrix.
    The reference's traces live in host memory when `Prover::prove` is called (proving_worker.rs:140,465-467).

    The object OWNS its memory: the caller's array is copied once into a buffer the library's runtime allocated pinned
    (aero_host_alloc = hipHostMalloc). (Round 3 registered the caller's own numpy array: pinned and pageable hand-overs of "the same"
    trace aliased one buffer and a user-pointer registration lived on pages whose lifetime was the allocator's. A Rust host does the
    equivalent by filling a buffer from aero_host_alloc, or registers the Vec it owns with aero_host_register.)"""

    def __init__(self, trace: np.ndarray, device=None):
        """device: allocate the buffer on the NUMA node of that GPU (aero_host_alloc_near; falls back to aero_host_alloc)."""
        src = np.ascontiguousarray(trace, np.uint64)
        self.width, n = src.shape
        self.log_n = int(n).bit_length() - 1
        ptr = C.c_void_p()
        if device is None:
            rc = lib().aero_host_alloc(C.c_size_t(src.nbytes), C.byref(ptr))
        else:
            rc = lib().aero_host_alloc_near(C.c_size_t(src.nbytes), C.c_int32(device), C.byref(ptr))
        if rc != 0:
            raise AeroError(rc, lib().aero_last_error(None).decode())
        # The numpy views are built over an owner object whose finaliser frees the memory: a view the caller still holds (`t = pinned.array`,
        # a slice of it) keeps the buffer alive after release() / after this object is gone - no use-after-free in the host process.
        self._owner = _PinnedBlock.from_address(ptr.value, src.nbytes)
        flat = np.frombuffer(self._owner, dtype=np.uint64)
        self.array = flat.reshape(src.shape)
        self.array[...] = src

    def release(self):
        """Drops this object's reference; the memory is freed once no numpy view of it is alive any more."""
        self.array = None
        self._owner = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class _PinnedBlock:
    """Owner of one aero_host_alloc'd block, exposed through the buffer protocol (a ctypes array over the address); freed when the last
    numpy view of it is gone."""

    @staticmethod
    def from_address(addr, nbytes):
        holder = _PinnedHolder(addr)
        arr = (C.c_uint8 * nbytes).from_address(addr)
        arr._aero_holder = holder          # the ctypes array is the base object of every view: it carries the finaliser
        return arr


class _PinnedHolder:
    def __init__(self, addr):
        self.addr = addr

    def __del__(self):
        try:
            if self.addr:
                lib().aero_host_free(C.c_void_p(self.addr))
                self.addr = 0
        except Exception:
            pass


def proof_container(inputs: bytes, proof: bytes) -> bytes:
    """bincode ProofData{input_bytes, proof_bytes} (miden-proof-generator/src/lib.rs:1-6)."""
    out = u8p()
    n = C.c_size_t(0)
    a = np.frombuffer(inputs, np.uint8) if inputs else np.zeros(1, np.uint8)
    b = np.frombuffer(proof, np.uint8) if proof else np.zeros(1, np.uint8)
    rc = lib().aero_proof_container(_p8(a), C.c_size_t(len(inputs)), _p8(b), C.c_size_t(len(proof)), C.byref(out), C.byref(n))
    if rc != 0:
        raise AeroError(rc, "proof_container")
    data = C.string_at(out, n.value)
    lib().aero_free(out)
    return data


class Matrix:
    """Device column-major matrix handle (winter Matrix<Felt>)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle

    @property
    def shape(self):
        c = C.c_uint32(0)
        r = C.c_uint64(0)
        lib().aero_matrix_shape(self.h, C.byref(c), C.byref(r))
        return (c.value, r.value)

    @property
    def device_ptr(self):
        p = u64p()
        lib().aero_matrix_device_ptr(self.h, C.byref(p))
        return C.cast(p, C.c_void_p).value

    def download(self):
        cols, rows = self.shape
        out = np.zeros((cols, rows), np.uint64)
        self.ctx._ck(lib().aero_matrix_download(self.ctx.h, self.h, _p64(out)))
        return out

    def free(self):
        if self.h:
            lib().aero_matrix_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MerkleTree:
    def __init__(self, ctx, handle, root, n):
        self.ctx = ctx
        self.h = handle
        self.root = root
        self.n = n

    def prove_batch(self, positions) -> bytes:
        """MerkleTree::prove_batch + BatchMerkleProof::serialize_nodes."""
        pos = np.ascontiguousarray(positions, np.uint64)
        cap = 1 + len(pos) * (1 + 32 * 40)
        out = np.zeros(cap, np.uint8)
        n = C.c_size_t(0)
        self.ctx._ck(lib().aero_merkle_open_batch(self.ctx.h, self.h, _p64(pos), C.c_uint32(len(pos)), _p8(out), C.c_size_t(cap), C.byref(n)))
        return out[:n.value].tobytes()

    def nodes(self):
        out = np.zeros((2 * self.n, 32), np.uint8)
        self.ctx._ck(lib().aero_merkle_nodes(self.ctx.h, self.h, _p8(out)))
        return out

    def free(self):
        if self.h:
            lib().aero_tree_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class FriLayers:
    """Device FRI layers kept by aero_fri_build_layers (winter-fri FriProver after build_layers)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle

    def open(self, positions) -> bytes:
        """Serialised FriProof (the FRI section of StarkProof::to_bytes) for LDE-domain query positions."""
        pos = np.ascontiguousarray(positions, np.uint64)
        out = u8p()
        n = C.c_size_t(0)
        self.ctx._ck(lib().aero_fri_open(self.ctx.h, self.h, _p64(pos), C.c_uint32(pos.size), C.byref(out), C.byref(n)))
        data = C.string_at(out, n.value)
        lib().aero_free(out)
        return data

    def free(self):
        if self.h:
            lib().aero_fri_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One GPU + one HIP stream + a device memory pool."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        rc = lib().aero_ctx_create(C.c_int32(device), C.byref(self.h))
        if rc != 0:
            raise AeroError(rc, lib().aero_last_error(None).decode())

    def _ck(self, rc):
        if rc != 0:
            raise AeroError(rc, lib().aero_last_error(self.h).decode())

    def close(self):
        if self.h:
            lib().aero_ctx_destroy(self.h)
            self.h = None

    def selftest(self, samples=1 << 16, seed=1):
        """aero_selftest: device field arithmetic against a 128-bit host reference."""
        self._ck(lib().aero_selftest(self.h, C.c_uint32(samples), C.c_uint64(seed)))

    def synchronize(self):
        self._ck(lib().aero_ctx_synchronize(self.h))

    def set_self_verify(self, mode):
        """aero_ctx_set_self_verify: -1 / "auto" (default: on for proofs made by more than one rank), 0 / False off, 1 / True on. A proof the
        library's own verifier rejects then comes back as AeroError(code -8, AERO_E_SELF_VERIFY) instead of bytes."""
        m = -1 if mode in (-1, "auto", None) else int(bool(mode))
        self._ck(lib().aero_ctx_set_self_verify(self.h, C.c_int32(m)))

    def commit_trace(self, trace: "Matrix", options: ProofOptions, comm=None):
        """aero_commit_trace_sharded: the main segment's commitment alone (interpolate, extend, exchange, subtree, root all-gather).
        Returns (root, [subtree roots]) as bytes."""
        world = comm.world if comm is not None else 1
        root = (C.c_uint8 * 32)()
        subs = (C.c_uint8 * (32 * world))()
        rc = lib().aero_commit_trace_sharded(self.h, C.byref(comm.struct) if comm is not None else None, trace.h, C.byref(options), root, subs)
        if rc != 0 and getattr(comm, "last_error", None) is not None:
            raise AeroError(rc, f"{lib().aero_last_error(self.h).decode()} ({comm.last_error!r})")
        self._ck(rc)
        b = bytes(subs)
        return bytes(root), [b[32 * r:32 * r + 32] for r in range(world)]

    # ---- matrices / stage 1
    def trace_upload(self, trace: np.ndarray) -> Matrix:
        t = np.ascontiguousarray(trace, np.uint64)
        w, n = t.shape
        log_n = int(n).bit_length() - 1
        if (1 << log_n) != n:
            raise AeroError(-1, "trace length must be a power of two")
        h = C.c_void_p()
        self._ck(lib().aero_trace_upload(self.h, _p64(t), C.c_uint32(w), C.c_uint32(log_n), C.byref(h)))
        return Matrix(self, h)

    def trace_file_load(self, path):
        """Stream an AEROTRC file to the device (aero_trace_file_load). Returns (Matrix, air_id, (aux_width, aux_rands, aux_degree))."""
        h, aid, air = C.c_void_p(), C.c_uint32(0), FibAirDesc()
        self._ck(lib().aero_trace_file_load(self.h, os.fsencode(path), C.byref(h), C.byref(aid), C.byref(air)))
        return Matrix(self, h), aid.value, (air.aux_width, air.aux_rands, air.aux_degree)

    def interpolate_columns(self, trace: Matrix) -> Matrix:
        h = C.c_void_p()
        self._ck(lib().aero_interpolate_columns(self.h, trace.h, C.byref(h)))
        return Matrix(self, h)

    def evaluate_columns_over(self, polys: Matrix, log_blowup: int) -> Matrix:
        h = C.c_void_p()
        self._ck(lib().aero_evaluate_columns_over(self.h, polys.h, C.c_uint32(log_blowup), C.byref(h)))
        return Matrix(self, h)

    def poly_eval(self, polys: Matrix, z: int):
        cols, _ = polys.shape
        out = np.zeros(cols, np.uint64)
        self._ck(lib().aero_poly_eval(self.h, polys.h, C.c_uint64(z), _p64(out)))
        return out.tolist()

    # ---- hashing / Merkle
    def hash_rows(self, rows: np.ndarray) -> np.ndarray:
        """HashingWorkItem.data (n_rows, width) -> HashingResult.hashes (n_rows, 32)."""
        r = np.ascontiguousarray(rows, np.uint64)
        n, w = r.shape
        out = np.zeros((n, 32), np.uint8)
        self._ck(lib().aero_hash_rows(self.h, _p64(r) if n else None, C.c_uint32(w), C.c_uint64(n), _p8(out) if n else _p8(np.zeros(1, np.uint8))))
        return out

    def hash_matrix_rows(self, m: Matrix) -> np.ndarray:
        _, rows = m.shape
        out = np.zeros((rows, 32), np.uint8)
        self._ck(lib().aero_hash_matrix_rows(self.h, m.h, _p8(out)))
        return out

    def merkle_from_leaves(self, leaves: np.ndarray) -> MerkleTree:
        l = np.ascontiguousarray(leaves, np.uint8)
        h = C.c_void_p()
        root = np.zeros(32, np.uint8)
        self._ck(lib().aero_merkle_from_leaves(self.h, _p8(l), C.c_uint64(l.shape[0]), C.byref(h), _p8(root)))
        return MerkleTree(self, h, root.tobytes(), l.shape[0])

    def merkle_commit_rows(self, m: Matrix) -> MerkleTree:
        h = C.c_void_p()
        root = np.zeros(32, np.uint8)
        self._ck(lib().aero_merkle_commit_rows(self.h, m.h, C.byref(h), _p8(root)))
        return MerkleTree(self, h, root.tobytes(), m.shape[1])

    # ---- constraints / FRI / grinding
    def eval_constraints_fib(self, lde: Matrix, log_blowup, results, coeffs, field_extension=1, fragment_offset=0, num_fragments=1):
        """ConstraintComputeWorkItem -> ConstraintComputeResult for FibAir. Returns (frag_index, cols (3*deg, rows))."""
        w, N = lde.shape
        deg = 2 if field_extension == 2 else 1
        n = N >> log_blowup
        rows = 2 * n // num_fragments
        res = np.ascontiguousarray(results, np.uint64)
        co = np.ascontiguousarray(coeffs, np.uint64)
        assert co.size == 2 * deg * (w + w + w // 2), "coeffs: (alpha, beta) per transition constraint then per assertion"
        out = np.zeros((3 * deg, rows), np.uint64)
        fi = C.c_uint64(0)
        self._ck(lib().aero_eval_constraints_fib(self.h, lde.h, C.c_uint32(log_blowup), _p64(res), _p64(co), C.c_uint8(field_extension),
                                                 C.c_uint32(fragment_offset), C.c_uint32(num_fragments), _p64(out), C.byref(fi)))
        return fi.value, out

    def aux_columns_fib(self, trace: Matrix, air, rands, field_extension=1) -> Matrix:
        """The stand-in AIR's auxiliary columns from the main trace and the drawn elements (aero_aux_columns_fib)."""
        r = np.array(rands, dtype=np.uint64, ndmin=1)
        h = C.c_void_p()
        desc = FibAirDesc(*air)
        self._ck(lib().aero_aux_columns_fib(self.h, trace.h, C.byref(desc), _p64(r), C.c_uint8(field_extension), C.byref(h)))
        return Matrix(self, h)

    def eval_constraints_air(self, lde: Matrix, aux_lde, air, log_blowup, results, rands, coeffs, field_extension=1, fragment_offset=0, num_fragments=1):
        """ConstraintComputeWorkItem -> ConstraintComputeResult for FibAir with its auxiliary segment. Returns (frag_index, cols)."""
        w, N = lde.shape
        deg = 2 if field_extension == 2 else 1
        A, R, D = air
        Cc = 2 if (not A or D <= 2) else (4 if D <= 4 else 8)
        n = N >> log_blowup
        rows = Cc * n // num_fragments
        res = np.array(results, dtype=np.uint64, ndmin=1)
        co = np.array(coeffs, dtype=np.uint64, ndmin=1)
        rd = np.array(rands, dtype=np.uint64, ndmin=1) if A else None
        out = np.zeros((3 * deg, rows), np.uint64)
        fi = C.c_uint64(0)
        desc = FibAirDesc(A, R, D)
        self._ck(lib().aero_eval_constraints_air(self.h, lde.h, aux_lde.h if aux_lde is not None else None, C.byref(desc), C.c_uint32(log_blowup),
                                                 _p64(res), _p64(rd) if rd is not None else None, _p64(co), C.c_uint8(field_extension),
                                                 C.c_uint32(fragment_offset), C.c_uint32(num_fragments), _p64(out), C.byref(fi)))
        return fi.value, out

    def _msg_call(self, fn, data: bytes, *extra) -> bytes:
        buf = np.frombuffer(data, np.uint8)
        out, n = u8p(), C.c_size_t(0)
        self._ck(fn(self.h, _p8(buf), C.c_size_t(len(data)), *extra, C.byref(out), C.byref(n)))
        res = C.string_at(out, n.value)
        lib().aero_free(out)
        return res

    def worker_hash_rows(self, work_item: bytes) -> bytes:
        """bincode HashingWorkItem -> bincode HashingResult (aero_worker_hash_rows; messages: aero_amd.messages)."""
        return self._msg_call(lib().aero_worker_hash_rows, work_item)

    def worker_eval_constraints(self, work_item: bytes, air: "Air", pub=None) -> bytes:
        """bincode ConstraintComputeWorkItem -> bincode ConstraintComputeResult (aero_worker_eval_constraints); `air` = the program
        (the message carries no AIR identity), `pub` = its public inputs (None: the elements of the message's Miden PublicInputs)."""
        if pub is None:
            return self._msg_call(lib().aero_worker_eval_constraints, work_item, air.h, None, C.c_uint32(0))
        pb = np.array(pub, dtype=np.uint64, ndmin=1)
        return self._msg_call(lib().aero_worker_eval_constraints, work_item, air.h, _p64(pb), C.c_uint32(pb.size))

    # ---- program AIRs (include/aero_air.h)
    def prove_air(self, air: "Air", trace, pub, options: ProofOptions, comm=None) -> bytes:
        """Prover::prove for a program AIR: `trace` = device Matrix (resident) or host ndarray / PinnedTrace; `pub` = the program's
        public inputs. Returns the proof bytes."""
        pb = np.array(pub, dtype=np.uint64, ndmin=1) if len(pub) else np.zeros(1, np.uint64)
        proof, plen = u8p(), C.c_size_t(0)
        if isinstance(trace, Matrix):
            rc = lib().aero_prove_air(self.h, C.byref(comm.struct) if comm is not None else None, air.h, trace.h, _p64(pb), C.c_uint32(len(pub)),
                                      C.byref(options), C.byref(proof), C.byref(plen))
        else:
            t = trace.array if isinstance(trace, PinnedTrace) else np.ascontiguousarray(trace, np.uint64)
            if comm is not None:      # one proof over the communicator's ranks, each copying its share of the columns from host memory
                rc = lib().aero_prove_air_sharded_host(self.h, C.byref(comm.struct), air.h, _p64(t), C.c_uint32(int(t.shape[1]).bit_length() - 1), _p64(pb),
                                                       C.c_uint32(len(pub)), C.byref(options), C.byref(proof), C.byref(plen))
            else:
                rc = lib().aero_prove_air_host(self.h, air.h, _p64(t), C.c_uint32(int(t.shape[1]).bit_length() - 1), _p64(pb), C.c_uint32(len(pub)),
                                               C.byref(options), C.byref(proof), C.byref(plen))
        if rc != 0 and getattr(comm, "last_error", None) is not None:
            raise AeroError(rc, f"{lib().aero_last_error(self.h).decode()} ({comm.last_error!r})")
        self._ck(rc)
        data = C.string_at(proof, plen.value)
        lib().aero_free(proof)
        return data

    def validate_trace(self, air: "Air", trace: Matrix, pub, aux: Matrix = None, rands=None, field_extension=1):
        """`Trace::validate(&air)` on the device (aero_air_validate_trace). Returns None when the trace satisfies the program, else
        (row, "transition" | "assertion", index) of the first failing check; without `aux` only the main segment is checked."""
        pb = np.array(pub, dtype=np.uint64, ndmin=1) if len(pub) else np.zeros(1, np.uint64)
        rv = np.ascontiguousarray(rands, np.uint64) if rands is not None else None
        out = C.c_uint64(0)
        self._ck(lib().aero_air_validate_trace(self.h, air.h, trace.h, aux.h if aux is not None else None, _p64(pb), C.c_uint32(len(pub)),
                                               _p64(rv) if rv is not None else None, C.c_uint8(field_extension), C.byref(out)))
        if out.value == 0xFFFFFFFFFFFFFFFF:
            return None
        row, ident = out.value >> 24, out.value & 0xFFFFFF
        return (row, "assertion", ident & 0x7FFFFF) if ident & 0x800000 else (row, "transition", ident)

    def eval_constraints_program(self, air: "Air", lde: Matrix, aux_lde, log_blowup, pub, rands, coeffs, field_extension=1, fragment_offset=0,
                                 num_fragments=1):
        """ConstraintComputeWorkItem -> ConstraintComputeResult for a program AIR. Returns (frag_index, cols (divisors*deg, rows))."""
        _, N = lde.shape
        deg = 2 if field_extension == 2 else 1
        n = N >> log_blowup
        ncols = air.num_divisors(int(n).bit_length() - 1)
        rows = air.info()["ce_blowup"] * n // num_fragments
        pb = np.array(pub, dtype=np.uint64, ndmin=1) if len(pub) else np.zeros(1, np.uint64)
        co = np.array(coeffs, dtype=np.uint64, ndmin=1)
        rd = np.array(rands, dtype=np.uint64, ndmin=1) if aux_lde is not None else None
        out = np.zeros((ncols * deg, rows), np.uint64)
        fi = C.c_uint64(0)
        self._ck(lib().aero_eval_constraints_program(self.h, air.h, lde.h, aux_lde.h if aux_lde is not None else None, C.c_uint32(log_blowup), _p64(pb),
                                                     C.c_uint32(len(pub)), _p64(rd) if rd is not None else None, _p64(co), C.c_uint8(field_extension),
                                                     C.c_uint32(fragment_offset), C.c_uint32(num_fragments), _p64(out), C.byref(fi)))
        return fi.value, out

    def aux_columns_program(self, air: "Air", trace: Matrix, pub, rands, field_extension=1) -> Matrix:
        """`build_aux_segment` from the program's builders (aero_aux_columns_program)."""
        pb = np.array(pub, dtype=np.uint64, ndmin=1) if len(pub) else np.zeros(1, np.uint64)
        r = np.array(rands, dtype=np.uint64, ndmin=1)
        h = C.c_void_p()
        self._ck(lib().aero_aux_columns_program(self.h, air.h, trace.h, _p64(pb), C.c_uint32(len(pub)), _p64(r), C.c_uint8(field_extension), C.byref(h)))
        return Matrix(self, h)

    def composition_poly_program(self, air: "Air", numer_cols: np.ndarray, log_n: int, field_extension=1) -> Matrix:
        a = np.ascontiguousarray(numer_cols, np.uint64)
        h = C.c_void_p()
        self._ck(lib().aero_composition_poly_program(self.h, air.h, _p64(a), C.c_uint32(log_n), C.c_uint8(field_extension), C.byref(h)))
        return Matrix(self, h)

    def composition_poly_air(self, numer_cols: np.ndarray, log_n: int, num_columns: int, field_extension=1) -> Matrix:
        a = np.ascontiguousarray(numer_cols, np.uint64)
        h = C.c_void_p()
        self._ck(lib().aero_composition_poly_air(self.h, _p64(a), C.c_uint32(log_n), C.c_uint32(num_columns), C.c_uint8(field_extension), C.byref(h)))
        return Matrix(self, h)

    def fri_fold(self, values: np.ndarray, fold: int, alpha: int) -> np.ndarray:
        v = np.ascontiguousarray(values, np.uint64)
        out = np.zeros(max(v.size // fold, 1), np.uint64)
        self._ck(lib().aero_fri_fold(self.h, _p64(v), C.c_uint64(v.size), C.c_uint32(fold), C.c_uint64(alpha), _p64(out)))
        return out

    def grind(self, seed: bytes, bits: int) -> int:
        s = np.frombuffer(seed, np.uint8).copy()
        n = C.c_uint64(0)
        self._ck(lib().aero_grind(self.h, _p8(s), C.c_uint32(bits), C.byref(n)))
        return n.value

    # ---- composition polynomial / DEEP / FRI layers (the stages inside prove_after_constraint_eval)
    def composition_poly_fib(self, numer_cols: np.ndarray, log_n: int, field_extension=1) -> Matrix:
        a = np.ascontiguousarray(numer_cols, np.uint64)
        h = C.c_void_p()
        self._ck(lib().aero_composition_poly_fib(self.h, _p64(a), C.c_uint32(log_n), C.c_uint8(field_extension), C.byref(h)))
        return Matrix(self, h)

    def deep_compose(self, trace_lde: Matrix, comp_lde: Matrix, log_blowup, z, ood_frame, ood_evals, coeffs, field_extension=1) -> Matrix:
        # dtype given at conversion: lists of Python ints >= 2^63 must not pass through float64
        arr = [np.array(v, dtype=np.uint64, ndmin=1) for v in (z, ood_frame, ood_evals, coeffs)]
        h = C.c_void_p()
        self._ck(lib().aero_deep_compose(self.h, trace_lde.h, comp_lde.h, C.c_uint32(log_blowup), C.c_uint8(field_extension),
                                         _p64(arr[0]), _p64(arr[1]), _p64(arr[2]), _p64(arr[3]), C.byref(h)))
        return Matrix(self, h)

    def fri_build_layers(self, evals: Matrix, options: ProofOptions, seed: bytes):
        """Returns (FriLayers handle, [roots], coin seed after the remainder commitment)."""
        s_in = np.frombuffer(seed, np.uint8).copy()
        roots = np.zeros(32 * 40, np.uint8)
        n = C.c_uint32(0)
        s_out = np.zeros(32, np.uint8)
        h = C.c_void_p()
        self._ck(lib().aero_fri_build_layers(self.h, evals.h, C.byref(options), _p8(s_in), _p8(roots), C.c_size_t(roots.size), C.byref(n),
                                             _p8(s_out), C.byref(h)))
        return FriLayers(self, h), [roots[32 * i:32 * i + 32].tobytes() for i in range(n.value)], s_out.tobytes()

    # ---- whole proof
    def prove_fib(self, trace, options: ProofOptions):
        """Prover::prove for FibAir. `trace` is a device Matrix (resident) or a host ndarray (copied in).
        Returns (proof_bytes, public_inputs)."""
        proof = u8p()
        plen = C.c_size_t(0)
        if isinstance(trace, Matrix):
            w, _ = trace.shape
            pub = np.zeros(w // 2, np.uint64)
            rc = lib().aero_prove_fib(self.h, trace.h, C.byref(options), C.byref(proof), C.byref(plen), _p64(pub))
        else:
            t = trace.array if isinstance(trace, PinnedTrace) else np.ascontiguousarray(trace, np.uint64)
            w, n = t.shape
            pub = np.zeros(w // 2, np.uint64)
            rc = lib().aero_prove_fib_host(self.h, _p64(t), C.c_uint32(w), C.c_uint32(int(n).bit_length() - 1), C.byref(options),
                                           C.byref(proof), C.byref(plen), _p64(pub))
        self._ck(rc)
        data = C.string_at(proof, plen.value)
        lib().aero_free(proof)
        return data, pub.tolist()

    def prove_fib_sharded(self, comm, trace: "Matrix", options: ProofOptions):
        """ONE proof proven cooperatively by comm.world GPUs (aero_prove_fib_sharded); `comm` = aero_amd.shard.TorchComm.
        Every rank passes the whole trace and gets the same bytes as prove_fib. Returns (proof_bytes, public_inputs)."""
        proof = u8p()
        plen = C.c_size_t(0)
        w, _ = trace.shape
        pub = np.zeros(w // 2, np.uint64)
        rc = lib().aero_prove_fib_sharded(self.h, C.byref(comm.struct), trace.h, C.byref(options), C.byref(proof), C.byref(plen), _p64(pub))
        if rc != 0 and getattr(comm, "last_error", None) is not None:
            raise AeroError(rc, f"{lib().aero_last_error(self.h).decode()} ({comm.last_error!r})")
        self._ck(rc)
        data = C.string_at(proof, plen.value)
        lib().aero_free(proof)
        return data, pub.tolist()

    def prove_fib_sharded_host(self, comm, trace, options: ProofOptions, aux=(0, 0, 2)):
        """ONE proof over comm.world GPUs with the trace in HOST memory (aero_prove_fib_sharded_host): this rank copies only its
        share of the columns. Returns (proof_bytes, public_inputs)."""
        t = trace.array if isinstance(trace, PinnedTrace) else np.ascontiguousarray(trace, np.uint64)
        w, n = t.shape
        pub = np.zeros(w // 2, np.uint64)
        proof, plen = u8p(), C.c_size_t(0)
        air = FibAirDesc(*aux)
        rc = lib().aero_prove_fib_sharded_host(self.h, C.byref(comm.struct) if comm is not None else None, _p64(t), C.c_uint32(w),
                                               C.c_uint32(int(n).bit_length() - 1), C.byref(air), C.byref(options), C.byref(proof), C.byref(plen), _p64(pub))
        if rc != 0 and getattr(comm, "last_error", None) is not None:
            raise AeroError(rc, f"{lib().aero_last_error(self.h).decode()} ({comm.last_error!r})")
        self._ck(rc)
        data = C.string_at(proof, plen.value)
        lib().aero_free(proof)
        return data, pub.tolist()

    def prove_fib_aux(self, trace: "Matrix", aux_width, aux_rands, options: ProofOptions, comm=None, aux_degree=2):
        """FibAir plus one auxiliary segment of `aux_width` columns built from `aux_rands` coin elements, aux transition
        constraint of degree `aux_degree` (aero_prove_fib_air); comm = None or a shard communicator.
        Returns (proof_bytes, public_inputs)."""
        proof = u8p()
        plen = C.c_size_t(0)
        air = FibAirDesc(aux_width, aux_rands, aux_degree)
        if isinstance(trace, (PinnedTrace, np.ndarray)):      # trace in host memory: the copy is part of the call
            assert comm is None, "sharded proofs take a device-resident trace"
            t = trace.array if isinstance(trace, PinnedTrace) else np.ascontiguousarray(trace, np.uint64)
            w, n = t.shape
            pub = np.zeros(w // 2, np.uint64)
            rc = lib().aero_prove_fib_air_host(self.h, _p64(t), C.c_uint32(w), C.c_uint32(int(n).bit_length() - 1), C.byref(air), C.byref(options),
                                               C.byref(proof), C.byref(plen), _p64(pub))
            self._ck(rc)
            data = C.string_at(proof, plen.value)
            lib().aero_free(proof)
            return data, pub.tolist()
        w, _ = trace.shape
        pub = np.zeros(w // 2, np.uint64)
        rc = lib().aero_prove_fib_air(self.h, C.byref(comm.struct) if comm is not None else None, trace.h, C.byref(air),
                                      C.byref(options), C.byref(proof), C.byref(plen), _p64(pub))
        if rc != 0 and getattr(comm, "last_error", None) is not None:
            raise AeroError(rc, f"{lib().aero_last_error(self.h).decode()} ({comm.last_error!r})")
        self._ck(rc)
        data = C.string_at(proof, plen.value)
        lib().aero_free(proof)
        return data, pub.tolist()

    # ---- instrumentation
    def set_stage_timing(self, on):
        self._ck(lib().aero_set_stage_timing(self.h, C.c_int32(1 if on else 0)))

    def last_stage_ms(self):
        out = np.zeros(12, np.float64)
        self._ck(lib().aero_last_stage_ms(self.h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return dict(zip(STAGE_NAMES, out.tolist()))

    def set_kernel_timing(self, on, only_kernel=None):
        self._ck(lib().aero_set_kernel_timing(self.h, C.c_int32(1 if on else 0), only_kernel.encode() if only_kernel else None))

    def kernel_timing_report(self):
        """{kernel_name: (calls, total_ms, algorithmic_bytes)} measured with HIP events on the launch stream; resets
        the counters."""
        buf = C.create_string_buffer(1 << 16)
        self._ck(lib().aero_kernel_timing_report(self.h, buf, C.c_size_t(1 << 16)))
        out = {}
        for line in buf.value.decode().strip().split("\n"):
            if line:
                name, calls, ms, nbytes = line.split()
                out[name] = (int(calls), float(ms), float(nbytes))
        return out

    def memory_stats(self):
        a = C.c_uint64(0)
        b = C.c_uint64(0)
        lib().aero_memory_stats(self.h, C.byref(a), C.byref(b))
        return a.value, b.value


class _BorrowedContext(Context):
    """A pool slot's context: owned by the pool (never destroyed from Python)."""

    def __init__(self, handle):
        self.h = handle

    def close(self):
        self.h = None


class Pool:
    """`slots` contexts on one device, each with its own worker thread inside the library (aero_pool_*): several independent
    proofs in flight per GPU through one call."""

    def __init__(self, device=0, slots=8):
        self.h = C.c_void_p()
        rc = lib().aero_pool_create(C.c_int32(device), C.c_uint32(slots), C.byref(self.h))
        if rc != 0:
            raise AeroError(rc, lib().aero_last_error(None).decode())
        lib().aero_pool_ctx.restype = C.c_void_p
        self.slots = slots
        self.ctxs = [_BorrowedContext(C.c_void_p(lib().aero_pool_ctx(self.h, C.c_uint32(i)))) for i in range(slots)]

    def ctx(self, slot) -> Context:
        return self.ctxs[slot]

    def set_self_verify(self, mode):
        """aero_pool_set_self_verify: Context.set_self_verify for every slot (between batches)."""
        m = -1 if mode in (-1, "auto", None) else int(bool(mode))
        rc = lib().aero_pool_set_self_verify(self.h, C.c_int32(m))
        if rc != 0:
            raise AeroError(rc, "aero_pool_set_self_verify")

    def prove_fib_queue(self, host_traces, options: ProofOptions, aux=(0, 0, 2)):
        """A queue of DIFFERENT host traces of one shape (PinnedTrace or C-contiguous (width, n) uint64 arrays): trace t goes to slot
        t mod slots, every proof comes back (aero_pool_prove_fib_queue). Returns [(proof_bytes, public_inputs)] in queue order."""
        arrs = [t.array if isinstance(t, PinnedTrace) else np.ascontiguousarray(t, np.uint64) for t in host_traces]
        n = len(arrs)
        w, rows = arrs[0].shape
        assert all(a.shape == (w, rows) for a in arrs), "the traces of one queue must have one shape"
        ptrs = (u64p * n)(*[_p64(a) for a in arrs])
        proofs = (u8p * n)()
        lens = (C.c_size_t * n)()
        pubs = np.zeros(n * (w // 2), np.uint64)
        air = FibAirDesc(aux[0], aux[1], aux[2])
        rc = lib().aero_pool_prove_fib_queue(self.h, ptrs, C.c_uint32(n), C.c_uint32(w), C.c_uint32(int(rows).bit_length() - 1), C.byref(air),
                                             C.byref(options), proofs, lens, _p64(pubs))
        if rc != 0:
            msgs = [lib().aero_last_error(c.h).decode() for c in self.ctxs]
            raise AeroError(rc, "; ".join(m for m in msgs if m))
        out = []
        for i in range(n):
            out.append((C.string_at(proofs[i], lens[i]), pubs[i * (w // 2):(i + 1) * (w // 2)].tolist()))
            lib().aero_free(proofs[i])
        return out

    def prove_air_queue(self, air: "Air", host_traces, pubs, options: ProofOptions):
        """The same for a constraint program: pubs[t] = the statement of trace t (aero_pool_prove_air_queue). Returns the proofs in queue order."""
        arrs = [t.array if isinstance(t, PinnedTrace) else np.ascontiguousarray(t, np.uint64) for t in host_traces]
        n = len(arrs)
        w, rows = arrs[0].shape
        assert all(a.shape == (w, rows) for a in arrs) and len(pubs) == n
        npub = len(pubs[0])
        pb = np.array([list(p) for p in pubs], dtype=np.uint64).reshape(-1) if npub else np.zeros(1, np.uint64)
        ptrs = (u64p * n)(*[_p64(a) for a in arrs])
        proofs = (u8p * n)()
        lens = (C.c_size_t * n)()
        rc = lib().aero_pool_prove_air_queue(self.h, air.h, ptrs, C.c_uint32(n), C.c_uint32(int(rows).bit_length() - 1), _p64(pb), C.c_uint32(npub),
                                             C.byref(options), proofs, lens)
        if rc != 0:
            msgs = [lib().aero_last_error(c.h).decode() for c in self.ctxs]
            raise AeroError(rc, "; ".join(m for m in msgs if m))
        out = []
        for i in range(n):
            out.append(C.string_at(proofs[i], lens[i]))
            lib().aero_free(proofs[i])
        return out

    def placement(self):
        """(NUMA node of the pool's device or -1, number of worker threads bound to that node's CPUs) - aero_pool_placement."""
        node, pinned = C.c_int32(-1), C.c_uint32(0)
        lib().aero_pool_placement(self.h, C.byref(node), C.byref(pinned))
        return node.value, pinned.value

    def prove_fib(self, traces, options: ProofOptions, aux=(0, 0, 2), rounds=1):
        """traces[i] = Matrix resident on slot i's context. Every slot proves its trace `rounds` times back to back; returns
        [(proof_bytes, public_inputs)] of the last round."""
        n = len(traces)
        arr = (C.c_void_p * n)(*[t.h for t in traces])
        proofs = (u8p * n)()
        lens = (C.c_size_t * n)()
        widths = [t.shape[0] for t in traces]
        pubs = np.zeros(sum(w // 2 for w in widths), np.uint64)
        air = FibAirDesc(aux[0], aux[1], aux[2])
        rc = lib().aero_pool_prove_fib(self.h, arr, C.c_uint32(n), C.byref(air), C.byref(options), C.c_uint32(rounds), proofs, lens, _p64(pubs))
        if rc != 0:
            msgs = [lib().aero_last_error(c.h).decode() for c in self.ctxs[:n]]
            for i in range(n):
                if proofs[i]:
                    lib().aero_free(proofs[i])
            raise AeroError(rc, "; ".join(m for m in msgs if m))
        out, off = [], 0
        for i in range(n):
            data = C.string_at(proofs[i], lens[i])
            lib().aero_free(proofs[i])
            out.append((data, pubs[off:off + widths[i] // 2].tolist()))
            off += widths[i] // 2
        return out

    def prove_fib_host(self, host_traces, options: ProofOptions, aux=(0, 0, 2), rounds=1):
        """host_traces[i] = PinnedTrace (or a C-contiguous (width, n) uint64 ndarray) in HOST memory, all of one shape. Every slot
        copies its trace to the device and proves it, `rounds` times back to back (aero_pool_prove_fib_host); returns
        [(proof_bytes, public_inputs)] of the last round."""
        arrs = [t.array if isinstance(t, PinnedTrace) else np.ascontiguousarray(t, np.uint64) for t in host_traces]
        n = len(arrs)
        w, rows = arrs[0].shape
        assert all(a.shape == (w, rows) for a in arrs), "host traces of one batch must have one shape"
        ptrs = (u64p * n)(*[_p64(a) for a in arrs])
        proofs = (u8p * n)()
        lens = (C.c_size_t * n)()
        pubs = np.zeros(n * (w // 2), np.uint64)
        air = FibAirDesc(aux[0], aux[1], aux[2])
        rc = lib().aero_pool_prove_fib_host(self.h, ptrs, C.c_uint32(w), C.c_uint32(int(rows).bit_length() - 1), C.c_uint32(n), C.byref(air),
                                            C.byref(options), C.c_uint32(rounds), proofs, lens, _p64(pubs))
        if rc != 0:
            msgs = [lib().aero_last_error(c.h).decode() for c in self.ctxs[:n]]
            for i in range(n):
                if proofs[i]:
                    lib().aero_free(proofs[i])
            raise AeroError(rc, "; ".join(m for m in msgs if m))
        out = []
        for i in range(n):
            data = C.string_at(proofs[i], lens[i])
            lib().aero_free(proofs[i])
            out.append((data, pubs[i * (w // 2):(i + 1) * (w // 2)].tolist()))
        return out

    def prove_air(self, air: "Air", traces, pub, options: ProofOptions, rounds=1):
        """The same for an AIR given as a constraint program: traces[i] = Matrix resident on slot i's context, or (all of them) host
        arrays / PinnedTrace of one shape; one program and one statement for the whole batch. Returns the proof bytes per slot."""
        n = len(traces)
        pb = np.array(pub, dtype=np.uint64, ndmin=1) if len(pub) else np.zeros(1, np.uint64)
        proofs = (u8p * n)()
        lens = (C.c_size_t * n)()
        if isinstance(traces[0], Matrix):
            arr = (C.c_void_p * n)(*[t.h for t in traces])
            rc = lib().aero_pool_prove_air(self.h, air.h, arr, C.c_uint32(n), _p64(pb), C.c_uint32(len(pub)), C.byref(options), C.c_uint32(rounds), proofs, lens)
        else:
            arrs = [t.array if isinstance(t, PinnedTrace) else np.ascontiguousarray(t, np.uint64) for t in traces]
            assert all(a.shape == arrs[0].shape for a in arrs), "host traces of one batch must have one shape"
            ptrs = (u64p * n)(*[_p64(a) for a in arrs])
            rc = lib().aero_pool_prove_air_host(self.h, air.h, ptrs, C.c_uint32(int(arrs[0].shape[1]).bit_length() - 1), C.c_uint32(n), _p64(pb),
                                                C.c_uint32(len(pub)), C.byref(options), C.c_uint32(rounds), proofs, lens)
        if rc != 0:
            msgs = [lib().aero_last_error(c.h).decode() for c in self.ctxs[:n]]
            for i in range(n):
                if proofs[i]:
                    lib().aero_free(proofs[i])
            raise AeroError(rc, "; ".join(m for m in msgs if m))
        out = []
        for i in range(n):
            out.append(C.string_at(proofs[i], lens[i]))
            lib().aero_free(proofs[i])
        return out

    def close(self):
        if self.h:
            lib().aero_pool_destroy(self.h)
            self.h = None
            for c in self.ctxs:
                c.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Air:
    """A parsed, validated and compiled AEROAIR program (aero_air_load; format: include/aero_air.h; builder: aero_amd.air)."""
    INFO = ["main_width", "aux_width", "aux_rands", "num_pub", "num_exemptions", "main_transition", "aux_transition", "main_assertions",
            "aux_assertions", "ce_blowup", "num_periodic", "num_nodes", "instructions", "registers_base", "registers_ext", "has_aux_builders"]

    def __init__(self, program: bytes):
        self.program = bytes(program)
        buf = np.frombuffer(self.program, np.uint8)
        self.h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = lib().aero_air_load(_p8(buf), C.c_size_t(len(self.program)), C.byref(self.h), err, C.c_size_t(512))
        if rc != 0:
            raise AeroError(rc, err.value.decode(errors="replace"))

    def info(self):
        out = (C.c_uint32 * 16)()
        lib().aero_air_info(self.h, out)
        return dict(zip(self.INFO, list(out)))

    def jit_compile(self, log_n, field_extension=1, fused=True):
        """Build the run-time compiled evaluation kernel ahead of the first proof (aero_air_jit_compile; no GPU needed)."""
        rc = lib().aero_air_jit_compile(self.h, C.c_uint32(log_n), C.c_uint32(field_extension), C.c_int32(1 if fused else 0))
        if rc != 0:
            raise AeroError(rc, lib().aero_last_error(None).decode())

    def prepare(self, log_n, options: "ProofOptions", world=1):
        """Build exactly the kernel a proof of 2^log_n rows under `options` will ask for, ahead of that proof (aero_air_prepare):
        the cold start (hiprtc) moves off the first proof's clock; pools and sharded entry points do this themselves."""
        rc = lib().aero_air_prepare(self.h, C.c_uint32(log_n), C.byref(options), C.c_uint32(world))
        if rc != 0:
            raise AeroError(rc, lib().aero_last_error(None).decode())

    def jit_source(self, log_n, field_extension=1, fused=True) -> str:
        out, n = u8p(), C.c_size_t(0)
        rc = lib().aero_air_jit_source(self.h, C.c_uint32(log_n), C.c_uint32(field_extension), C.c_int32(1 if fused else 0), C.byref(out), C.byref(n))
        if rc != 0:
            raise AeroError(rc, lib().aero_last_error(None).decode())
        data = C.string_at(out, n.value)
        lib().aero_free(out)
        return data.decode()

    def num_divisors(self, log_n):
        v = C.c_uint32(0)
        rc = lib().aero_air_num_divisors(self.h, C.c_uint32(log_n), C.byref(v))
        if rc != 0:
            raise AeroError(rc, lib().aero_last_error(None).decode())
        return v.value

    def free(self):
        if getattr(self, "h", None):
            lib().aero_air_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def fib_program(width, aux=(0, 0, 2)) -> bytes:
    """The built-in FibAir(width) [+ auxiliary segment (aux_width, aux_rands, aux_degree)] as an AEROAIR program (aero_air_fib_program)."""
    out, n = u8p(), C.c_size_t(0)
    desc = FibAirDesc(*aux) if aux and aux[0] else None
    rc = lib().aero_air_fib_program(C.c_uint32(width), C.byref(desc) if desc is not None else None, C.byref(out), C.byref(n))
    if rc != 0:
        raise AeroError(rc, lib().aero_last_error(None).decode())
    data = C.string_at(out, n.value)
    lib().aero_free(out)
    return data


def prove_fib_sharded_local(trace, options: ProofOptions, world, aux=(0, 0, 2), devices=None, min_peer_digests=0):
    """aero_prove_fib_sharded_local: ONE proof by `world` ranks of this process (a context and a thread per rank inside the library,
    exchanges through the in-process communicator), trace in host memory. Returns (proofs [bytes per rank], public_inputs, rank_ms,
    bytes_sent per rank)."""
    t = trace.array if isinstance(trace, PinnedTrace) else np.ascontiguousarray(trace, np.uint64)
    w, n = t.shape
    dev = (C.c_int32 * world)(*(devices or [0] * world))
    proofs, lens = (u8p * world)(), (C.c_size_t * world)()
    pub = np.zeros(w // 2 + 1, np.uint64)
    ms, sent = (C.c_double * world)(), (C.c_uint64 * world)()
    err = C.create_string_buffer(512)
    air = FibAirDesc(*aux)
    rc = lib().aero_prove_fib_sharded_local(dev, C.c_uint32(world), _p64(t), C.c_uint32(w), C.c_uint32(int(n).bit_length() - 1), C.byref(air),
                                            C.byref(options), C.c_uint32(min_peer_digests), proofs, lens, _p64(pub), ms, sent, err, C.c_size_t(512))
    if rc != 0:
        raise AeroError(rc, err.value.decode(errors="replace"))
    out = []
    for r in range(world):
        out.append(C.string_at(proofs[r], lens[r]))
        lib().aero_free(proofs[r])
    return out, pub[:w // 2].tolist(), list(ms), list(sent)


def synth_vm_program(log_n, pairs, aux=0, rands=4) -> bytes:
    """The VM-shaped synthetic AIR as an AEROAIR program (aero_air_synth_vm_program); pairs = 26, aux = 9, rands = 16: Miden's shape."""
    out, n = u8p(), C.c_size_t(0)
    rc = lib().aero_air_synth_vm_program(C.c_uint32(log_n), C.c_uint32(pairs), C.c_uint32(aux), C.c_uint32(rands), C.byref(out), C.byref(n))
    if rc != 0:
        raise AeroError(rc, lib().aero_last_error(None).decode())
    data = C.string_at(out, n.value)
    lib().aero_free(out)
    return data


def synth_vm_trace(log_n, pairs):
    """A trace that satisfies synth_vm_program: ((20 + 2 pairs, 2^log_n) uint64, public inputs)."""
    t = np.zeros((20 + 2 * pairs, 1 << log_n), np.uint64)
    pub = np.zeros(pairs + 1, np.uint64)
    rc = lib().aero_air_synth_vm_trace(C.c_uint32(log_n), C.c_uint32(pairs), _p64(t), _p64(pub))
    if rc != 0:
        raise AeroError(rc, "synth_vm_trace: bad shape")
    return t, pub.tolist()


class VerifyPolicy(C.Structure):
    """aero_verify_policy (include/aero_stark.h)."""
    _fields_ = [("min_query_security_bits", C.c_uint32), ("expected_log_n", C.c_uint32), ("allow_unknown_air", C.c_uint32),
                ("cairo_compat", C.c_uint32), ("require_options", C.c_uint32), ("options", ProofOptions),
                ("min_conjectured_security_bits", C.c_uint32)]


def _policy(min_query_security_bits, expected_log_n, allow_unknown_air, cairo_compat, require_options, min_conjectured_security_bits):
    return VerifyPolicy(96 if min_query_security_bits is None else min_query_security_bits, expected_log_n, 1 if allow_unknown_air else 0,
                        1 if cairo_compat else 0, 1 if require_options is not None else 0,
                        require_options if require_options is not None else ProofOptions(), min_conjectured_security_bits)


def verify_fib(proof: bytes, pub_elements, air, min_query_security_bits=None, expected_log_n=0, allow_unknown_air=False, cairo_compat=False,
               require_options=None, min_conjectured_security_bits=0):
    """aero_verify_fib (host only, no GPU): raises AeroError(-7, reason) when the proof is rejected.
    air = (aux_width, aux_rands, aux_degree) of the built-in FibAir - mandatory; None is accepted only together with
    allow_unknown_air=True (everything except the OOD constraint check, what the reference's Cairo verifier does).
    min_query_security_bits=None leaves the library default (96) in force."""
    buf = np.frombuffer(proof, np.uint8)
    pub = np.array(pub_elements, dtype=np.uint64, ndmin=1)
    err = C.create_string_buffer(512)
    desc = FibAirDesc(*air) if air is not None else None
    pol = _policy(min_query_security_bits, expected_log_n, allow_unknown_air, cairo_compat, require_options, min_conjectured_security_bits)
    rc = lib().aero_verify_fib(_p8(buf), C.c_size_t(len(proof)), _p64(pub) if pub.size else None, C.c_uint32(pub.size),
                               C.byref(desc) if desc is not None else None, C.byref(pol), err, C.c_size_t(512))
    if rc != 0:
        raise AeroError(rc, err.value.decode(errors="replace"))


def verify_air(proof: bytes, pub, air: Air, min_query_security_bits=None, expected_log_n=0, require_options=None, min_conjectured_security_bits=0):
    """aero_verify_air (host only): aero_verify_fib with the out-of-domain constraint check evaluated from the program."""
    buf = np.frombuffer(proof, np.uint8)
    pb = np.array(pub, dtype=np.uint64, ndmin=1) if len(pub) else np.zeros(1, np.uint64)
    err = C.create_string_buffer(512)
    pol = _policy(min_query_security_bits, expected_log_n, False, False, require_options, min_conjectured_security_bits)
    rc = lib().aero_verify_air(_p8(buf), C.c_size_t(len(proof)), _p64(pb), C.c_uint32(len(pub)), air.h, C.byref(pol), err, C.c_size_t(512))
    if rc != 0:
        raise AeroError(rc, err.value.decode(errors="replace"))


def proof_security_bits(proof: bytes):
    """(query_bits, field_bits) of a proof's self-declared parameters (aero_proof_security_bits)."""
    buf = np.frombuffer(proof, np.uint8)
    q, f = C.c_uint32(0), C.c_uint32(0)
    rc = lib().aero_proof_security_bits(_p8(buf), C.c_size_t(len(proof)), C.byref(q), C.byref(f))
    if rc != 0:
        raise AeroError(rc, "proof_security_bits: malformed proof")
    return q.value, f.value


AIR_FIB, AIR_MIDEN_PROCESSOR, AIR_PROGRAM = 0, 1, 2     # AIR_PROGRAM: the constraint set travels as an AEROAIR program next to the trace file


def trace_file_write(path, trace: np.ndarray, air=(0, 0, 2), air_id=AIR_FIB):
    """Write a trace in the AEROTRC hand-over format (include/aero_stark.h: aero_trace_file_write)."""
    t = np.ascontiguousarray(trace, np.uint64)
    w, n = t.shape
    rc = lib().aero_trace_file_write(os.fsencode(path), _p64(t), C.c_uint32(w), C.c_uint32(int(n).bit_length() - 1), C.c_uint32(air_id),
                                     C.byref(FibAirDesc(*air)))
    if rc != 0:
        raise AeroError(rc, lib().aero_last_error(None).decode())


def trace_file_info(path):
    """(width, log_n, air_id, (aux_width, aux_rands, aux_degree)) of an AEROTRC file."""
    w, ln, aid, air = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), FibAirDesc()
    rc = lib().aero_trace_file_info(os.fsencode(path), C.byref(w), C.byref(ln), C.byref(aid), C.byref(air))
    if rc != 0:
        raise AeroError(rc, lib().aero_last_error(None).decode())
    return w.value, ln.value, aid.value, (air.aux_width, air.aux_rands, air.aux_degree)


CAIRO_COMMANDS = {"proof": 0, "public-inputs": 1, "trace-queries": 2, "constraint-queries": 3, "fri-queries": 4}


def cairo_memory(command: str, proof: bytes = b"", input_bytes: bytes = b"", indexes=()) -> str:
    """The JSON array `stark_parser <file> <command>` prints (aero_cairo_memory); `indexes` = query positions for the three
    *-queries commands."""
    pb = np.frombuffer(proof, np.uint8) if proof else None
    ib = np.frombuffer(input_bytes, np.uint8) if input_bytes else None
    idx = np.array(list(indexes), dtype=np.uint64)
    out, n = C.c_char_p(), C.c_size_t(0)
    err = C.create_string_buffer(512)
    rc = lib().aero_cairo_memory(C.c_uint32(CAIRO_COMMANDS[command]), _p8(pb) if pb is not None else None, C.c_size_t(len(proof)),
                                 _p8(ib) if ib is not None else None, C.c_size_t(len(input_bytes)), _p64(idx) if idx.size else None,
                                 C.c_uint32(idx.size), C.byref(out), C.byref(n), err, C.c_size_t(512))
    if rc != 0:
        raise AeroError(rc, err.value.decode(errors="replace"))
    text = C.string_at(out, n.value).decode()
    lib().aero_free(out)
    return text


def _pb_call(fn, data: bytes) -> bytes:
    buf = np.frombuffer(data, np.uint8)
    out, n = u8p(), C.c_size_t(0)
    err = C.create_string_buffer(512)
    rc = fn(_p8(buf), C.c_size_t(len(data)), C.byref(out), C.byref(n), err, C.c_size_t(512))
    if rc != 0:
        raise AeroError(rc, err.value.decode(errors="replace"))
    res = C.string_at(out, n.value)
    lib().aero_free(out)
    return res


def proof_to_protobuf(proof: bytes) -> bytes:
    """sdk.StarkProof bytes (aero_proof_to_protobuf)."""
    return _pb_call(lib().aero_proof_to_protobuf, proof)


def miden_public_inputs_to_protobuf(input_bytes: bytes) -> bytes:
    """sdk.MidenPublicInputs bytes (aero_miden_public_inputs_to_protobuf)."""
    return _pb_call(lib().aero_miden_public_inputs_to_protobuf, input_bytes)


def _pb_call2(fn, proof: bytes, input_bytes: bytes) -> bytes:
    pb, ib = np.frombuffer(proof, np.uint8), np.frombuffer(input_bytes, np.uint8)
    out, n = u8p(), C.c_size_t(0)
    err = C.create_string_buffer(512)
    rc = fn(_p8(pb), C.c_size_t(len(proof)), _p8(ib), C.c_size_t(len(input_bytes)), C.byref(out), C.byref(n), err, C.c_size_t(512))
    if rc != 0:
        raise AeroError(rc, err.value.decode(errors="replace"))
    res = C.string_at(out, n.value)
    lib().aero_free(out)
    return res


def worker_message_info(kind: str, msg: bytes):
    """Shape of a worker message, validated on the host (aero_worker_message_info): kind = "hashing" | "constraints"."""
    buf = np.frombuffer(msg, np.uint8)
    out = (C.c_uint64 * 8)()
    err = C.create_string_buffer(512)
    rc = lib().aero_worker_message_info(C.c_uint32({"hashing": 0, "constraints": 1}[kind]), _p8(buf), C.c_size_t(len(msg)), out, err, C.c_size_t(512))
    if rc != 0:
        raise AeroError(rc, err.value.decode(errors="replace"))
    v = list(out)
    if kind == "hashing":
        return dict(rows=v[0], batch_idx=v[1], min_width=v[2], max_width=v[3], elements=v[4])
    return dict(main_width=v[0], aux_width=v[1], aux_rands=v[2], trace_len=v[3], blowup=v[4], fragment_offset=v[5], num_fragments=v[6], coefficient_pairs=v[7])


def prover_output(proof: bytes, input_bytes: bytes) -> bytes:
    """bincode ProverOutput { proof, program_outputs, public_inputs } with the three protobuf payloads (aero_prover_output)."""
    return _pb_call2(lib().aero_prover_output, proof, input_bytes)


def device_count():
    return lib().aero_device_count()
