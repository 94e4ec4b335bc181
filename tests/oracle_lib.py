"""ctypes binding of oracle/liboracle.so — the CPU restatement used as the parity checker.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under aero_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.environ.get("AERO_ORACLE_PATH") or os.path.join(ORACLE_DIR, "liboracle.so")      # override: sanitizer builds (tools/build_asan.sh)

u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("capi.cpp", "stark.hpp", "prover.hpp", "air.hpp", "gl.hpp", "blake2s.hpp")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


def _p8(a):
    return a.ctypes.data_as(u8p)


def _p64(a):
    return a.ctypes.data_as(u64p)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.orc_last_error.restype = C.c_char_p
        for name in ("orc_gl_add", "orc_gl_sub", "orc_gl_mul", "orc_gl_mul_slow", "orc_gl_pow"):
            getattr(L, name).restype = C.c_uint64
            getattr(L, name).argtypes = [C.c_uint64, C.c_uint64]
        L.orc_gl_inv.restype = C.c_uint64
        L.orc_gl_inv.argtypes = [C.c_uint64]
        L.orc_gl_root_of_unity.restype = C.c_uint64
        L.orc_gl_root_of_unity.argtypes = [C.c_int]
        L.orc_leading_zeros.restype = C.c_uint32
        L.orc_leading_zeros.argtypes = [u8p, C.c_uint64]
        L.orc_artifact.restype = C.c_long
        L.orc_max_threads.restype = C.c_int

    def err(self):
        return self.lib.orc_last_error().decode()

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError("oracle: " + self.err())

    # ---- field
    def add(self, a, b): return self.lib.orc_gl_add(a, b)
    def sub(self, a, b): return self.lib.orc_gl_sub(a, b)
    def mul(self, a, b): return self.lib.orc_gl_mul(a, b)
    def mul_slow(self, a, b): return self.lib.orc_gl_mul_slow(a, b)
    def inv(self, a): return self.lib.orc_gl_inv(a)
    def pow(self, a, e): return self.lib.orc_gl_pow(a, e)
    def root_of_unity(self, log_n): return self.lib.orc_gl_root_of_unity(log_n)

    def e2_mul(self, a, b):
        A = (C.c_uint64 * 2)(*a); B = (C.c_uint64 * 2)(*b); O = (C.c_uint64 * 2)()
        self.lib.orc_e2_mul(A, B, O)
        return (O[0], O[1])

    def e2_inv(self, a):
        A = (C.c_uint64 * 2)(*a); O = (C.c_uint64 * 2)()
        self.lib.orc_e2_inv(A, O)
        return (O[0], O[1])

    def set_threads(self, n): self.lib.orc_set_threads(int(n))
    def max_threads(self): return self.lib.orc_max_threads()

    # ---- hashing
    def blake2s(self, data: bytes) -> bytes:
        out = np.zeros(32, np.uint8)
        buf = np.frombuffer(data, np.uint8) if data else np.zeros(1, np.uint8)
        self.lib.orc_blake2s(_p8(buf), C.c_size_t(len(data)), _p8(out))
        return out.tobytes()

    def hash_elements(self, elems) -> bytes:
        e = np.ascontiguousarray(elems, np.uint64)
        out = np.zeros(32, np.uint8)
        self.lib.orc_hash_elements(_p64(e if e.size else np.zeros(1, np.uint64)), C.c_size_t(e.size), _p8(out))
        return out.tobytes()

    def hash_rows(self, cols: np.ndarray) -> np.ndarray:
        """cols: (W, rows) uint64 column-major matrix -> (rows, 32) uint8"""
        cols = np.ascontiguousarray(cols, np.uint64)
        W, rows = cols.shape
        out = np.zeros((rows, 32), np.uint8)
        self.lib.orc_hash_rows(_p64(cols), C.c_uint32(W), C.c_size_t(rows), _p8(out))
        return out

    def merkle_nodes(self, leaves: np.ndarray) -> np.ndarray:
        leaves = np.ascontiguousarray(leaves, np.uint8)
        n = leaves.shape[0]
        out = np.zeros((2 * n, 32), np.uint8)
        self._ck(self.lib.orc_merkle_nodes(_p8(leaves), C.c_size_t(n), _p8(out)))
        return out

    def batch_proof(self, leaves: np.ndarray, positions) -> bytes:
        leaves = np.ascontiguousarray(leaves, np.uint8)
        pos = np.ascontiguousarray(positions, np.uint64)
        cap = 1 + len(pos) * (1 + 32 * 40)
        out = np.zeros(cap, np.uint8)
        ol = C.c_size_t(0)
        self._ck(self.lib.orc_batch_proof(_p8(leaves), C.c_size_t(leaves.shape[0]), _p64(pos), C.c_size_t(len(pos)),
                                          _p8(out), C.c_size_t(cap), C.byref(ol)))
        return out[:ol.value].tobytes()

    # ---- coin
    def coin_new(self, pub) -> bytes:
        p = np.ascontiguousarray(pub, np.uint64)
        seed = np.zeros(32, np.uint8)
        self.lib.orc_coin_new(_p64(p), C.c_size_t(p.size), _p8(seed))
        return seed.tobytes()

    def coin_reseed(self, seed: bytes, d: bytes) -> bytes:
        s = np.frombuffer(seed, np.uint8).copy(); dd = np.frombuffer(d, np.uint8).copy()
        self.lib.orc_coin_reseed(_p8(s), _p8(dd))
        return s.tobytes()

    def coin_reseed_int(self, seed: bytes, v: int) -> bytes:
        s = np.frombuffer(seed, np.uint8).copy()
        self.lib.orc_coin_reseed_int(_p8(s), C.c_uint64(v))
        return s.tobytes()

    def coin_draw(self, seed: bytes, ctr: int):
        s = np.frombuffer(seed, np.uint8).copy()
        c = C.c_uint64(ctr); o = C.c_uint64(0)
        self._ck(self.lib.orc_coin_draw(_p8(s), C.byref(c), C.byref(o)))
        return o.value, c.value

    def coin_draw_integers(self, seed: bytes, ctr: int, k: int, domain: int):
        s = np.frombuffer(seed, np.uint8).copy()
        c = C.c_uint64(ctr); out = np.zeros(k, np.uint64)
        self._ck(self.lib.orc_coin_draw_integers(_p8(s), C.byref(c), C.c_size_t(k), C.c_uint64(domain), _p64(out)))
        return out.tolist(), c.value

    def leading_zeros(self, seed: bytes, nonce: int) -> int:
        s = np.frombuffer(seed, np.uint8).copy()
        return self.lib.orc_leading_zeros(_p8(s), C.c_uint64(nonce))

    # ---- polynomial stages
    def intt(self, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, np.uint64).copy()
        self.lib.orc_intt(_p64(a), C.c_size_t(a.size))
        return a

    def lde(self, coeffs: np.ndarray, blowup: int) -> np.ndarray:
        c = np.ascontiguousarray(coeffs, np.uint64)
        out = np.zeros(c.size * blowup, np.uint64)
        self.lib.orc_lde(_p64(c), C.c_size_t(c.size), C.c_size_t(blowup), _p64(out))
        return out

    def fib_trace(self, W: int, log_n: int) -> np.ndarray:
        out = np.zeros((W, 1 << log_n), np.uint64)
        self.lib.orc_fib_trace(C.c_uint32(W), C.c_int(log_n), _p64(out))
        return out

    def fri_fold(self, values: np.ndarray, fold: int, alpha: int) -> np.ndarray:
        v = np.ascontiguousarray(values, np.uint64)
        out = np.zeros(v.size // fold, np.uint64)
        self.lib.orc_fri_fold(_p64(v), C.c_size_t(v.size), C.c_uint32(fold), C.c_uint64(alpha), _p64(out))
        return out

    # ---- containers / proofs
    def container_split(self, data: bytes):
        buf = np.frombuffer(data, np.uint8)
        a, b, c, d = (C.c_size_t(0) for _ in range(4))
        self._ck(self.lib.orc_container_split(_p8(buf), C.c_size_t(len(data)), C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return data[a.value:a.value + b.value], data[c.value:c.value + d.value]

    def miden_pub_elements(self, inputs: bytes):
        buf = np.frombuffer(inputs, np.uint8)
        out = np.zeros(len(inputs) // 8 + 8, np.uint64); n = C.c_size_t(0)
        self._ck(self.lib.orc_miden_pub_elements(_p8(buf), C.c_size_t(len(inputs)), _p64(out), C.c_size_t(out.size), C.byref(n)))
        return out[:n.value].tolist()

    def proof_roundtrip(self, proof: bytes) -> bytes:
        buf = np.frombuffer(proof, np.uint8)
        out = np.zeros(len(proof) + 64, np.uint8); n = C.c_size_t(0)
        self._ck(self.lib.orc_proof_roundtrip(_p8(buf), C.c_size_t(len(proof)), _p8(out), C.c_size_t(out.size), C.byref(n)))
        return out[:n.value].tobytes()

    def verify(self, proof: bytes, pub, air_kind=1, W=0, log_n=0, want_info=False):
        """Raises RuntimeError when the proof is rejected. Returns the transcript dict if want_info."""
        buf = np.frombuffer(proof, np.uint8)
        p = np.ascontiguousarray(pub, np.uint64)
        info = C.create_string_buffer(1 << 16) if want_info else None
        rc = self.lib.orc_verify(_p8(buf), C.c_size_t(len(proof)), _p64(p), C.c_size_t(p.size), C.c_int(air_kind),
                                 C.c_uint32(W), C.c_int(log_n), info, C.c_size_t(1 << 16 if want_info else 0))
        self._ck(rc)
        if want_info:
            d = {}
            for line in info.value.decode().strip().split("\n"):
                k, v = line.split("=", 1)
                d[k] = v
            return d
        return None

    def prove_fib(self, W, log_n, opt7, trace=None, keep_artifacts=False):
        """Returns (proof_bytes, pub_results(list), times(dict))."""
        o = (C.c_uint8 * 7)(*opt7)
        proof = u8p(); plen = C.c_size_t(0)
        pub = np.zeros(W // 2, np.uint64); times = np.zeros(12, np.float64)
        tr = None
        if trace is not None:
            tr = np.ascontiguousarray(trace, np.uint64)
            assert tr.shape == (W, 1 << log_n)
        self._ck(self.lib.orc_prove_fib(_p64(tr) if tr is not None else None, C.c_uint32(W), C.c_int(log_n), o,
                                        C.byref(proof), C.byref(plen), _p64(pub), times.ctypes.data_as(C.POINTER(C.c_double)),
                                        C.c_int(1 if keep_artifacts else 0)))
        data = C.string_at(proof, plen.value)
        self.lib.orc_free(proof)
        names = ["interpolate", "lde", "trace_commit", "constraints", "composition", "comp_commit", "ood", "deep", "fri",
                 "grind", "queries", "total"]
        return data, pub.tolist(), dict(zip(names, times.tolist()))

    def prove_fib_aux(self, W, log_n, A, R, opt7, trace=None, D=2, keep_artifacts=False):
        """FibAir(W) with an auxiliary segment of A columns built from R coin elements, aux constraint degree D.
        Returns (proof, pub, times)."""
        o = (C.c_uint8 * 7)(*opt7)
        proof = u8p(); plen = C.c_size_t(0)
        pub = np.zeros(W // 2, np.uint64); times = np.zeros(12, np.float64)
        tr = None
        if trace is not None:
            tr = np.ascontiguousarray(trace, np.uint64)
            assert tr.shape == (W, 1 << log_n)
        self._ck(self.lib.orc_prove_fib_aux(_p64(tr) if tr is not None else None, C.c_uint32(W), C.c_int(log_n), C.c_uint32(A),
                                            C.c_uint32(R), C.c_uint32(D), o, C.byref(proof), C.byref(plen), _p64(pub),
                                            times.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(1 if keep_artifacts else 0)))
        data = C.string_at(proof, plen.value)
        self.lib.orc_free(proof)
        names = ["interpolate", "lde", "trace_commit", "constraints", "composition", "comp_commit", "ood", "deep", "fri",
                 "grind", "queries", "total"]
        return data, pub.tolist(), dict(zip(names, times.tolist()))

    def verify_fib_aux(self, proof: bytes, pub, W, log_n, A, R, D=2):
        """Full verification incl. the OOD constraint check for FibAir(W) + aux segment (A, R); raises when rejected."""
        buf = np.frombuffer(proof, np.uint8)
        p = np.ascontiguousarray(pub, np.uint64)
        self._ck(self.lib.orc_verify_fib_aux(_p8(buf), C.c_size_t(len(proof)), _p64(p), C.c_size_t(p.size), C.c_uint32(W),
                                             C.c_int(log_n), C.c_uint32(A), C.c_uint32(R), C.c_uint32(D)))

    # ---- AIR-as-data (oracle/air.hpp)
    def prove_air(self, program: bytes, trace, pub, opt7, keep_artifacts=False):
        """Prove `trace` (W, n) against an AEROAIR program with public inputs `pub`. Returns (proof_bytes, times)."""
        tr = np.ascontiguousarray(trace, np.uint64)
        W, n = tr.shape
        pg = np.frombuffer(program, np.uint8)
        pb = np.array(pub, dtype=np.uint64, ndmin=1) if len(pub) else np.zeros(1, np.uint64)
        o = (C.c_uint8 * 7)(*opt7)
        proof = u8p(); plen = C.c_size_t(0)
        times = np.zeros(12, np.float64)
        self._ck(self.lib.orc_prove_air(_p8(pg), C.c_size_t(len(program)), _p64(tr), C.c_uint32(W), C.c_int(int(n).bit_length() - 1), _p64(pb),
                                        C.c_size_t(len(pub)), o, C.byref(proof), C.byref(plen), times.ctypes.data_as(C.POINTER(C.c_double)),
                                        C.c_int(1 if keep_artifacts else 0)))
        data = C.string_at(proof, plen.value)
        self.lib.orc_free(proof)
        names = ["interpolate", "lde", "trace_commit", "constraints", "composition", "comp_commit", "ood", "deep", "fri",
                 "grind", "queries", "total"]
        return data, dict(zip(names, times.tolist()))

    def verify_air(self, proof: bytes, program: bytes, pub, log_n):
        """Full verification incl. the OOD constraint check evaluated from the program; raises when rejected."""
        buf = np.frombuffer(proof, np.uint8)
        pg = np.frombuffer(program, np.uint8)
        pb = np.array(pub, dtype=np.uint64, ndmin=1) if len(pub) else np.zeros(1, np.uint64)
        self._ck(self.lib.orc_verify_air(_p8(buf), C.c_size_t(len(proof)), _p8(pg), C.c_size_t(len(program)), _p64(pb), C.c_size_t(len(pub)),
                                         C.c_int(log_n)))

    def air_info(self, program: bytes, log_n):
        pg = np.frombuffer(program, np.uint8)
        out = (C.c_uint64 * 8)()
        self._ck(self.lib.orc_air_info(_p8(pg), C.c_size_t(len(program)), C.c_int(log_n), out))
        keys = ["ce_blowup", "columns", "transition", "assertions", "main_width", "aux_width", "aux_rands", "nodes"]
        return dict(zip(keys, list(out)))

    def air_check_trace(self, program: bytes, trace, pub):
        tr = np.ascontiguousarray(trace, np.uint64)
        W, n = tr.shape
        pg = np.frombuffer(program, np.uint8)
        pb = np.array(pub, dtype=np.uint64, ndmin=1) if len(pub) else np.zeros(1, np.uint64)
        self._ck(self.lib.orc_air_check_trace(_p8(pg), C.c_size_t(len(program)), _p64(tr), C.c_uint32(W), C.c_int(int(n).bit_length() - 1),
                                              _p64(pb), C.c_size_t(len(pub))))

    def artifact(self, name: str, count: int) -> np.ndarray:
        out = np.zeros(count, np.uint64)
        r = self.lib.orc_artifact(name.encode(), _p64(out), C.c_size_t(count))
        if r < 0:
            raise RuntimeError(f"oracle artifact {name}: rc {r}")
        return out[:r]


_inst = None


def load():
    global _inst
    if _inst is None:
        build()
        _inst = Oracle(C.CDLL(LIB_PATH))
    return _inst
