"""bincode 1.3 messages of the reference's worker seam (aero-sdk/miden-wasm/src/utils.rs:302-450), host side.

The library takes and returns the raw bytes (`Context.worker_hash_rows`, `Context.worker_eval_constraints`, `prover_output`);
this module builds and reads those bytes for hosts and tests written in Python. Layouts (little endian, u64 lengths, usize as
u64; restated in aero_amd/csrc/worker_messages.hpp):

  HashingWorkItem           rows: u64 count, per row u64 length + u64 elements; then u64 batch_idx          utils.rs:358-362
  HashingResult             u64 batch_idx, u64 count, count x 32 bytes                                      utils.rs:411-415
  ConstraintComputeWorkItem seq(3)[layout bytes, u64 trace length, meta bytes], seq(1)[public input bytes],
                            seq(1)[7 option bytes], aux random elements per segment, seq(2)[transition pairs,
                            boundary pairs], bytes(TraceLdeWrapper), u64 fragment_offset, u64 num_fragments   utils.rs:302-347
  TraceLdeWrapper           seq(3)[main columns, aux segments' columns, u64 blowup]                          utils.rs:262-299
  ConstraintComputeResult   u64 frag_index, u64 frag_num, columns (u64 count, per column u64 length + data) utils.rs:417-422
  ProverOutput              three byte vectors: proof, program_outputs, public_inputs (protobuf)            utils.rs:424-430
"""
import struct

import numpy as np


def _u64(v):
    return struct.pack("<Q", int(v))


def _felts(values):
    a = np.ascontiguousarray(values, dtype=np.uint64)
    return _u64(a.size) + a.astype("<u8").tobytes()


def _bytes(b):
    return _u64(len(b)) + bytes(b)


class _Reader:
    def __init__(self, data):
        self.b, self.o = bytes(data), 0

    def u64(self):
        if len(self.b) - self.o < 8:
            raise ValueError("truncated message")
        (v,) = struct.unpack_from("<Q", self.b, self.o)
        self.o += 8
        return v

    def felts(self):
        n = self.u64()
        if n > (len(self.b) - self.o) // 8:
            raise ValueError("sequence length exceeds the message")
        a = np.frombuffer(self.b, dtype="<u8", count=n, offset=self.o).astype(np.uint64)
        self.o += 8 * n
        return a

    def raw(self, n):
        if len(self.b) - self.o < n:
            raise ValueError("truncated message")
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def end(self):
        if self.o != len(self.b):
            raise ValueError("trailing bytes")


def encode_hashing_work_item(rows, batch_idx=0) -> bytes:
    """rows: iterable of element sequences (they may differ in length)."""
    rows = list(rows)
    return _u64(len(rows)) + b"".join(_felts(r) for r in rows) + _u64(batch_idx)


def decode_hashing_result(data):
    """-> (batch_idx, [32-byte digests])."""
    r = _Reader(data)
    batch_idx, n = r.u64(), r.u64()
    digests = [r.raw(32) for _ in range(n)]
    r.end()
    return batch_idx, digests


def miden_public_inputs(program_hash, stack_inputs, outputs_stack, overflow_addrs=()) -> bytes:
    """Miden `PublicInputs` bytes: 4 hash elements, then three u64-counted lists."""
    out = np.ascontiguousarray(program_hash, dtype=np.uint64).astype("<u8").tobytes()
    assert len(out) == 32
    for part in (stack_inputs, outputs_stack, overflow_addrs):
        out += _felts(part)
    return out


def encode_trace_lde(main_cols, aux_segments, blowup) -> bytes:
    out = _u64(3) + _u64(len(main_cols)) + b"".join(_felts(c) for c in main_cols)
    out += _u64(len(aux_segments))
    for seg in aux_segments:
        out += _u64(len(seg)) + b"".join(_felts(c) for c in seg)
    return out + _u64(blowup)


def encode_constraint_work_item(layout, trace_len, public_inputs, options, aux_rand_elements, transition, boundary, main_cols, aux_segments,
                                blowup, fragment_offset, num_fragments, meta=b"") -> bytes:
    """layout = (main width, aux width, aux rands); options = the 7 option bytes; transition / boundary = sequences of (alpha, beta);
    aux_rand_elements = one element list per auxiliary segment; main_cols / aux_segments = the trace LDE, column-major."""
    out = _u64(3) + _bytes(bytes(layout)) + _u64(trace_len) + _bytes(meta)
    out += _u64(1) + _bytes(public_inputs)
    out += _u64(1) + _bytes(bytes(options))
    out += _u64(len(aux_rand_elements)) + b"".join(_felts(v) for v in aux_rand_elements)
    out += _u64(2)
    for pairs in (transition, boundary):
        flat = np.ascontiguousarray(pairs, dtype=np.uint64).reshape(-1)
        out += _u64(flat.size // 2) + flat.astype("<u8").tobytes()
    out += _bytes(encode_trace_lde(main_cols, aux_segments, blowup))
    return out + _u64(fragment_offset) + _u64(num_fragments)


def decode_constraint_result(data):
    """-> (frag_index, frag_num, columns as a 2-d uint64 array)."""
    r = _Reader(data)
    frag_index, frag_num, ncols = r.u64(), r.u64(), r.u64()
    cols = [r.felts() for _ in range(ncols)]
    r.end()
    return frag_index, frag_num, np.stack(cols) if cols else np.zeros((0, 0), np.uint64)


def decode_prover_output(data):
    """-> (proof, program_outputs, public_inputs) protobuf payloads."""
    r = _Reader(data)
    parts = [r.raw(r.u64()) for _ in range(3)]
    r.end()
    return tuple(parts)
