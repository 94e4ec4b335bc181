// The messages of the reference's worker seam (aero-sdk/miden-wasm/src/utils.rs:302-450), host side only.
//
// The browser SDK posts bincode 1.3 messages (`to_uint8array` / `from_uint8array`, utils.rs:442-450: default options = little
// endian, fixed-width integers, u64 sequence lengths, usize as u64) between its proving worker and the hashing / constraint
// workers. This header restates those byte layouts so that the library can take the very bytes the reference's pool posts
// (pool.rs:98,119) and answer with the bytes its proving worker expects back:
//
//   HashingWorkItem          { data: Vec<Vec<Felt>>, batch_idx: usize }                               utils.rs:358-362
//   HashingResult            { batch_idx: usize, hashes: Vec<[u8; 32]> }                              utils.rs:411-415
//   ConstraintComputeWorkItem{ trace_info, public_inputs, proof_options, aux_rand_elements,
//                              constraint_coeffs, trace_lde_wrapper: bytes, computation_fragment }    utils.rs:302-347
//   TraceLdeWrapper          { trace_lde: (main columns, aux segments' columns, blowup) }             utils.rs:262-299
//   ConstraintComputeResult  { frag_index, frag_num, constraint_evaluations: Vec<Vec<Felt>> }         utils.rs:417-422
//   ProverOutput             { proof, program_outputs, public_inputs: Vec<u8> (protobuf) }            utils.rs:424-430
//
// A Felt travels as its canonical u64 (`FeltWrapper`, utils.rs:364-395); the reader reduces a value >= p the way `Felt::new`
// does. winter types travel as `seq(1)` of their own byte form (`winter_serde!`, utils.rs:60-105): TraceLayout = main width,
// aux width, aux rands (one byte each, one aux segment), ProofOptions = the 7 option bytes of the proof context, PublicInputs =
// Miden's layout (program hash, stack inputs, outputs.stack, overflow addresses).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "gl_field.hpp"
#include "prover.hpp"

namespace aero {
namespace wm {

[[noreturn]] inline void bad(const std::string& what) { throw Error(ST_BAD_ARG, "worker message: " + what); }

struct Rd {
    const uint8_t* p;
    size_t len, off = 0;
    uint64_t u64() {
        if (len - off < 8) bad("truncated");
        uint64_t v;
        memcpy(&v, p + off, 8);
        off += 8;
        return v;
    }
    // a sequence length whose elements take at least `min_elem_bytes` each: bounded by what is left of the message
    size_t count(size_t min_elem_bytes) {
        const uint64_t n = u64();
        if (min_elem_bytes && n > (len - off) / min_elem_bytes) bad("sequence length exceeds the message");
        return (size_t)n;
    }
    const uint8_t* bytes(size_t n) {
        if (len - off < n) bad("truncated");
        const uint8_t* q = p + off;
        off += n;
        return q;
    }
    uint64_t felt() { const uint64_t v = u64(); return v >= gl::P ? v - gl::P : v; }      // Felt::new reduces
    std::vector<uint64_t> felts() {
        const size_t n = count(8);
        std::vector<uint64_t> v(n);
        for (size_t i = 0; i < n; i++) v[i] = felt();
        return v;
    }
    std::vector<uint8_t> byte_vec() { const size_t n = count(1); const uint8_t* q = bytes(n); return std::vector<uint8_t>(q, q + n); }
    void expect_seq(uint64_t n, const char* what) { if (u64() != n) bad(std::string(what) + ": unexpected sequence length"); }
    void end() const { if (off != len) bad("trailing bytes"); }
};

struct Wr {
    std::vector<uint8_t> b;
    void u64(uint64_t v) { const size_t o = b.size(); b.resize(o + 8); memcpy(b.data() + o, &v, 8); }
    void raw(const void* q, size_t n) { const size_t o = b.size(); b.resize(o + n); if (n) memcpy(b.data() + o, q, n); }
    void byte_vec(const void* q, size_t n) { u64(n); raw(q, n); }
};

// A HashingWorkItem without copying its rows: offsets[i] = position (in 8-byte words from the start of the message) of row i's
// length field; every field of this message is a u64, so the rows can be hashed where they lie once the bytes are on the device.
inline uint64_t scan_hashing_work_item(const uint8_t* p, size_t len, std::vector<uint64_t>& offsets) {
    if (len % 8) bad("a HashingWorkItem is a sequence of 8-byte fields");
    Rd r{p, len};
    const size_t n = r.count(8);
    offsets.resize(n);
    for (size_t i = 0; i < n; i++) {
        offsets[i] = r.off / 8;
        const size_t k = r.count(8);
        if (k > 0xFFFFFFu) bad("a row of more than 2^24 elements");
        r.bytes(8 * k);
    }
    const uint64_t batch_idx = r.u64();
    r.end();
    return batch_idx;
}
inline std::vector<uint8_t> emit_hashing_result(uint64_t batch_idx, const uint8_t* digests, size_t count) {
    Wr w;
    w.u64(batch_idx);
    w.u64(count);
    w.raw(digests, 32 * count);      // [u8; 32] is a tuple for serde: no length prefix
    return std::move(w.b);
}

struct ConstraintWorkItem {
    uint32_t main_width = 0, aux_width = 0, aux_rands = 0;     // TraceLayout (one auxiliary segment)
    uint64_t trace_len = 0;
    std::vector<uint8_t> meta, public_inputs;
    uint8_t options[7] = {0};
    std::vector<std::vector<uint64_t>> aux_rand_elements;      // per auxiliary segment
    std::vector<uint64_t> coeffs;                              // (alpha, beta) per transition constraint, then per assertion
    size_t n_transition = 0, n_boundary = 0;
    struct Col { const uint8_t* data; size_t n; };            // a column where it lies in the message (n u64, possibly >= p)
    std::vector<Col> main_cols;                                // the trace LDE, column-major
    std::vector<std::vector<Col>> aux_segments;
    uint64_t blowup = 0, fragment_offset = 0, num_fragments = 0;
};
inline void parse_trace_lde(const uint8_t* p, size_t len, ConstraintWorkItem& w) {
    Rd r{p, len};
    r.expect_seq(3, "trace_lde");
    auto col = [&]() { const size_t n = r.count(8); return ConstraintWorkItem::Col{r.bytes(8 * n), n}; };
    const size_t nc = r.count(8);
    for (size_t c = 0; c < nc; c++) w.main_cols.push_back(col());
    const size_t ns = r.count(8);
    for (size_t s = 0; s < ns; s++) {
        const size_t na = r.count(8);
        std::vector<ConstraintWorkItem::Col> seg;
        for (size_t c = 0; c < na; c++) seg.push_back(col());
        w.aux_segments.push_back(std::move(seg));
    }
    w.blowup = r.u64();
    r.end();
}
inline ConstraintWorkItem parse_constraint_work_item(const uint8_t* p, size_t len) {
    Rd r{p, len};
    ConstraintWorkItem w;
    r.expect_seq(3, "trace_info");                             // serialize_trace_info: layout bytes, length, meta
    {
        const std::vector<uint8_t> layout = r.byte_vec();
        if (layout.size() != 3) bad("trace layout: expected main width, one auxiliary width, one auxiliary rand count");
        w.main_width = layout[0]; w.aux_width = layout[1]; w.aux_rands = layout[2];
    }
    w.trace_len = r.u64();
    w.meta = r.byte_vec();
    r.expect_seq(1, "public_inputs");
    w.public_inputs = r.byte_vec();
    r.expect_seq(1, "proof_options");
    {
        const std::vector<uint8_t> o = r.byte_vec();
        if (o.size() != 7) bad("proof options: expected 7 bytes");
        memcpy(w.options, o.data(), 7);
    }
    {
        const size_t segs = r.count(8);
        for (size_t s = 0; s < segs; s++) w.aux_rand_elements.push_back(r.felts());
    }
    r.expect_seq(2, "constraint_coeffs");
    w.n_transition = r.count(16);
    for (size_t i = 0; i < 2 * w.n_transition; i++) w.coeffs.push_back(r.felt());
    w.n_boundary = r.count(16);
    for (size_t i = 0; i < 2 * w.n_boundary; i++) w.coeffs.push_back(r.felt());
    {
        const size_t n = r.count(1);                           // serde_bytes
        parse_trace_lde(r.bytes(n), n, w);
    }
    w.fragment_offset = r.u64();
    w.num_fragments = r.u64();
    r.end();
    return w;
}
// columns[c] = `rows` values, c < ncols
inline std::vector<uint8_t> emit_constraint_result(uint64_t frag_index, uint64_t frag_num, const uint64_t* cols, size_t ncols, size_t rows) {
    Wr w;
    w.u64(frag_index);
    w.u64(frag_num);
    w.u64(ncols);
    for (size_t c = 0; c < ncols; c++) { w.u64(rows); w.raw(cols + c * rows, 8 * rows); }
    return std::move(w.b);
}
inline std::vector<uint8_t> emit_prover_output(const std::vector<uint8_t>& proof_pb, const std::vector<uint8_t>& outputs_pb, const std::vector<uint8_t>& inputs_pb) {
    Wr w;
    w.byte_vec(proof_pb.data(), proof_pb.size());
    w.byte_vec(outputs_pb.data(), outputs_pb.size());
    w.byte_vec(inputs_pb.data(), inputs_pb.size());
    return std::move(w.b);
}

}  // namespace wm
}  // namespace aero
