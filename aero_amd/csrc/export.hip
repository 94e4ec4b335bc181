// Re-encoders of a finished proof for the reference's two downstream consumers (host only, no GPU work):
//
//  1. Cairo-memory JSON — what `stark_parser <file> proof | public-inputs | trace-queries | constraint-queries | fri-queries`
//     prints and src/stark_verifier/utils.py:10-23 (`write_into_memory`) loads into the Cairo VM. Behaviour restated from
//     miden-to-cairo-parser/src/memory.rs:31-123 (segments, pointers resolved to offsets at assembly time, value formats) and
//     miden-to-cairo-parser/src/lib.rs:41-260,395-436 (which fields are written, in which order, for which sub-command);
//     the structs the Cairo side overlays on the image: src/stark_verifier/air/stark_proof.cairo, air/pub_inputs.cairo,
//     channel.cairo:206-244 (QueriesProof = length + digests).
//  2. protobuf `sdk.StarkProof` / `sdk.MidenPublicInputs` — the SDK's wire format: aero-sdk/proto/*.proto, field contents
//     after aero-sdk/miden-wasm/src/convert/convert_proof.rs:13-307. The encoder the reference uses (prost 0.11.8) is a
//     third-party crate that is not in the mount; what is restated here is the published proto3 wire format it implements
//     (fields in tag order, scalar zero values and empty bytes / repeated fields omitted, present sub-messages always written,
//     `repeated uint64` packed). tests/test_export_cpu.py parses the bytes with the official protobuf runtime against the
//     same schema and re-serialises them to the identical bytes.
//
// Both consumers are base-field only in the reference (`parse::<Felt>`; convert_proof.rs:140-147 `todo!()` for extensions):
// proofs over the quadratic extension are refused with AERO_E_UNSUPPORTED, not re-interpreted.
#include <cstdio>
#include <cstring>

#include "../../include/aero_stark.h"
#include "proof_format.hpp"
#include "worker_messages.hpp"

namespace aero {
namespace {

using fmt::Parsed;
using fmt::rd64;

// ---------------------------------------------------------------------------------------------------------------------------
// Cairo memory image: a list of segments; `alloc` appends a segment and leaves a pointer entry to it in the current one;
// at assembly the segments are concatenated in creation order and every pointer becomes the DECIMAL offset of its target
// segment, every value a hex string (memory.rs:45-62,126-150).
struct CairoImage {
    struct Entry { bool is_ptr; size_t target; std::string text; };
    std::vector<std::vector<Entry>> segs;
    CairoImage() { segs.emplace_back(); }

    struct Cursor {
        CairoImage* img;
        size_t seg;
        void value(uint64_t v) {   // `format!("{:#X}", value)`: 0x + upper-case hex, no padding
            char b[24];
            snprintf(b, sizeof b, "0x%llX", (unsigned long long)v);
            img->segs[seg].push_back({false, 0, b});
        }
        void felt(uint64_t v) {    // lib.rs:222-231: 0x + 16 lower-case hex digits (big-endian bytes of the canonical value)
            char b[24];
            snprintf(b, sizeof b, "0x%016llx", (unsigned long long)v);
            img->segs[seg].push_back({false, 0, b});
        }
        Cursor alloc() {
            const size_t s = img->segs.size();
            img->segs[seg].push_back({true, s, ""});
            img->segs.emplace_back();
            return Cursor{img, s};
        }
        // ByteDigest<32>: eight u32 read little-endian from the digest bytes (lib.rs:172-179)
        void digest(const Digest& d) { for (int i = 0; i < 8; i++) value(d.w[i]); }
        void felts_array(const std::vector<uint64_t>& v) { Cursor c = alloc(); for (uint64_t x : v) c.felt(x); }
        void sized_felts(const std::vector<uint64_t>& v) { value(v.size()); felts_array(v); }
        void values_array(const std::vector<uint64_t>& v) { Cursor c = alloc(); for (uint64_t x : v) c.value(x); }
        void sized_values(const std::vector<uint64_t>& v) { value(v.size()); values_array(v); }
        void sized_digests(const std::vector<Digest>& v) { value(v.size()); Cursor c = alloc(); for (auto& d : v) c.digest(d); }
    };
    Cursor root() { return Cursor{this, 0}; }

    std::string to_json() const {   // serde_json::to_string(&Vec<String>): ["a","b",...] without spaces
        std::vector<size_t> start;
        size_t total = 0;
        for (auto& s : segs) { start.push_back(total); total += s.size(); }
        std::string out = "[";
        bool first = true;
        for (auto& s : segs)
            for (auto& e : s) {
                if (!first) out += ',';
                first = false;
                out += '"';
                out += e.is_ptr ? std::to_string(start[e.target]) : e.text;
                out += '"';
            }
        out += ']';
        return out;
    }
};

std::vector<uint64_t> elems(const Bytes& b, size_t first, size_t count) {
    std::vector<uint64_t> v(count);
    for (size_t i = 0; i < count; i++) v[i] = rd64(b, first + i);
    return v;
}
int ilog2z(uint64_t x) { int r = 0; while ((1ull << r) < x) r++; return r; }

void require_base_field(const Parsed& pr, const char* who) {
    if (pr.opt.field_extension != EXT_NONE)
        throw Error(AERO_E_UNSUPPORTED, std::string(who) + ": the reference's encoder reads every element as a base-field element (extension fields are `todo!()` there)");
}
// Everything the re-encoders need from a proof, cut into the pieces the reference's `parse` helpers return.
struct Pieces {
    size_t n, N, Q, W, A, C, F;
    int layers;
    std::vector<Digest> trace_roots, fri_roots;
    Digest constraint_root;
    std::vector<uint64_t> main_cur, main_next, aux_cur, aux_next, ood_evals;
    std::vector<uint64_t> main_states, aux_states, constraint_evals;   // row-major tables, Q rows
    std::vector<std::vector<uint64_t>> fri_values;                      // per layer, flattened rows of F values
    std::vector<uint64_t> remainder;
};
Pieces cut(const Parsed& pr, const char* who) {
    require_base_field(pr, who);
    try { pr.opt.validate(); } catch (const Error& e) { fmt::bad(e.what()); }
    Pieces p;
    p.n = pr.trace_length(); p.N = pr.lde_domain_size(); p.Q = pr.opt.num_queries; p.W = pr.W; p.A = pr.A;
    p.C = pr.num_composition_columns(); p.F = pr.opt.fri_folding_factor; p.layers = pr.num_fri_layers();
    if (pr.log_n < 1 || pr.log_n > 32) fmt::bad("unsupported trace length");
    const size_t segs = pr.num_trace_segments();
    // `Commitments::parse(num_trace_segments, num_fri_layers)`: trace roots, constraint root, FRI roots incl. the remainder's
    if (pr.num_roots() != segs + 1 + (size_t)p.layers + 1) fmt::bad("wrong number of commitments");
    for (size_t i = 0; i < segs; i++) p.trace_roots.push_back(pr.root(i));
    p.constraint_root = pr.root(segs);
    for (size_t i = segs + 1; i < pr.num_roots(); i++) p.fri_roots.push_back(pr.root(i));
    // `OodFrame::parse(main_width, aux_width, num_evaluations)`: trace_states = current row (main || aux) then next row
    const size_t TW = p.W + p.A;
    if (pr.ood_trace_states.size() != 2 * TW * 8) fmt::bad("bad OOD frame length");
    if (pr.ood_evaluations.empty() || pr.ood_evaluations.size() % 8) fmt::bad("bad OOD evaluations length");
    p.main_cur = elems(pr.ood_trace_states, 0, p.W); p.aux_cur = elems(pr.ood_trace_states, p.W, p.A);
    p.main_next = elems(pr.ood_trace_states, TW, p.W); p.aux_next = elems(pr.ood_trace_states, TW + p.W, p.A);
    p.ood_evals = elems(pr.ood_evaluations, 0, p.C);
    // `TraceQueries::new` / `ConstraintQueries::new`: Table of num_queries rows
    if (pr.trace_queries.size() != segs) fmt::bad("wrong number of trace query blocks");
    if (pr.trace_queries[0].values.size() != p.Q * p.W * 8) fmt::bad("bad trace query length");
    p.main_states = elems(pr.trace_queries[0].values, 0, p.Q * p.W);
    if (p.A) {
        if (pr.trace_queries[1].values.size() != p.Q * p.A * 8) fmt::bad("bad auxiliary trace query length");
        p.aux_states = elems(pr.trace_queries[1].values, 0, p.Q * p.A);
    }
    if (pr.constraint_queries.values.size() != p.Q * p.C * 8) fmt::bad("bad constraint query length");
    p.constraint_evals = elems(pr.constraint_queries.values, 0, p.Q * p.C);
    if ((int)pr.fri_layers.size() != p.layers) fmt::bad("wrong number of FRI layers");
    for (auto& l : pr.fri_layers) {
        if (l.values.size() % (8 * p.F)) fmt::bad("bad FRI layer length");
        p.fri_values.push_back(elems(l.values, 0, l.values.size() / 8));
    }
    if (pr.fri_remainder.size() % 8) fmt::bad("bad remainder length");
    p.remainder = elems(pr.fri_remainder, 0, pr.fri_remainder.size() / 8);
    return p;
}
std::vector<Digest> row_hashes(const std::vector<uint64_t>& table, size_t rows, size_t width) {
    std::vector<Digest> out(rows);
    for (size_t i = 0; i < rows; i++) out[i] = b2s::hash_elements(table.data() + i * width, (uint32_t)width);
    return out;
}
bool same(const Digest& a, const Digest& b) { return memcmp(a.w, b.w, sizeof a.w) == 0; }

// ---- `stark_parser <file> proof` (lib.rs:65-75 and the impls it calls) ------------------------------------------------------
std::string cairo_proof(const Parsed& pr) {
    const Pieces p = cut(pr, "cairo_memory");
    if (!p.A)
        throw Error(AERO_E_UNSUPPORTED, "cairo_memory: the reference's encoder unwraps the auxiliary OOD frame and the auxiliary trace states "
                                        "(miden-to-cairo-parser/src/lib.rs:138,147): a proof without an auxiliary segment has no Cairo-memory image");
    CairoImage img;
    auto t = img.root();
    // Context (lib.rs:77-93): trace layout (:181-196), length, log2(length), meta, modulus bytes, options (:198-210), LDE domain size
    t.value(p.W);
    t.value(1);                                              // num_aux_segments
    t.values_array({(uint64_t)p.A});
    t.values_array({(uint64_t)pr.R});
    t.value(p.n);
    t.value((uint64_t)pr.log_n);
    t.value(pr.trace_meta.size());
    t.values_array(std::vector<uint64_t>(pr.trace_meta.begin(), pr.trace_meta.end()));
    t.value(8);
    {
        std::vector<uint64_t> mod(8);
        for (int i = 0; i < 8; i++) mod[i] = (gl::P >> (8 * i)) & 0xff;
        t.values_array(mod);
    }
    t.value(pr.opt.num_queries); t.value(pr.opt.blowup_factor); t.value((uint64_t)ilog2z(pr.opt.blowup_factor)); t.value(pr.opt.grinding_factor);
    t.value(pr.opt.hash_fn); t.value(pr.opt.field_extension);
    t.value(pr.opt.fri_folding_factor); t.value(1ull << pr.opt.fri_log_max_remainder);
    t.value(p.N);
    // Commitments (lib.rs:95-125)
    { auto c = t.alloc(); for (auto& d : p.trace_roots) c.digest(d); }
    { auto c = t.alloc(); c.digest(p.constraint_root); }
    t.value(p.fri_roots.size());
    { auto c = t.alloc(); for (auto& d : p.fri_roots) c.digest(d); }
    // OOD frame (lib.rs:127-143, EvaluationFrame :215-220)
    t.sized_felts(p.main_cur); t.sized_felts(p.main_next);
    t.sized_felts(p.aux_cur); t.sized_felts(p.aux_next);
    t.sized_felts(p.ood_evals);
    // pow_nonce
    t.value(pr.nonce);
    // trace queries (lib.rs:145-152), constraint queries (:154-160): Table = rows, columns, data (:162-168)
    t.value(p.Q); t.value(p.W); t.felts_array(p.main_states);
    t.value(p.Q); t.value(p.A); t.felts_array(p.aux_states);
    t.value(p.Q); t.value(p.C); t.felts_array(p.constraint_evals);
    // FRI remainder (lib.rs:73)
    t.sized_felts(p.remainder);
    return img.to_json();
}

// ---- `stark_parser <file> public-inputs` (lib.rs:41-57): Miden `PublicInputs` bytes = program hash (4 elements) || u64 count ||
// stack inputs || u64 count || outputs.stack || u64 count || overflow addresses. Program hash and stack inputs
// are `Felt`s (16-digit hex), the outputs are plain u64 (aero-sdk/miden-wasm/src/utils.rs:485-486,535).
std::string cairo_public_inputs(const uint8_t* b, size_t len) {
    const fmt::MidenInputs m = fmt::parse_miden_inputs(b, len);
    CairoImage img;
    auto t = img.root();
    t.sized_felts(m.hash);
    t.sized_felts(m.stack_inputs);
    t.sized_values(m.out_stack);
    t.sized_values(m.overflow);
    return img.to_json();
}

// ---- `trace-queries`, `constraint-queries` (lib.rs:395-419): per query proof one child segment holding, per index, the sized
// authentication path (`BatchMerkleProof::into_paths`). The positions must be the ones the proof was opened at, in that order
// (the Cairo hints pass the positions they drew); the reconstructed root is checked against the commitment.
void write_paths(CairoImage::Cursor& t, const std::vector<std::vector<Digest>>& paths) {
    auto child = t.alloc();
    for (auto& p : paths) child.sized_digests(p);
}
std::string cairo_trace_queries(const Parsed& pr, const std::vector<uint64_t>& idx) {
    const Pieces p = cut(pr, "cairo_memory");
    if (idx.size() != p.Q) fmt::bad("the number of indexes differs from the number of queries");
    CairoImage img;
    auto t = img.root();
    const size_t widths[2] = {p.W, p.A};
    const std::vector<uint64_t>* tables[2] = {&p.main_states, &p.aux_states};
    for (size_t s = 0; s < pr.num_trace_segments(); s++) {
        const auto leaves = row_hashes(*tables[s], p.Q, widths[s]);
        const auto known = fmt::batch_known_nodes(p.N, idx, leaves, pr.trace_queries[s].paths);
        if (!same(known.at(1), p.trace_roots[s])) fmt::bad("trace openings do not match the commitment at these positions");
        write_paths(t, fmt::batch_into_paths(p.N, idx, leaves, pr.trace_queries[s].paths));
    }
    return img.to_json();
}
std::string cairo_constraint_queries(const Parsed& pr, const std::vector<uint64_t>& idx) {
    const Pieces p = cut(pr, "cairo_memory");
    if (idx.size() != p.Q) fmt::bad("the number of indexes differs from the number of queries");
    CairoImage img;
    auto t = img.root();
    const auto leaves = row_hashes(p.constraint_evals, p.Q, p.C);
    if (!same(fmt::batch_root(p.N, idx, leaves, pr.constraint_queries.paths), p.constraint_root)) fmt::bad("constraint openings do not match the commitment at these positions");
    write_paths(t, fmt::batch_into_paths(p.N, idx, leaves, pr.constraint_queries.paths));
    return img.to_json();
}
// ---- `fri-queries` (lib.rs:421-436 fold_positions, :438-470): per layer one child segment holding, per folded position,
// the sized path followed by a pointer to the row's `folding_factor` values.
std::string cairo_fri_queries(const Parsed& pr, const std::vector<uint64_t>& idx) {
    const Pieces p = cut(pr, "cairo_memory");
    CairoImage img;
    auto t = img.root();
    std::vector<uint64_t> pos = idx;
    size_t dom = p.N;
    for (int l = 0; l < p.layers; l++) {
        pos = fold_positions(pos, dom, p.F);
        const size_t rows = dom / p.F;
        if (p.fri_values[l].size() != pos.size() * p.F) fmt::bad("FRI layer values do not match the folded positions");
        const auto leaves = row_hashes(p.fri_values[l], pos.size(), p.F);
        const auto paths = fmt::batch_into_paths(rows, pos, leaves, pr.fri_layers[l].paths);
        if (!same(fmt::batch_root(rows, pos, leaves, pr.fri_layers[l].paths), p.fri_roots[l])) fmt::bad("FRI layer openings do not match the commitment at these positions");
        auto child = t.alloc();
        for (size_t i = 0; i < pos.size(); i++) {
            child.sized_digests(paths[i]);
            child.felts_array(std::vector<uint64_t>(p.fri_values[l].begin() + i * p.F, p.fri_values[l].begin() + (i + 1) * p.F));
        }
        dom = rows;
    }
    return img.to_json();
}

// ---------------------------------------------------------------------------------------------------------------------------
// proto3 wire format
struct Pb {
    Bytes b;
    void varint(uint64_t v) { while (v >= 0x80) { b.push_back((uint8_t)(v | 0x80)); v >>= 7; } b.push_back((uint8_t)v); }
    void tag(uint32_t field, uint32_t wire) { varint(((uint64_t)field << 3) | wire); }
    void u64(uint32_t field, uint64_t v) { if (v) { tag(field, 0); varint(v); } }                       // scalar: zero is not written
    void bytes(uint32_t field, const uint8_t* p, size_t n) { if (n) { tag(field, 2); varint(n); b.insert(b.end(), p, p + n); } }
    void msg(uint32_t field, const Pb& m) { tag(field, 2); varint(m.b.size()); b.insert(b.end(), m.b.begin(), m.b.end()); }   // present message: always written
    void packed(uint32_t field, const std::vector<uint64_t>& v) {
        if (v.empty()) return;
        Pb in;
        for (uint64_t x : v) in.varint(x);
        msg(field, in);
    }
};
Pb pb_felt(uint64_t v) {            // FieldElement { bytes element = 2 }: little-endian bytes (convert_proof.rs:55-61)
    uint8_t le[8];
    for (int i = 0; i < 8; i++) le[i] = (uint8_t)(v >> (8 * i));
    Pb m;
    m.bytes(2, le, 8);
    return m;
}
Pb pb_digest(const Digest& d) {     // Digest { bytes data = 2 }
    Pb m;
    m.bytes(2, reinterpret_cast<const uint8_t*>(d.w), 32);
    return m;
}
void pb_felts(Pb& m, uint32_t field, const std::vector<uint64_t>& v) { for (uint64_t x : v) m.msg(field, pb_felt(x)); }
Pb pb_table(size_t rows, size_t cols, const std::vector<uint64_t>& data) {   // Table { n_rows = 1, n_cols = 2, elements = 3 }
    Pb m;
    m.u64(1, rows); m.u64(2, cols);
    pb_felts(m, 3, data);
    return m;
}
Pb pb_frame(const std::vector<uint64_t>& cur, const std::vector<uint64_t>& next) {
    Pb m;
    pb_felts(m, 1, cur); pb_felts(m, 2, next);
    return m;
}
// BatchMerkleProof { leaves = 1, nodes = 2 (layers of digests), depth = 3 } (convert_proof.rs:282-307)
Pb pb_batch(const std::vector<Digest>& leaves, const Bytes& paths, int depth) {
    Pb m;
    for (auto& l : leaves) m.msg(1, pb_digest(l));
    for (auto& vec : fmt::batch_vectors(paths)) {
        Pb layer;
        for (auto& d : vec) layer.msg(1, pb_digest(d));
        m.msg(2, layer);
    }
    m.u64(3, (uint64_t)depth);
    return m;
}
Bytes protobuf_proof(const Parsed& pr) {
    const Pieces p = cut(pr, "proof_to_protobuf");
    if (pr.opt.hash_fn != HASH_BLAKE2S_256) throw Error(AERO_E_UNSUPPORTED, "proof_to_protobuf: only Blake2s_256 has a protobuf enum value");
    Pb out;
    {   // Context = 1 (convert_proof.rs:71-88): trace_layout, trace_length, trace_meta, field_modulus, options
        Pb layout;
        layout.u64(1, p.W);
        if (p.A) { layout.packed(2, {(uint64_t)p.A}); layout.packed(3, {(uint64_t)pr.R}); layout.u64(4, 1); }
        Pb opts;   // ProofOptions (convert_proof.rs:109-128); BLAKE2S = 0, NONE = 0, GOLDILOCKS = 0 are not written
        opts.u64(1, pr.opt.num_queries); opts.u64(2, pr.opt.blowup_factor); opts.u64(3, pr.opt.grinding_factor);
        opts.u64(6, pr.opt.fri_folding_factor); opts.u64(7, 1ull << pr.opt.fri_log_max_remainder);
        Pb modulus;
        {
            uint8_t le[8];
            for (int i = 0; i < 8; i++) le[i] = (uint8_t)(gl::P >> (8 * i));
            modulus.bytes(2, le, 8);
        }
        Pb ctx;
        ctx.msg(1, layout);
        ctx.u64(2, p.n);
        ctx.bytes(3, pr.trace_meta.data(), pr.trace_meta.size());
        ctx.msg(4, modulus);
        ctx.msg(5, opts);
        out.msg(1, ctx);
    }
    {   // Commitments = 2 (convert_proof.rs:158-178)
        Pb c;
        for (auto& d : p.trace_roots) c.msg(1, pb_digest(d));
        c.msg(2, pb_digest(p.constraint_root));
        for (auto& d : p.fri_roots) c.msg(3, pb_digest(d));
        out.msg(2, c);
    }
    const int depth = ilog2z(p.N);
    {   // TraceQueries = 3 (convert_proof.rs:193-209): main_states, aux_states, one BatchMerkleProof per segment
        Pb t;
        t.msg(1, pb_table(p.Q, p.W, p.main_states));
        if (p.A) t.msg(2, pb_table(p.Q, p.A, p.aux_states));
        t.msg(3, pb_batch(row_hashes(p.main_states, p.Q, p.W), pr.trace_queries[0].paths, depth));
        if (p.A) t.msg(3, pb_batch(row_hashes(p.aux_states, p.Q, p.A), pr.trace_queries[1].paths, depth));
        out.msg(3, t);
    }
    {   // ConstraintQueries = 4 (convert_proof.rs:211-221)
        Pb c;
        c.msg(1, pb_table(p.Q, p.C, p.constraint_evals));
        c.msg(2, pb_batch(row_hashes(p.constraint_evals, p.Q, p.C), pr.constraint_queries.paths, depth));
        out.msg(4, c);
    }
    {   // OodFrame = 5 (convert_proof.rs:34-53)
        Pb o;
        o.msg(1, pb_frame(p.main_cur, p.main_next));
        if (p.A) o.msg(2, pb_frame(p.aux_cur, p.aux_next));
        pb_felts(o, 3, p.ood_evals);
        out.msg(5, o);
    }
    {   // FriProof = 6 (convert_proof.rs:223-255): layers (values + BatchMerkleProof), remainder, log2(num_partitions)
        Pb f;
        size_t dom = p.N;
        for (int l = 0; l < p.layers; l++) {
            const size_t rows = dom / p.F, nq = p.fri_values[l].size() / p.F;
            Pb layer;
            pb_felts(layer, 1, p.fri_values[l]);
            layer.msg(2, pb_batch(row_hashes(p.fri_values[l], nq, p.F), pr.fri_layers[l].paths, ilog2z(rows)));
            f.msg(1, layer);
            dom = rows;
        }
        pb_felts(f, 2, p.remainder);
        f.u64(3, pr.fri_log_partitions);
        out.msg(6, f);
    }
    out.u64(7, pr.nonce);   // pow_nonce = 7
    return out.b;
}
// MidenPublicInputs { program_hash = 1, stack_inputs = 2, outputs = 3 { stack = 1, overflow_addrs = 2 } } (convert_proof.rs:257-280)
Bytes protobuf_public_inputs(const uint8_t* b, size_t len) {
    const fmt::MidenInputs m = fmt::parse_miden_inputs(b, len);
    Pb out, hash, outputs;
    hash.bytes(2, b, 32);
    out.msg(1, hash);
    pb_felts(out, 2, m.stack_inputs);
    pb_felts(outputs, 1, m.out_stack);
    pb_felts(outputs, 2, m.overflow);
    out.msg(3, outputs);
    return out.b;
}

// MidenProgramOutputs { stack = 1, overflow_addrs = 2 } (miden_vm.proto:7-12): what the proving worker posts as `program_outputs`
Bytes protobuf_program_outputs(const uint8_t* b, size_t len) {
    const fmt::MidenInputs m = fmt::parse_miden_inputs(b, len);
    Pb outputs;
    pb_felts(outputs, 1, m.out_stack);
    pb_felts(outputs, 2, m.overflow);
    return outputs.b;
}

template <class Fn> int32_t guarded(char* err, size_t cap, Fn&& fn) {
    auto put = [&](const std::string& s) { if (err && cap) { size_t k = std::min(cap - 1, s.size()); memcpy(err, s.data(), k); err[k] = 0; } };
    try {
        put("");
        fn();
        return AERO_OK;
    } catch (const Error& e) {
        put(e.what());
        return e.code ? e.code : AERO_E_INTERNAL;
    } catch (const std::bad_alloc&) {
        put("host allocation failed");
        return AERO_E_OOM;
    } catch (const std::exception& e) {
        put(e.what());
        return AERO_E_INTERNAL;
    }
}

}  // namespace
}  // namespace aero

extern "C" {

int32_t aero_cairo_memory(uint32_t what, const uint8_t* proof, size_t proof_len, const uint8_t* input_bytes, size_t input_len, const uint64_t* indexes,
                          uint32_t n_indexes, char** json_out, size_t* json_len, char* err, size_t err_cap) {
    using namespace aero;
    return guarded(err, err_cap, [&] {
        if (!json_out || !json_len) fail("cairo_memory: null output");
        *json_out = nullptr; *json_len = 0;
        std::string js;
        if (what == AERO_CAIRO_PUBLIC_INPUTS) {
            if (!input_bytes) fail("cairo_memory: public-inputs needs the input bytes of the container");
            js = cairo_public_inputs(input_bytes, input_len);
        } else {
            if (!proof) fail("cairo_memory: null proof");
            const fmt::Parsed pr = fmt::parse(proof, proof_len);
            std::vector<uint64_t> idx;
            if (what != AERO_CAIRO_PROOF) {
                if (!indexes && n_indexes) fail("cairo_memory: null indexes");
                idx.assign(indexes, indexes + n_indexes);
                for (uint64_t v : idx) if (v >= pr.lde_domain_size()) fail("cairo_memory: index outside the LDE domain");
            }
            if (what == AERO_CAIRO_PROOF) js = cairo_proof(pr);
            else if (what == AERO_CAIRO_TRACE_QUERIES) js = cairo_trace_queries(pr, idx);
            else if (what == AERO_CAIRO_CONSTRAINT_QUERIES) js = cairo_constraint_queries(pr, idx);
            else if (what == AERO_CAIRO_FRI_QUERIES) js = cairo_fri_queries(pr, idx);
            else fail("cairo_memory: unknown sub-command");
        }
        char* buf = (char*)malloc(js.size() + 1);
        if (!buf) throw std::bad_alloc();
        memcpy(buf, js.c_str(), js.size() + 1);
        *json_out = buf; *json_len = js.size();
    });
}

int32_t aero_proof_to_protobuf(const uint8_t* proof, size_t proof_len, uint8_t** out, size_t* out_len, char* err, size_t err_cap) {
    using namespace aero;
    return guarded(err, err_cap, [&] {
        if (!proof || !out || !out_len) fail("proof_to_protobuf: null argument");
        const Bytes b = protobuf_proof(fmt::parse(proof, proof_len));
        uint8_t* buf = (uint8_t*)malloc(b.size() ? b.size() : 1);
        if (!buf) throw std::bad_alloc();
        memcpy(buf, b.data(), b.size());
        *out = buf; *out_len = b.size();
    });
}

int32_t aero_miden_public_inputs_to_protobuf(const uint8_t* input_bytes, size_t input_len, uint8_t** out, size_t* out_len, char* err, size_t err_cap) {
    using namespace aero;
    return guarded(err, err_cap, [&] {
        if (!input_bytes || !out || !out_len) fail("public_inputs_to_protobuf: null argument");
        const Bytes b = protobuf_public_inputs(input_bytes, input_len);
        uint8_t* buf = (uint8_t*)malloc(b.size() ? b.size() : 1);
        if (!buf) throw std::bad_alloc();
        memcpy(buf, b.data(), b.size());
        *out = buf; *out_len = b.size();
    });
}

// Host-side look at a worker message without touching a GPU: validates the layout and reports its shape.
int32_t aero_worker_message_info(uint32_t kind, const uint8_t* msg, size_t len, uint64_t out[8], char* err, size_t err_cap) {
    using namespace aero;
    return guarded(err, err_cap, [&] {
        if (!msg || !out) fail("worker_message_info: null argument");
        for (int i = 0; i < 8; i++) out[i] = 0;
        if (kind == AERO_MSG_HASHING_WORK_ITEM) {
            std::vector<uint64_t> offs;
            const uint64_t batch = wm::scan_hashing_work_item(msg, len, offs);
            uint64_t lo = ~0ull, hi = 0, total = 0;
            for (uint64_t o : offs) { uint64_t k; memcpy(&k, msg + 8 * o, 8); lo = k < lo ? k : lo; hi = k > hi ? k : hi; total += k; }
            out[0] = offs.size(); out[1] = batch; out[2] = offs.empty() ? 0 : lo; out[3] = hi; out[4] = total;
        } else if (kind == AERO_MSG_CONSTRAINT_WORK_ITEM) {
            const wm::ConstraintWorkItem w = wm::parse_constraint_work_item(msg, len);
            out[0] = w.main_width; out[1] = w.aux_width; out[2] = w.aux_rands; out[3] = w.trace_len; out[4] = w.blowup;
            out[5] = w.fragment_offset; out[6] = w.num_fragments; out[7] = w.n_transition + w.n_boundary;
        } else fail("worker_message_info: unknown message kind");
    });
}
// The message the reference's proving worker hands back to the SDK (proving_worker.rs:205-222, utils.rs:424-430): bincode
// ProverOutput { proof, program_outputs, public_inputs }, each the protobuf encoding of the SDK type.
int32_t aero_prover_output(const uint8_t* proof, size_t proof_len, const uint8_t* input_bytes, size_t input_len, uint8_t** out, size_t* out_len,
                           char* err, size_t err_cap) {
    using namespace aero;
    return guarded(err, err_cap, [&] {
        if (!proof || !input_bytes || !out || !out_len) fail("prover_output: null argument");
        *out = nullptr; *out_len = 0;
        const Bytes msg = wm::emit_prover_output(protobuf_proof(fmt::parse(proof, proof_len)), protobuf_program_outputs(input_bytes, input_len),
                                                 protobuf_public_inputs(input_bytes, input_len));
        uint8_t* buf = (uint8_t*)malloc(msg.size() ? msg.size() : 1);
        if (!buf) throw std::bad_alloc();
        memcpy(buf, msg.data(), msg.size());
        *out = buf; *out_len = msg.size();
    });
}

}  // extern "C"
