// Host-side reader of the proof bytes this backend emits (= `StarkProof::to_bytes`) and of the
// BatchMerkleProof blocks inside them. Shared by the verifier (verify.hip) and the re-encoders (export.hip: Cairo-memory
// JSON after miden-to-cairo-parser, protobuf after aero-sdk/miden-wasm/src/convert). No GPU code.
//
// Layout witnesses in the reference: miden-to-cairo-parser/src/lib.rs:65-75 (field order), :95-125 (commitments),
// :127-158 (OOD frame, queries), aero-sdk/miden-wasm/src/convert/convert_proof.rs:13-28; byte-level layout confirmed on
// proofs/fib.bin (tests/test_oracle_golden.py).
#pragma once
#include <algorithm>
#include <cstring>
#include <map>

#include "prover.hpp"

namespace aero {
namespace fmt {

struct FormatError : Error {
    explicit FormatError(const std::string& s) : Error(-7 /* AERO_E_VERIFY */, s) {}
};
[[noreturn]] inline void bad(const std::string& why) { throw FormatError("proof: " + why); }

struct Reader {
    const uint8_t* p;
    size_t n, off = 0;
    void need(size_t k) const { if (off + k > n) bad("proof is truncated"); }
    uint8_t u8() { need(1); return p[off++]; }
    uint64_t le(int bytes) { need(bytes); uint64_t v = 0; for (int i = 0; i < bytes; i++) v |= (uint64_t)p[off + i] << (8 * i); off += bytes; return v; }
    Bytes bytes(size_t k) { need(k); Bytes b(p + off, p + off + k); off += k; return b; }
};

struct Parsed {
    uint32_t W = 0, A = 0, R = 0;
    int log_n = 0;
    Bytes trace_meta;
    ProofOptions opt{};
    Bytes commitments;
    std::vector<QueriesBytes> trace_queries;
    QueriesBytes constraint_queries;
    Bytes ood_trace_states, ood_evaluations;
    std::vector<QueriesBytes> fri_layers;
    Bytes fri_remainder;
    uint8_t fri_log_partitions = 0;
    uint64_t nonce = 0;

    int deg() const { return opt.field_extension == EXT_QUADRATIC ? 2 : 1; }
    size_t trace_length() const { return (size_t)1 << log_n; }
    size_t lde_domain_size() const { return trace_length() * opt.blowup_factor; }
    int num_fri_layers() const { return aero::num_fri_layers(lde_domain_size(), opt.fri_folding_factor, 1ull << opt.fri_log_max_remainder); }
    size_t num_trace_segments() const { return A ? 2 : 1; }
    // composition columns = number of OOD constraint evaluations (`air.ce_blowup_factor()` for the AIRs on this path)
    size_t num_composition_columns() const { return ood_evaluations.size() / (8 * (size_t)deg()); }
    Digest root(size_t i) const { Digest d; memcpy(d.w, commitments.data() + 32 * i, 32); return d; }
    size_t num_roots() const { return commitments.size() / 32; }
};

inline Parsed parse(const uint8_t* data, size_t len) {
    Reader r{data, len};
    Parsed q;
    q.W = r.u8(); q.A = r.u8(); q.R = r.u8(); q.log_n = r.u8();
    q.trace_meta = r.bytes(r.le(2));
    if (r.u8() != 8 || r.le(8) != gl::P) bad("wrong field modulus");
    uint8_t o[7];
    for (auto& b : o) b = r.u8();
    q.opt = ProofOptions::from_bytes(o);
    q.commitments = r.bytes(r.le(2));
    for (int s = 0; s < (q.A ? 2 : 1); s++) {
        QueriesBytes t;
        t.values = r.bytes(r.le(4)); t.paths = r.bytes(r.le(4));
        q.trace_queries.push_back(t);
    }
    q.constraint_queries.values = r.bytes(r.le(4)); q.constraint_queries.paths = r.bytes(r.le(4));
    q.ood_trace_states = r.bytes(r.le(2));
    q.ood_evaluations = r.bytes(r.le(2));
    const int layers = r.u8();
    for (int l = 0; l < layers; l++) {
        QueriesBytes t;
        t.values = r.bytes(r.le(4)); t.paths = r.bytes(r.le(4));
        q.fri_layers.push_back(t);
    }
    q.fri_remainder = r.bytes(r.le(2));
    q.fri_log_partitions = r.u8();
    if (q.fri_log_partitions != 0) bad("partitioned FRI proofs are not supported");
    q.nonce = r.le(8);
    if (r.off != len) bad("trailing bytes after the proof");
    if (q.commitments.size() % 32) bad("commitments are not a whole number of digests");
    return q;
}

inline uint64_t rd64(const Bytes& b, size_t word) {
    if ((word + 1) * 8 > b.size()) bad("element index out of range");
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v |= (uint64_t)b[word * 8 + i] << (8 * i);
    if (v >= gl::P) bad("non-canonical field element");
    return v;
}

// A serialised BatchMerkleProof (`BatchMerkleProof::serialize_nodes`: u8 #vectors, per vector u8 len + len digests) as
// vectors of digests: winter-crypto's `nodes: Vec<Vec<Digest>>`.
inline std::vector<std::vector<Digest>> batch_vectors(const Bytes& paths) {
    std::vector<std::vector<Digest>> out;
    if (paths.empty()) bad("batch proof is empty");
    size_t off = 0;
    const size_t nv = paths[off++];
    for (size_t v = 0; v < nv; v++) {
        if (off >= paths.size()) bad("batch proof is truncated");
        const size_t len = paths[off++];
        std::vector<Digest> vec(len);
        for (size_t i = 0; i < len; i++) {
            if (off + 32 > paths.size()) bad("batch proof is truncated");
            memcpy(vec[i].w, paths.data() + off, 32);
            off += 32;
        }
        out.push_back(std::move(vec));
    }
    if (off != paths.size()) bad("batch proof: trailing bytes");
    return out;
}

// Every node a BatchMerkleProof determines: the opened leaves, the digests it carries (placed by the node-selection
// plan, prover.hip batch_proof_indices) and all their ancestors up to the root (heap indices: root 1, leaves n + j).
inline std::map<uint64_t, Digest> batch_known_nodes(size_t n_leaves, const std::vector<uint64_t>& positions, const std::vector<Digest>& leaves,
                                                    const Bytes& paths) {
    std::map<uint64_t, Digest> known;
    if (positions.size() != leaves.size()) bad("batch proof: leaves do not match the positions");
    if (n_leaves == 1) {
        if (positions.size() != 1 || positions[0] != 0) bad("bad opening of a single-leaf tree");
        known[1] = leaves[0];
        return known;
    }
    const auto plan = batch_proof_indices(n_leaves, positions);
    const auto vecs = batch_vectors(paths);
    if (vecs.size() != plan.size()) bad("batch proof: wrong number of paths");
    for (size_t v = 0; v < plan.size(); v++) {
        if (vecs[v].size() != plan[v].size()) bad("batch proof: wrong path length");
        for (size_t i = 0; i < plan[v].size(); i++) known[plan[v][i]] = vecs[v][i];
    }
    std::vector<uint64_t> level;
    for (size_t i = 0; i < positions.size(); i++) { known[n_leaves + positions[i]] = leaves[i]; level.push_back(n_leaves + positions[i]); }
    while (!level.empty() && level[0] > 1) {
        std::sort(level.begin(), level.end());
        level.erase(std::unique(level.begin(), level.end()), level.end());
        std::vector<uint64_t> next;
        for (uint64_t idx : level) {
            const uint64_t parent = idx >> 1;
            if (known.count(parent)) continue;
            auto l = known.find(idx & ~1ull), r = known.find(idx | 1ull);
            if (l == known.end() || r == known.end()) bad("batch proof: missing sibling");
            known[parent] = b2s::merge(l->second, r->second);
            next.push_back(parent);
        }
        level.swap(next);
    }
    if (!known.count(1)) bad("batch proof: root not reached");
    return known;
}
inline Digest batch_root(size_t n_leaves, const std::vector<uint64_t>& positions, const std::vector<Digest>& leaves, const Bytes& paths) {
    return batch_known_nodes(n_leaves, positions, leaves, paths).at(1);
}
// `BatchMerkleProof::into_paths(indexes)`: one authentication path per position, in position order; path = the leaf, then
// the sibling at every level from the leaves up to (not including) the root — depth + 1 digests, the form
// src/stark_verifier/channel.cairo:206-244 (`verify_merkle_proof(length, path, position, root)`) consumes.
inline std::vector<std::vector<Digest>> batch_into_paths(size_t n_leaves, const std::vector<uint64_t>& positions, const std::vector<Digest>& leaves,
                                                         const Bytes& paths) {
    const auto known = batch_known_nodes(n_leaves, positions, leaves, paths);
    std::vector<std::vector<Digest>> out;
    for (size_t i = 0; i < positions.size(); i++) {
        std::vector<Digest> path{leaves[i]};
        for (uint64_t idx = n_leaves + positions[i]; idx > 1; idx >>= 1) {
            auto it = known.find(idx ^ 1);
            if (it == known.end()) bad("batch proof: a path needs a node the proof does not determine");
            path.push_back(it->second);
        }
        out.push_back(std::move(path));
    }
    return out;
}

// Miden's `PublicInputs` as the bincode container and the worker messages carry them: program hash (4 elements),
// then three u64-counted lists: stack inputs, outputs.stack, overflow addresses.
struct MidenInputs { std::vector<uint64_t> hash, stack_inputs, out_stack, overflow; };
inline MidenInputs parse_miden_inputs(const uint8_t* b, size_t len) {
    Reader r{b, len};
    MidenInputs m;
    for (int i = 0; i < 4; i++) m.hash.push_back(r.le(8));
    std::vector<uint64_t>* parts[3] = {&m.stack_inputs, &m.out_stack, &m.overflow};
    for (auto* v : parts) {
        const uint64_t cnt = r.le(8);
        if (cnt > (len - r.off) / 8) bad("public inputs are truncated");
        for (uint64_t i = 0; i < cnt; i++) v->push_back(r.le(8));
    }
    if (r.off != len) bad("trailing bytes after the public inputs");
    for (uint64_t v : m.hash) if (v >= gl::P) bad("non-canonical program hash element");
    for (uint64_t v : m.stack_inputs) if (v >= gl::P) bad("non-canonical stack input");
    return m;
}

}  // namespace fmt
}  // namespace aero
