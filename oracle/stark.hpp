// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/gl.hpp header). Never linked into libaero_stark.so.
//
// CPU restatement of the Winterfell 0.4 (starkoracles fork) proving pipeline that Aero's
// miden-proof-generator drives, plus the matching verifier.
//
// The reference's own prover source is NOT in /root/reference (winterfell/ and miden/ are empty,
// un-vendored git submodules: /root/reference/.gitmodules:1-6; crates winter-{prover,air,math,crypto,
// fri,utils} semver "0.4", commit unpinned because Cargo.lock is git-ignored). What IS in the tree and is
// followed line by line here:
//   verifier / transcript ......... src/stark_verifier/stark_verifier.cairo:105-304
//   coin, hash_elements ........... src/stark_verifier/crypto/random.cairo:31-342
//   Merkle node / path hashing .... src/stark_verifier/channel.cairo:136-244
//   DEEP composition .............. src/stark_verifier/composer.cairo:17-316
//   FRI verification + geometry ... src/stark_verifier/fri/fri_verifier.cairo:56-461, channel.cairo:80-133
//   coefficient draw order ........ src/stark_verifier/air/air_instance.cairo:115-205
//   proof field order / layout .... miden-to-cairo-parser/src/lib.rs:65-208,395-436,
//                                   aero-sdk/miden-wasm/src/convert/convert_proof.rs:13-307
//   prover stage order / seams .... aero-sdk/miden-wasm/src/proving_worker.rs:238-439,
//                                   constraints_worker.rs:14-79, hashing_worker.rs:12-26
// Prover internals that only exist upstream (constraint degree adjustment, divisor handling, composition
// column split, BatchMerkleProof node order, FRI layer build, grinding) restate the published winter-*
// 0.4 algorithms; they are pinned by (i) the golden proof proofs/fib.bin, which this verifier must accept
// byte-for-byte, and (ii) prove->verify self-consistency including the OOD constraint check that the Cairo
// verifier leaves commented out (stark_verifier.cairo:151-159,183-187).
#pragma once
#include <algorithm>
#include <cstdio>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "blake2s.hpp"
#include "gl.hpp"

namespace orc {

typedef std::vector<uint8_t> Bytes;
typedef std::vector<uint64_t> Col;

struct Err : std::runtime_error { explicit Err(const std::string& s) : std::runtime_error(s) {} };

static inline int ilog2(uint64_t x) { int r = 0; while ((1ULL << r) < x) r++; return r; }

// ------------------------------------------------------------------------------------------------
// Proof options — the 7 bytes at the end of the proof context (context.proto:24-33,
// convert_inputs.rs:54-66). Enum ids: hash_fn Blake2s_256 = 4, field_ext None = 1 / Quadratic = 2
// (fib.bin header bytes `1b 08 10 04 01 08 08`).
struct Options {
    uint8_t num_queries, blowup, grinding, hash_fn, field_ext, fri_fold, log_max_remainder;
};
static const uint8_t HASH_BLAKE2S_256 = 4, EXT_NONE = 1, EXT_QUADRATIC = 2;

// ------------------------------------------------------------------------------------------------
// Field traits so that prover/verifier are written once for E = F_p and E = F_p^2.
struct FB {
    typedef uint64_t T;
    enum { DEG = 1 };
    static T zero() { return 0; }
    static T one() { return 1; }
    static T from(uint64_t b) { return b; }
    static T add(T a, T b) { return gl_add(a, b); }
    static T sub(T a, T b) { return gl_sub(a, b); }
    static T mul(T a, T b) { return gl_mul(a, b); }
    static T mulb(T a, uint64_t b) { return gl_mul(a, b); }
    static T inv(T a) { return gl_inv(a); }
    static T conj(T a) { return a; }
    static bool eq(T a, T b) { return a == b; }
    static uint64_t comp(T a, int) { return a; }
    static T make(const uint64_t* c) { return c[0]; }
};
struct FQ {
    typedef Fe2 T;
    enum { DEG = 2 };
    static T zero() { return Fe2{0, 0}; }
    static T one() { return Fe2{1, 0}; }
    static T from(uint64_t b) { return Fe2{b, 0}; }
    static T add(T a, T b) { return e2_add(a, b); }
    static T sub(T a, T b) { return e2_sub(a, b); }
    static T mul(T a, T b) { return e2_mul(a, b); }
    static T mulb(T a, uint64_t b) { return e2_mulb(a, b); }
    static T inv(T a) { return e2_inv(a); }
    static T conj(T a) { return e2_conj(a); }
    static bool eq(T a, T b) { return e2_eq(a, b); }
    static uint64_t comp(T a, int i) { return i ? a.a1 : a.a0; }
    static T make(const uint64_t* c) { return Fe2{c[0], c[1]}; }
};
template <class F> static typename F::T f_pow(typename F::T b, uint64_t e) {
    typename F::T r = F::one();
    while (e) { if (e & 1) r = F::mul(r, b); b = F::mul(b, b); e >>= 1; }
    return r;
}
// E-elements -> flat base elements (a0, a1 per element) for hashing / serialisation.
template <class F> static void f_flatten(const typename F::T* v, size_t n, Col& out) {
    for (size_t i = 0; i < n; i++) for (int k = 0; k < F::DEG; k++) out.push_back(F::comp(v[i], k));
}
template <class F> static Digest f_hash(const typename F::T* v, size_t n) {
    if (n * F::DEG <= 64) {   // hot path (FRI rows): no heap traffic
        uint64_t flat[64];
        size_t k = 0;
        for (size_t i = 0; i < n; i++) for (int d = 0; d < F::DEG; d++) flat[k++] = F::comp(v[i], d);
        return hash_elements(flat, k);
    }
    Col flat; f_flatten<F>(v, n, flat);
    return hash_elements(flat.data(), flat.size());
}

// ------------------------------------------------------------------------------------------------
// Random coin — random.cairo:31-37 (new), :108-128,320-326 (reseed), :159-166 (draw),
// :201-252 (draw_integers), :282-316 (leading zeros). Field `seed`,`counter`: parser lib.rs:335-339.
struct Coin {
    Digest seed;
    uint64_t ctr;
    // seed0 = hash_elements(public-input elements) (random.cairo:254-280), coin seed = BLAKE2s(seed0)
    // (random.cairo:31-37 with n_bytes = 32, stark_verifier.cairo:83-91).
    static Coin from_pub_elements(const uint64_t* e, size_t n) {
        Digest s0 = hash_elements(e, n);
        Coin c; c.seed = blake2s(s0.b, 32); c.ctr = 0;
        return c;
    }
    void reseed(const Digest& d) { seed = merge(seed, d); ctr = 0; }                 // reseed_endian
    void reseed_int(uint64_t v) { seed = merge_with_int(seed, v); ctr = 0; }         // reseed_with_int
    Digest next() { ctr += 1; return merge_with_int(seed, ctr); }                    // draw_digest
    static uint64_t le64(const uint8_t* p) { uint64_t v = 0; for (int i = 7; i >= 0; i--) v = (v << 8) | p[i]; return v; }
    // Base-field draw: first 8 bytes LE (random.cairo:165); values >= p are rejected and redrawn
    // (winter-crypto 0.4 RandomCoin::draw / from_random_bytes — upstream behaviour, probability 2^-32).
    uint64_t draw_base() {
        for (int i = 0; i < 1000; i++) { Digest d = next(); uint64_t v = le64(d.b); if (v < P) return v; }
        throw Err("coin: failed to draw");
    }
    // Quadratic draw: first 16 bytes = two LE u64, both must be canonical (winter-math QuadExtension::from_random_bytes).
    Fe2 draw_quad() {
        for (int i = 0; i < 1000; i++) {
            Digest d = next(); uint64_t a = le64(d.b), b = le64(d.b + 8);
            if (a < P && b < P) return Fe2{a, b};
        }
        throw Err("coin: failed to draw");
    }
    template <class F> typename F::T draw();
    std::vector<uint64_t> draw_integers(size_t k, uint64_t domain) {                 // random.cairo:210-252
        std::vector<uint64_t> out;
        uint64_t mask = domain - 1;
        for (int i = 0; i < 1000 && out.size() < k; i++) {
            Digest d = next(); uint64_t v = le64(d.b) & mask;
            if (std::find(out.begin(), out.end(), v) == out.end()) out.push_back(v);
        }
        if (out.size() != k) throw Err("coin: failed to draw integers");
        return out;
    }
    // Leading zero bits of the digest read big-endian from byte 0 (random.cairo:282-316, capped at 64).
    // For grinding factors that are multiples of 8 this equals winter-crypto 0.4's
    // `u64::from_le_bytes(digest[..8]).trailing_zeros()`; fib.bin (grinding 16) cannot distinguish them.
    static uint32_t leading_zeros(const Digest& d) {
        uint32_t z = 0;
        for (int i = 0; i < 8; i++) {
            if (d.b[i] == 0) { z += 8; continue; }
            for (int bit = 7; bit >= 0; bit--) { if (d.b[i] >> bit & 1) return z; z++; }
        }
        return z;
    }
    uint32_t check_leading_zeros(uint64_t nonce) const { return leading_zeros(merge_with_int(seed, nonce)); }
};
template <> inline uint64_t Coin::draw<FB>() { return draw_base(); }
template <> inline Fe2 Coin::draw<FQ>() { return draw_quad(); }

// ------------------------------------------------------------------------------------------------
// Merkle tree — channel.cairo:136-175 (pairwise BLAKE2s of 64 bytes), root = node 1, children 2i, 2i+1.
// t[N + i] = leaf i, t[i] = merge(t[2i], t[2i+1]).
struct MerkleTree {
    std::vector<Digest> t;
    size_t n = 0;
    MerkleTree() {}
    explicit MerkleTree(const std::vector<Digest>& leaves) {
        n = leaves.size();
        if (n < 2 || (n & (n - 1))) throw Err("merkle: leaf count must be a power of two >= 2");
        t.resize(2 * n);
#pragma omp parallel for schedule(static) if (n >= 4096)
        for (size_t i = 0; i < n; i++) t[n + i] = leaves[i];
        for (size_t lvl = n / 2; lvl >= 1; lvl /= 2) {
#pragma omp parallel for schedule(static) if (lvl >= 4096)
            for (size_t i = lvl; i < 2 * lvl; i++) t[i] = merge(t[2 * i], t[2 * i + 1]);
        }
    }
    const Digest& root() const { return t[1]; }
    int depth() const { return ilog2(n); }
};

// BatchMerkleProof node selection (winter-crypto 0.4 MerkleTree::prove_batch, restated):
// indexes normalised to even, ascending; at level 0 each normalised pair contributes the leaf that was not
// queried; on the way up `nodes[i]` is indexed by the cursor position in the CURRENT level's index list.
static std::vector<std::vector<Digest>> batch_prove(const MerkleTree& mt, const std::vector<uint64_t>& positions) {
    std::set<uint64_t> qs(positions.begin(), positions.end());
    if (qs.size() != positions.size()) throw Err("merkle: duplicate positions");
    std::set<uint64_t> norm;
    for (uint64_t p : positions) { if (p >= mt.n) throw Err("merkle: position out of range"); norm.insert(p - (p & 1)); }
    std::vector<std::vector<Digest>> nodes;
    std::vector<uint64_t> next;
    for (uint64_t e : norm) {
        std::vector<Digest> miss;
        for (uint64_t i = e; i < e + 2; i++) if (!qs.count(i)) miss.push_back(mt.t[mt.n + i]);
        nodes.push_back(miss);
        next.push_back((e + mt.n) >> 1);
    }
    for (int lvl = 1; lvl < mt.depth(); lvl++) {
        std::vector<uint64_t> idx = next;
        next.clear();
        size_t i = 0;
        while (i < idx.size()) {
            uint64_t sib = idx[i] ^ 1;
            if (i + 1 < idx.size() && idx[i + 1] == sib) i += 1;
            else nodes[i].push_back(mt.t[sib]);
            next.push_back(sib >> 1);
            i += 1;
        }
    }
    return nodes;
}
// serialize_nodes: u8 #vectors, then per vector u8 len + len*32 bytes.
static Bytes batch_serialize(const std::vector<std::vector<Digest>>& nodes) {
    Bytes out;
    if (nodes.size() > 255) throw Err("merkle: too many paths");
    out.push_back((uint8_t)nodes.size());
    for (auto& v : nodes) {
        if (v.size() > 255) throw Err("merkle: too many nodes");
        out.push_back((uint8_t)v.size());
        for (auto& d : v) out.insert(out.end(), d.b, d.b + 32);
    }
    return out;
}
static std::vector<std::vector<Digest>> batch_parse(const uint8_t* p, size_t len) {
    size_t off = 0;
    if (len < 1) throw Err("batch proof: truncated");
    size_t nv = p[off++];
    std::vector<std::vector<Digest>> nodes(nv);
    for (size_t v = 0; v < nv; v++) {
        if (off >= len) throw Err("batch proof: truncated");
        size_t k = p[off++];
        if (off + 32 * k > len) throw Err("batch proof: truncated");
        nodes[v].resize(k);
        for (size_t j = 0; j < k; j++) { memcpy(nodes[v][j].b, p + off, 32); off += 32; }
    }
    if (off != len) throw Err("batch proof: trailing bytes");
    return nodes;
}
// BatchMerkleProof::get_root restated: leaves are given in the (unsorted) order of `positions`.
static Digest batch_root(const std::vector<uint64_t>& positions, const std::vector<Digest>& leaves,
                         const std::vector<std::vector<Digest>>& nodes, int depth) {
    std::map<uint64_t, size_t> index_map;
    for (size_t i = 0; i < positions.size(); i++) {
        if (index_map.count(positions[i])) throw Err("batch proof: duplicate positions");
        index_map[positions[i]] = i;
    }
    std::set<uint64_t> norm;
    for (uint64_t p : positions) norm.insert(p - (p & 1));
    if (norm.size() != nodes.size()) throw Err("batch proof: vector count mismatch");
    uint64_t offset = 1ULL << depth;
    std::map<uint64_t, Digest> v;
    std::vector<uint64_t> next;
    std::vector<size_t> ptr;
    size_t i = 0;
    for (uint64_t e : norm) {
        Digest a, b;
        bool ha = index_map.count(e), hb = index_map.count(e + 1);
        if (ha && hb) { a = leaves[index_map[e]]; b = leaves[index_map[e + 1]]; ptr.push_back(0); }
        else if (ha) { if (nodes[i].empty()) throw Err("batch proof: missing node"); a = leaves[index_map[e]]; b = nodes[i][0]; ptr.push_back(1); }
        else { if (nodes[i].empty() || !hb) throw Err("batch proof: missing node"); a = nodes[i][0]; b = leaves[index_map[e + 1]]; ptr.push_back(1); }
        uint64_t parent = (offset + e) >> 1;
        v[parent] = merge(a, b);
        next.push_back(parent);
        i++;
    }
    for (int lvl = 1; lvl < depth; lvl++) {
        std::vector<uint64_t> idx = next;
        next.clear();
        size_t c = 0;
        while (c < idx.size()) {
            uint64_t node = idx[c], sib = node ^ 1;
            Digest s;
            if (c + 1 < idx.size() && idx[c + 1] == sib) { s = v.at(sib); c += 1; }
            else {
                // the sibling is the next unused digest of nodes[cursor position in THIS level's list]
                size_t slot = c;
                if (ptr[slot] >= nodes[slot].size()) throw Err("batch proof: missing node");
                s = nodes[slot][ptr[slot]++];
            }
            const Digest& nd = v.at(node);
            v[node >> 1] = (node & 1) ? merge(s, nd) : merge(nd, s);
            next.push_back(node >> 1);
            c += 1;
        }
    }
    return v.at(1);
}

// ------------------------------------------------------------------------------------------------
// Byte helpers (all little-endian — SURVEY a18).
struct Reader {
    const uint8_t* p; size_t len, off = 0;
    Reader(const uint8_t* p_, size_t l) : p(p_), len(l) {}
    void need(size_t k) const { if (off + k > len) throw Err("proof: truncated"); }
    uint8_t u8() { need(1); return p[off++]; }
    uint16_t u16() { need(2); uint16_t v = p[off] | (p[off + 1] << 8); off += 2; return v; }
    uint32_t u32() { need(4); uint32_t v = 0; for (int i = 3; i >= 0; i--) v = (v << 8) | p[off + i]; off += 4; return v; }
    uint64_t u64() { need(8); uint64_t v = Coin::le64(p + off); off += 8; return v; }
    Bytes bytes(size_t k) { need(k); Bytes b(p + off, p + off + k); off += k; return b; }
};
static inline void w8(Bytes& b, uint8_t v) { b.push_back(v); }
static inline void w16(Bytes& b, uint16_t v) { b.push_back(v & 0xff); b.push_back(v >> 8); }
static inline void w32(Bytes& b, uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((v >> (8 * i)) & 0xff); }
static inline void w64(Bytes& b, uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((v >> (8 * i)) & 0xff); }
static inline void wbytes(Bytes& b, const Bytes& s) { b.insert(b.end(), s.begin(), s.end()); }

// ------------------------------------------------------------------------------------------------
// StarkProof container (field order parser lib.rs:65-75, convert_proof.rs:13-28).
struct Proof {
    uint8_t main_width = 0, aux_width = 0, aux_rands = 0, log_n = 0;
    Bytes meta, modulus;
    Options opt{};
    Bytes commitments;                       // concatenated 32-byte roots
    struct Q { Bytes values, paths; };
    std::vector<Q> trace_queries;            // one per trace segment
    Q constraint_queries;
    Bytes ood_trace_states, ood_evaluations;
    std::vector<Q> fri_layers;
    Bytes fri_remainder;
    uint8_t fri_log_partitions = 0;
    uint64_t pow_nonce = 0;

    int num_segments() const { return 1 + (aux_width ? 1 : 0); }

    static Proof parse(const uint8_t* p, size_t len) {
        Reader r(p, len);
        Proof pr;
        pr.main_width = r.u8(); pr.aux_width = r.u8(); pr.aux_rands = r.u8(); pr.log_n = r.u8();
        pr.meta = r.bytes(r.u16());
        pr.modulus = r.bytes(r.u8());
        pr.opt.num_queries = r.u8(); pr.opt.blowup = r.u8(); pr.opt.grinding = r.u8(); pr.opt.hash_fn = r.u8();
        pr.opt.field_ext = r.u8(); pr.opt.fri_fold = r.u8(); pr.opt.log_max_remainder = r.u8();
        pr.commitments = r.bytes(r.u16());
        for (int s = 0; s < pr.num_segments(); s++) { Q q; q.values = r.bytes(r.u32()); q.paths = r.bytes(r.u32()); pr.trace_queries.push_back(q); }
        pr.constraint_queries.values = r.bytes(r.u32()); pr.constraint_queries.paths = r.bytes(r.u32());
        pr.ood_trace_states = r.bytes(r.u16()); pr.ood_evaluations = r.bytes(r.u16());
        int nl = r.u8();
        for (int l = 0; l < nl; l++) { Q q; q.values = r.bytes(r.u32()); q.paths = r.bytes(r.u32()); pr.fri_layers.push_back(q); }
        pr.fri_remainder = r.bytes(r.u16());
        pr.fri_log_partitions = r.u8();
        pr.pow_nonce = r.u64();
        if (r.off != len) throw Err("proof: trailing bytes");
        return pr;
    }
    Bytes to_bytes() const {
        Bytes b;
        w8(b, main_width); w8(b, aux_width); w8(b, aux_rands); w8(b, log_n);
        w16(b, (uint16_t)meta.size()); wbytes(b, meta);
        w8(b, (uint8_t)modulus.size()); wbytes(b, modulus);
        w8(b, opt.num_queries); w8(b, opt.blowup); w8(b, opt.grinding); w8(b, opt.hash_fn);
        w8(b, opt.field_ext); w8(b, opt.fri_fold); w8(b, opt.log_max_remainder);
        w16(b, (uint16_t)commitments.size()); wbytes(b, commitments);
        for (auto& q : trace_queries) { w32(b, (uint32_t)q.values.size()); wbytes(b, q.values); w32(b, (uint32_t)q.paths.size()); wbytes(b, q.paths); }
        w32(b, (uint32_t)constraint_queries.values.size()); wbytes(b, constraint_queries.values);
        w32(b, (uint32_t)constraint_queries.paths.size()); wbytes(b, constraint_queries.paths);
        w16(b, (uint16_t)ood_trace_states.size()); wbytes(b, ood_trace_states);
        w16(b, (uint16_t)ood_evaluations.size()); wbytes(b, ood_evaluations);
        w8(b, (uint8_t)fri_layers.size());
        for (auto& q : fri_layers) { w32(b, (uint32_t)q.values.size()); wbytes(b, q.values); w32(b, (uint32_t)q.paths.size()); wbytes(b, q.paths); }
        w16(b, (uint16_t)fri_remainder.size()); wbytes(b, fri_remainder);
        w8(b, fri_log_partitions);
        w64(b, pow_nonce);
        return b;
    }
};

// Outer container: bincode ProofData{input_bytes, proof_bytes} = u64 len || bytes || u64 len || bytes
// (miden-proof-generator/src/lib.rs:1-6, main.rs:49-51; parser lib.rs:25-39).
static void container_split(const uint8_t* p, size_t len, Bytes& inputs, Bytes& proof) {
    Reader r(p, len);
    inputs = r.bytes(r.u64());
    proof = r.bytes(r.u64());
    if (r.off != len) throw Err("container: trailing bytes");
}
static Bytes container_join(const Bytes& inputs, const Bytes& proof) {
    Bytes b; w64(b, inputs.size()); wbytes(b, inputs); w64(b, proof.size()); wbytes(b, proof);
    return b;
}
// Miden PublicInputs bytes -> coin seed elements (SURVEY a19 + random.cairo:254-280):
// 32 B program hash (4 LE u64) || u64 count || stack_inputs || u64 count || outputs.stack || u64 count || overflow_addrs.
// Known answer: program-hash elements of fib.bin (tests/integration/test_verifier.cairo:41-47).
static Col miden_pub_elements(const Bytes& in) {
    Reader r(in.data(), in.size());
    Col e;
    for (int i = 0; i < 4; i++) e.push_back(r.u64());
    for (int part = 0; part < 3; part++) { uint64_t k = r.u64(); for (uint64_t i = 0; i < k; i++) e.push_back(r.u64()); }
    if (r.off != in.size()) throw Err("public inputs: trailing bytes");
    return e;
}

// ------------------------------------------------------------------------------------------------
// FRI geometry helpers (parser lib.rs:421-436 fold_positions; fri_verifier.cairo:208-215 num layers).
static std::vector<uint64_t> fold_positions(const std::vector<uint64_t>& pos, uint64_t src_domain, uint64_t fold) {
    uint64_t target = src_domain / fold;
    std::vector<uint64_t> out;
    for (uint64_t p : pos) { uint64_t q = p % target; if (std::find(out.begin(), out.end(), q) == out.end()) out.push_back(q); }
    return out;
}
static int num_fri_layers(uint64_t domain, uint64_t fold, uint64_t max_remainder) {
    int r = 0;
    while (domain > max_remainder) { domain /= fold; r++; }
    return r;
}

// Degree-<k interpolant through (xs[j], ys[j]) evaluated at x (fri/polynomials.cairo:8-52: Lagrange form).
template <class F> static typename F::T lagrange_eval(const uint64_t* xs, const typename F::T* ys, int k, typename F::T x) {
    typename F::T acc = F::zero();
    for (int j = 0; j < k; j++) {
        typename F::T num = F::one();
        uint64_t den = 1;
        for (int m = 0; m < k; m++) if (m != j) {
            num = F::mul(num, F::sub(x, F::from(xs[m])));
            den = gl_mul(den, gl_sub(xs[j], xs[m]));
        }
        acc = F::add(acc, F::mul(ys[j], F::mulb(num, gl_inv(den))));
    }
    return acc;
}

// ------------------------------------------------------------------------------------------------
// NTT (natural order in, natural order out). Radix-2 DIT after a bit-reversal; twiddles w^k, k < n/2.
static void bit_reverse(uint64_t* a, size_t n, bool par = false) {
    int lg = ilog2(n);
    if (lg == 0) return;
#pragma omp parallel for schedule(static) if (par && n >= 65536)      // every swap touches its own pair (i, rev(i)), done by the smaller index only
    for (size_t i = 0; i < n; i++) {
        uint64_t x = i;   // byte-swap + bit tricks: reverse 64 bits, keep the top lg
        x = __builtin_bswap64(x);
        x = ((x & 0xF0F0F0F0F0F0F0F0ull) >> 4) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
        x = ((x & 0xCCCCCCCCCCCCCCCCull) >> 2) | ((x & 0x3333333333333333ull) << 2);
        x = ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((x & 0x5555555555555555ull) << 1);
        size_t j = (size_t)(x >> (64 - lg));
        if (i < j) std::swap(a[i], a[j]);
    }
}
// out[i] = c0 * base^i, i < count: chunks that start from their own power (parallel for long tables; the same values)
static void geometric(uint64_t* out, size_t count, uint64_t c0, uint64_t base) {
#pragma omp parallel if (count >= 65536)
    {
#ifdef _OPENMP
        const size_t nt = (size_t)omp_get_num_threads(), t = (size_t)omp_get_thread_num();
#else
        const size_t nt = 1, t = 0;
#endif
        const size_t lo = count * t / nt, hi = count * (t + 1) / nt;
        uint64_t w = gl_mul(c0, gl_pow(base, lo));
        for (size_t i = lo; i < hi; i++) { out[i] = w; w = gl_mul(w, base); }
    }
}
static Col twiddles(size_t n, uint64_t root) {
    Col tw(n / 2 ? n / 2 : 1);
    tw[0] = 1;
    geometric(tw.data(), n / 2, 1, root);
    return tw;
}
static void ntt_core(uint64_t* a, size_t n, const Col& tw, bool par = false) {
    bit_reverse(a, n, par);
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len / 2, step = n / len;
#pragma omp parallel for schedule(static) if (par && n >= 16384)
        for (size_t b = 0; b < n / 2; b++) {
            size_t s = (b / half) * len, j = b % half;
            uint64_t u = a[s + j], v = gl_mul(a[s + j + half], tw[j * step]);
            a[s + j] = gl_add(u, v);
            a[s + j + half] = gl_sub(u, v);
        }
    }
}
// evaluations over <w_n> (natural order) -> coefficients          [interpolate_columns, proving_worker.rs:273]
static void intt(uint64_t* a, size_t n, bool par = false) {
    Col tw = twiddles(n, gl_inv(gl_root_of_unity(ilog2(n))));
    ntt_core(a, n, tw, par);
    uint64_t ninv = gl_inv(n);
#pragma omp parallel for schedule(static) if (par && n >= 16384)
    for (size_t i = 0; i < n; i++) a[i] = gl_mul(a[i], ninv);
}
// coefficients (n) -> evaluations on offset * <w_{n*blowup}>, natural order: row j <-> x_j = offset * w_N^j
// (composer.cairo:34-38).                                           [evaluate_columns_over, proving_worker.rs:274]
// Done as `blowup` size-n NTTs: coset k uses coefficients scaled by (offset * w_N^k)^i and fills rows k + blowup*m.
static Col lde(const uint64_t* coeffs, size_t n, size_t blowup, uint64_t offset) {
    size_t N = n * blowup;
    Col out(N);
    uint64_t wN = gl_root_of_unity(ilog2(N));
    Col tw = twiddles(n, gl_root_of_unity(ilog2(n)));
#ifdef _OPENMP
    // More threads than cosets (a column's LDE has only `blowup` independent transforms: 8 threads' worth): the cosets one after the other,
    // every transform parallel over its butterflies, the coefficient scaling over chunks that start from their own power - what keeps an
    // all-core host busy on narrow traces (the CPU baseline of bench.py; same values, same order of field operations per element).
    // (only for long transforms: at 2^14 points a parallel region per butterfly stage costs more than the stage, most of all with 128 - 256 threads)
    if (n >= ((size_t)1 << 18) && (size_t)omp_get_max_threads() > blowup) {
        Col tmp(n);
        for (size_t k = 0; k < blowup; k++) {
            const uint64_t s = gl_mul(offset, gl_pow(wN, k));
#pragma omp parallel
            {
                const size_t nt = (size_t)omp_get_num_threads(), t = (size_t)omp_get_thread_num();
                const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
                uint64_t acc = gl_pow(s, lo);
                for (size_t i = lo; i < hi; i++) { tmp[i] = gl_mul(coeffs[i], acc); acc = gl_mul(acc, s); }
            }
            ntt_core(tmp.data(), n, tw, true);
#pragma omp parallel for schedule(static)
            for (size_t m = 0; m < n; m++) out[k + blowup * m] = tmp[m];
        }
        return out;
    }
#endif
#pragma omp parallel for schedule(dynamic, 1) if (n >= 4096)
    for (size_t k = 0; k < blowup; k++) {
        Col tmp(n);
        uint64_t s = gl_mul(offset, gl_pow(wN, k)), acc = 1;
        for (size_t i = 0; i < n; i++) { tmp[i] = gl_mul(coeffs[i], acc); acc = gl_mul(acc, s); }
        ntt_core(tmp.data(), n, tw);
        for (size_t m = 0; m < n; m++) out[k + blowup * m] = tmp[m];
    }
    return out;
}
// evaluations on offset * <w_n> -> coefficients (interpolate_poly_with_offset).
static void intt_coset(uint64_t* a, size_t n, uint64_t offset, bool par = false) {
    intt(a, n, par);
    const uint64_t oi = gl_inv(offset);
    if (par && n >= 65536) {
        Col pw(n);
        geometric(pw.data(), n, 1, oi);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) a[i] = gl_mul(a[i], pw[i]);
        return;
    }
    uint64_t acc = 1;
    for (size_t i = 0; i < n; i++) { a[i] = gl_mul(a[i], acc); acc = gl_mul(acc, oi); }
}
template <class F> static typename F::T horner(const uint64_t* c, size_t n, typename F::T x) {
    typename F::T acc = F::zero();
    for (size_t i = n; i-- > 0;) acc = F::add(F::mul(acc, x), F::from(c[i]));
    return acc;
}

// ------------------------------------------------------------------------------------------------
// AIR descriptions. Built-in AIR: FibAir(W), W even; pair k = columns (2k, 2k+1) = (a, b) with
// a' = a + b, b' = b + a'; seeds (1+2k, 2+2k); assertions a(0), b(0), b(n-1); public inputs = the W/2 results.
// AIR_OPAQUE describes a proof whose constraint system is not available (the golden Miden proof fib.bin):
// everything except the OOD constraint check is verified — exactly what src/stark_verifier does.
enum AirKind { AIR_OPAQUE = 0, AIR_FIB = 1 };

// Optional auxiliary segment (a synthetic stand-in for Miden's multiset-check columns, whose
// AIR is absent from the mount): A columns over E, built after the main commitment from R random elements drawn from the
// coin: p_c(0) = 1, p_c(i+1) = p_c(i) * (r_(c mod R) + main_(c mod W)(i))^(D-1); one degree-D transition constraint and one
// assertion p_c(0) = 1 per aux column. D = 2 is the plain multiset-check shape; D up to 8 raises the constraint-evaluation
// blowup and the number of composition columns to 8, the shape of the golden Miden proof (fib.bin: 8 composition columns). Transcript order, proof layout, OOD frame (main || aux) and DEEP coefficient order
// follow stark_verifier.cairo:117-130,266-294 (pinned by fib.bin, which has one aux segment); the constraint set itself
// is restatement-defined.
struct FibAir {
    enum { KNOWN = 1 };                   // the verifier can run the OOD constraint check against this AIR
    uint32_t W; int log_n; Col results;   // results[k] = b_k(n-1)
    uint32_t A = 0, R = 0;                // aux columns / aux random elements (A > 0 requires R > 0)
    uint32_t D = 2;                       // degree of the aux transition constraint: p' = p * (r + main)^(D-1), D in [2, 8]
    size_t n() const { return (size_t)1 << log_n; }
    size_t num_transition() const { return W + A; }
    size_t num_assertions() const { return W + W / 2 + A; }
    // constraint-evaluation blowup = number of composition columns = max(next_pow2(max constraint degree), 2)
    size_t ce_blowup() const { size_t d = A ? D : 1, e = 2; while (e < d) e <<= 1; return e; }
    size_t num_columns() const { return 3; }   // numerator columns = distinct divisors: transition, step 0, step n-1
    const Col& pub_elements() const { return results; }
    static Col seed(uint32_t k) { return Col{1 + 2 * (uint64_t)k, 2 + 2 * (uint64_t)k}; }
    // the model interface shared with ProgramAir (oracle/air.hpp); bodies below fib_eval_point
    template <class F> void build_aux(const std::vector<Col>& trace, const std::vector<typename F::T>& rands, std::vector<Col>& acols) const;
    template <class F, class CC> void eval_row(const CC& cc, const uint64_t* cur, const uint64_t* nxt, const typename F::T* acur,
                                               const typename F::T* anxt, const typename F::T* rands, uint64_t x, typename F::T* out) const;
    template <class F> void divide(const std::vector<std::vector<typename F::T>>& ce, std::vector<Col>& hcomp) const;
    template <class F, class CC> typename F::T ood_lhs(const CC& cc, const typename F::T* ood_cur, const typename F::T* ood_next,
                                                       const typename F::T* rands, typename F::T z) const;
};
// An AIR whose constraint system is not available (the golden Miden proof): no coefficient draws, no OOD check.
struct OpaqueAir {
    enum { KNOWN = 0 };
    uint32_t W = 0, A = 0, R = 0; int log_n = 0;
    size_t num_transition() const { return 0; }
    size_t num_assertions() const { return 0; }
    size_t ce_blowup() const { return 0; }
    template <class F, class CC> typename F::T ood_lhs(const CC&, const typename F::T*, const typename F::T*, const typename F::T*, typename F::T) const { return F::zero(); }
};
// Synthetic trace generator (column-major W x n). Pure function of (W, log_n).
static std::vector<Col> fib_trace(uint32_t W, int log_n) {
    size_t n = (size_t)1 << log_n;
    std::vector<Col> cols(W, Col(n));
    for (uint32_t k = 0; k < W / 2; k++) {
        uint64_t a = 1 + 2 * (uint64_t)k, b = 2 + 2 * (uint64_t)k;
        for (size_t i = 0; i < n; i++) {
            cols[2 * k][i] = a; cols[2 * k + 1][i] = b;
            uint64_t na = gl_add(a, b), nb = gl_add(b, na);
            a = na; b = nb;
        }
    }
    return cols;
}

// Constraint-combination bookkeeping for FibAir following winter-air 0.4 (restated):
//  * all W transition constraints have degree 1 -> one degree group, adjustment
//    adj_t = (ce_n - 1 + deg(divisor_t)) - (n - 1), divisor_t = (x^n - 1)/(x - w^(n-1)), deg = n - 1;
//  * assertions sorted by (stride = 0, first_step, column); one boundary group per (stride, first_step):
//    group 0 = step 0 (all W columns), group 1 = step n-1 (odd columns); divisor x - w^step, adjustment
//    adj_b = (ce_n - 1 + 1) - (n - 1). Coefficient pairs are consumed in that sorted order.
//  * aux transition constraints (degree D) form their own degree group: adj_x = (ce_n - 1 + (n - 1)) - D(n - 1);
//    coefficient pairs: main transition, aux transition, main assertions (sorted as above), aux assertions (step 0, by column).
struct FibCombine {
    uint64_t adj_t, adj_b, adj_x;
    template <class F> struct Coeffs { std::vector<typename F::T> ta, tb, ba, bb; };
    explicit FibCombine(const FibAir& air) {
        uint64_t n = air.n(), ce_n = n * air.ce_blowup();
        adj_t = (ce_n - 1 + (n - 1)) - (n - 1);
        adj_b = (ce_n - 1 + 1) - (n - 1);
        adj_x = (ce_n - 1 + (n - 1)) - (uint64_t)air.D * (n - 1);
    }
};
template <class F> static typename FibCombine::Coeffs<F> draw_constraint_coeffs(Coin& coin, size_t nt, size_t na) {
    // air_instance.cairo:115-142: (alpha, beta) pair per transition constraint, then per assertion.
    typename FibCombine::Coeffs<F> c;
    for (size_t i = 0; i < nt; i++) { c.ta.push_back(coin.draw<F>()); c.tb.push_back(coin.draw<F>()); }
    for (size_t i = 0; i < na; i++) { c.ba.push_back(coin.draw<F>()); c.bb.push_back(coin.draw<F>()); }
    return c;
}
// Numerators of the three constraint columns at one point x of the constraint-evaluation domain, given the
// frame (cur, next) of base-field trace values [evaluate_fragment, constraints_worker.rs:56-59].
// Generic over the frame element type so the verifier can reuse it at the OOD point (frame in E).
template <class F, class FV>
static void fib_eval_point(const FibAir& air, const FibCombine& cb, const typename FibCombine::Coeffs<F>& cc,
                           const typename FV::T* cur, const typename FV::T* nxt, typename FV::T x,
                           typename F::T out[3], const typename F::T* acur = nullptr, const typename F::T* anxt = nullptr,
                           const typename F::T* rands = nullptr) {
    // F is the coefficient field, FV the field of the frame values; products land in F (FV is F or the base field).
    auto lift = [](typename FV::T v) { uint64_t c[2] = {FV::comp(v, 0), FV::DEG > 1 ? FV::comp(v, 1) : 0}; return F::make(c); };
    typename F::T xt = f_pow<F>(lift(x), cb.adj_t), xb = f_pow<F>(lift(x), cb.adj_b);
    typename F::T acc = F::zero();
    for (uint32_t k = 0; k < air.W / 2; k++) {
        typename FV::T a = cur[2 * k], b = cur[2 * k + 1], na = nxt[2 * k], nb = nxt[2 * k + 1];
        typename FV::T t0 = FV::sub(na, FV::add(a, b));
        typename FV::T t1 = FV::sub(nb, FV::add(b, na));
        acc = F::add(acc, F::mul(F::add(cc.ta[2 * k], F::mul(cc.tb[2 * k], xt)), lift(t0)));
        acc = F::add(acc, F::mul(F::add(cc.ta[2 * k + 1], F::mul(cc.tb[2 * k + 1], xt)), lift(t1)));
    }
    if (air.A) {
        typename F::T xx = f_pow<F>(lift(x), cb.adj_x);
        for (uint32_t c = 0; c < air.A; c++) {
            typename F::T t = F::sub(anxt[c], F::mul(acur[c], f_pow<F>(F::add(rands[c % air.R], lift(cur[c % air.W])), air.D - 1)));
            acc = F::add(acc, F::mul(F::add(cc.ta[air.W + c], F::mul(cc.tb[air.W + c], xx)), t));
        }
    }
    out[0] = acc;
    typename F::T g0 = F::zero(), g1 = F::zero();
    for (uint32_t c = 0; c < air.A; c++) {
        size_t idx = air.W + air.W / 2 + c;
        g0 = F::add(g0, F::mul(F::add(cc.ba[idx], F::mul(cc.bb[idx], xb)), F::sub(acur[c], F::one())));
    }
    for (uint32_t c = 0; c < air.W; c++) {
        uint64_t v = FibAir::seed(c / 2)[c & 1];
        g0 = F::add(g0, F::mul(F::add(cc.ba[c], F::mul(cc.bb[c], xb)), lift(FV::sub(cur[c], FV::from(v)))));
    }
    for (uint32_t k = 0; k < air.W / 2; k++) {
        size_t idx = air.W + k;
        g1 = F::add(g1, F::mul(F::add(cc.ba[idx], F::mul(cc.bb[idx], xb)), lift(FV::sub(cur[2 * k + 1], FV::from(air.results[k])))));
    }
    out[1] = g0; out[2] = g1;
}

// ---- FibAir as a model (the AIR-specific steps of the prover and the verifier's OOD check) ----------------------------------
template <class F> void FibAir::build_aux(const std::vector<Col>& trace, const std::vector<typename F::T>& rands, std::vector<Col>& acols) const {
    typedef typename F::T T;
    const size_t n_ = n();
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t c = 0; c < A; c++) {
        T p = F::one();
        for (size_t i = 0; i < n_; i++) {
            for (int k = 0; k < F::DEG; k++) acols[c * F::DEG + k][i] = F::comp(p, k);
            p = F::mul(p, f_pow<F>(F::add(rands[c % R], F::from(trace[c % W][i])), D - 1));
        }
    }
}
template <class F, class CC> void FibAir::eval_row(const CC& cc, const uint64_t* cur, const uint64_t* nxt, const typename F::T* acur,
                                                    const typename F::T* anxt, const typename F::T* rands, uint64_t x, typename F::T* out) const {
    FibCombine cb(*this);
    fib_eval_point<F, FB>(*this, cb, cc, cur, nxt, x, out, acur, anxt, rands);
}
// `ConstraintEvaluationTable::into_poly`: divide each numerator column by its divisor on the constraint domain, sum -> H
template <class F> void FibAir::divide(const std::vector<std::vector<typename F::T>>& ce, std::vector<Col>& hcomp) const {
    typedef typename F::T T;
    const size_t n_ = n(), C = ce_blowup(), ceN = C * n_;
    const uint64_t g = gl_root_of_unity(log_n), gce = gl_root_of_unity(ilog2(ceN));
    const uint64_t wl = gl_pow(g, n_ - 1);
    // x^n takes only C distinct values on the ce domain: (7 w_ce^s)^n = 7^n * w_C^(s mod C)
    Col zinv(C);
    for (size_t k = 0; k < C; k++) zinv[k] = gl_inv(gl_sub(gl_mul(gl_pow(GEN, n_), gl_pow(gl_root_of_unity(ilog2(C)), k)), 1));
    const size_t CH = 1024;
#pragma omp parallel for schedule(static)
    for (size_t c0 = 0; c0 < ceN; c0 += CH) {
        size_t m = std::min(CH, ceN - c0);
        // batch-invert (x - 1) and (x - w^(n-1)) for the chunk
        Col d(2 * m), pre(2 * m);
        uint64_t x = gl_mul(GEN, gl_pow(gce, c0));
        Col xs(m);
        for (size_t i = 0; i < m; i++) { xs[i] = x; d[2 * i] = gl_sub(x, 1); d[2 * i + 1] = gl_sub(x, wl); x = gl_mul(x, gce); }
        uint64_t acc = 1;
        for (size_t i = 0; i < 2 * m; i++) { pre[i] = acc; acc = gl_mul(acc, d[i]); }
        uint64_t ia = gl_inv(acc);
        for (size_t i = 2 * m; i-- > 0;) { uint64_t inv = gl_mul(ia, pre[i]); ia = gl_mul(ia, d[i]); d[i] = inv; }
        for (size_t i = 0; i < m; i++) {
            size_t s = c0 + i;
            uint64_t tdiv = gl_mul(gl_sub(xs[i], wl), zinv[s % C]);    // 1 / ((x^n - 1)/(x - w^(n-1)))
            T h = F::mulb(ce[0][s], tdiv);
            h = F::add(h, F::mulb(ce[1][s], d[2 * i]));
            h = F::add(h, F::mulb(ce[2][s], d[2 * i + 1]));
            for (int k = 0; k < F::DEG; k++) hcomp[k][s] = F::comp(h, k);
        }
    }
}
// sum over the divisor groups of numerator(z) / divisor(z) at the out-of-domain point
template <class F, class CC> typename F::T FibAir::ood_lhs(const CC& cc, const typename F::T* ood_cur, const typename F::T* ood_next,
                                                            const typename F::T* rands, typename F::T z) const {
    typedef typename F::T T;
    FibCombine cb(*this);
    T num[3];
    fib_eval_point<F, F>(*this, cb, cc, ood_cur, ood_next, z, num, ood_cur + W, ood_next + W, rands);
    const uint64_t g = gl_root_of_unity(log_n);
    T zn = f_pow<F>(z, n()), wl = F::from(gl_pow(g, n() - 1));
    T lhs = F::mul(num[0], F::mul(F::sub(z, wl), F::inv(F::sub(zn, F::one()))));
    lhs = F::add(lhs, F::mul(num[1], F::inv(F::sub(z, F::one()))));
    lhs = F::add(lhs, F::mul(num[2], F::inv(F::sub(z, wl))));
    return lhs;
}

// ------------------------------------------------------------------------------------------------
// Verifier — stark_verifier.cairo:105-264 restated for any (W, aux, C, queries, fold, blowup) shape.
struct VerifyInfo {   // transcript values exposed for the golden-vector tests
    Digest coin_seed0, coin_seed, post_nonce_seed;
    std::vector<Digest> roots;
    uint64_t z[2] = {0, 0};
    std::vector<uint64_t> fri_alphas, positions, deep_evals;
    uint64_t lambda = 0, mu = 0;
    std::vector<std::pair<int, int>> batch_shapes;   // (vectors, total digests) per batch proof
};

template <class F, class M>
static void verify_impl(const Proof& pr, const Col& pub_elements, const M* air, VerifyInfo* info) {
    typedef typename F::T T;
    const size_t n = (size_t)1 << pr.log_n, B = pr.opt.blowup, N = n * B, Fd = pr.opt.fri_fold;
    const size_t W = pr.main_width, A = pr.aux_width, TW = W + A;
    const size_t EB = 8 * F::DEG;
    if (pr.opt.hash_fn != HASH_BLAKE2S_256) throw Err("verify: unsupported hash function");
    if (pr.modulus.size() != 8 || Coin::le64(pr.modulus.data()) != P) throw Err("verify: wrong field modulus");
    if (pr.ood_evaluations.size() % EB) throw Err("verify: bad OOD evaluations length");
    const size_t C = pr.ood_evaluations.size() / EB;
    const int layers = num_fri_layers(N, Fd, (uint64_t)1 << pr.opt.log_max_remainder);
    const size_t nroots = pr.num_segments() + 1 + layers + 1;
    if (pr.commitments.size() != 32 * nroots) throw Err("verify: wrong number of commitments");
    std::vector<Digest> roots(nroots);
    for (size_t i = 0; i < nroots; i++) memcpy(roots[i].b, pr.commitments.data() + 32 * i, 32);
    const uint64_t g = gl_root_of_unity(pr.log_n), gN = gl_root_of_unity(ilog2(N));

    // 0. coin (stark_verifier.cairo:83-91)
    Coin coin = Coin::from_pub_elements(pub_elements.data(), pub_elements.size());
    if (info) { info->coin_seed0 = hash_elements(pub_elements.data(), pub_elements.size()); info->coin_seed = coin.seed; info->roots = roots; }
    // 1. trace commitments (stark_verifier.cairo:117-130, 266-294)
    size_t ri = 0;
    coin.reseed(roots[ri++]);
    std::vector<T> aux_rands;
    if (A) { for (int i = 0; i < pr.aux_rands; i++) aux_rands.push_back(coin.draw<F>()); coin.reseed(roots[ri++]); }
    // constraint composition coefficients (air_instance.cairo:115-142). Draws never change the seed
    // (reseed resets the counter), so an opaque AIR may skip them.
    typename FibCombine::Coeffs<F> cc;
    if (M::KNOWN) cc = draw_constraint_coeffs<F>(coin, air->num_transition(), air->num_assertions());
    // 2. constraint commitment, z (stark_verifier.cairo:139-144)
    const Digest& croot = roots[ri++];
    coin.reseed(croot);
    T z = coin.draw<F>();
    // 3. OOD frame (stark_verifier.cairo:149-181, random.cairo:130-156)
    if (pr.ood_trace_states.size() != 2 * TW * EB) throw Err("verify: bad OOD frame length");
    std::vector<T> ood_cur(TW), ood_next(TW), ood_h(C);
    for (size_t i = 0; i < TW; i++) {
        uint64_t c0[2] = {0, 0}, c1[2] = {0, 0};
        for (int k = 0; k < F::DEG; k++) { c0[k] = Coin::le64(pr.ood_trace_states.data() + (i * F::DEG + k) * 8); c1[k] = Coin::le64(pr.ood_trace_states.data() + ((TW + i) * F::DEG + k) * 8); }
        for (int k = 0; k < F::DEG; k++) if (c0[k] >= P || c1[k] >= P) throw Err("verify: non-canonical element");
        ood_cur[i] = F::make(c0); ood_next[i] = F::make(c1);
    }
    for (size_t i = 0; i < C; i++) {
        uint64_t c0[2] = {0, 0};
        for (int k = 0; k < F::DEG; k++) { c0[k] = Coin::le64(pr.ood_evaluations.data() + (i * F::DEG + k) * 8); if (c0[k] >= P) throw Err("verify: non-canonical element"); }
        ood_h[i] = F::make(c0);
    }
    coin.reseed(f_hash<F>(ood_cur.data(), TW));
    coin.reseed(f_hash<F>(ood_next.data(), TW));
    coin.reseed(f_hash<F>(ood_h.data(), C));
    if (M::KNOWN) {
        // OOD consistency check (commented out in stark_verifier.cairo:151-159,183-187; winter-verifier does it):
        // sum_groups numerator(z)/divisor(z)  ==  sum_c z^c * H_c(z^C)     (reduce_evaluations, :296-304)
        if (W != air->W || A != air->A || (A && (size_t)pr.aux_rands != air->R) || C != air->ce_blowup() || pr.log_n != air->log_n)
            throw Err("verify: proof shape does not match the AIR");
        T lhs = air->template ood_lhs<F>(cc, ood_cur.data(), ood_next.data(), aux_rands.data(), z);
        T rhs = F::zero(), zp = F::one();
        for (size_t c = 0; c < C; c++) { rhs = F::add(rhs, F::mul(zp, ood_h[c])); zp = F::mul(zp, z); }
        if (!F::eq(lhs, rhs)) throw Err("verify: OOD constraint evaluations differ");
    }
    // 4. DEEP coefficients (air_instance.cairo:145-166): 3 per trace column, C, then (lambda, mu)
    std::vector<T> da(TW), db(TW), dg(TW), dc(C);
    for (size_t i = 0; i < TW; i++) { da[i] = coin.draw<F>(); db[i] = coin.draw<F>(); dg[i] = coin.draw<F>(); }
    for (size_t i = 0; i < C; i++) dc[i] = coin.draw<F>();
    T lambda = coin.draw<F>(), mu = coin.draw<F>();
    // FRI commit phase (fri_verifier.cairo:56-82): reseed with every FRI root (layers + remainder), draw alpha
    std::vector<T> alphas;
    for (int l = 0; l < layers + 1; l++) { coin.reseed(roots[ri + l]); alphas.push_back(coin.draw<F>()); }
    // 5. PoW + query positions (stark_verifier.cairo:204-221)
    coin.reseed_int(pr.pow_nonce);
    if (Coin::leading_zeros(coin.seed) < pr.opt.grinding) throw Err("verify: insufficient proof of work");
    if (info) info->post_nonce_seed = coin.seed;
    std::vector<uint64_t> pos = coin.draw_integers(pr.opt.num_queries, N);
    const size_t Q = pos.size();
    // trace + constraint queries (channel.cairo:206-267, 315-424) — ALL paths are authenticated here
    // (the Cairo code checks only the first 4: channel.cairo:345,410).
    std::vector<std::vector<uint64_t>> trows(Q, std::vector<uint64_t>());   // main || aux base-field values per query
    std::vector<std::vector<T>> trow_aux(Q);
    {
        size_t widths[2] = {W, A};
        for (int s = 0; s < pr.num_segments(); s++) {
            const Proof::Q& q = pr.trace_queries[s];
            size_t w = widths[s], eb = (s == 0) ? 8 : EB;   // main segment is base field; aux segment lives in E
            if (q.values.size() != Q * w * eb) throw Err("verify: bad trace query length");
            std::vector<Digest> leaves(Q);
            for (size_t i = 0; i < Q; i++) {
                Col row;
                for (size_t c = 0; c < w * eb / 8; c++) { uint64_t v = Coin::le64(q.values.data() + (i * w * eb / 8 + c) * 8); if (v >= P) throw Err("verify: non-canonical element"); row.push_back(v); }
                leaves[i] = hash_elements(row.data(), row.size());
                if (s == 0) trows[i] = row;
                else for (size_t c = 0; c < w; c++) trow_aux[i].push_back(F::make(row.data() + c * F::DEG));
            }
            auto nodes = batch_parse(q.paths.data(), q.paths.size());
            if (info) { int tot = 0; for (auto& v : nodes) tot += (int)v.size(); info->batch_shapes.push_back({(int)nodes.size(), tot}); }
            if (batch_root(pos, leaves, nodes, ilog2(N)) != roots[s]) throw Err("verify: trace query Merkle proof failed");
        }
    }
    std::vector<std::vector<T>> crows(Q, std::vector<T>(C));
    {
        const Proof::Q& q = pr.constraint_queries;
        if (q.values.size() != Q * C * EB) throw Err("verify: bad constraint query length");
        std::vector<Digest> leaves(Q);
        for (size_t i = 0; i < Q; i++) {
            Col row;
            for (size_t c = 0; c < C * F::DEG; c++) { uint64_t v = Coin::le64(q.values.data() + (i * C * F::DEG + c) * 8); if (v >= P) throw Err("verify: non-canonical element"); row.push_back(v); }
            leaves[i] = hash_elements(row.data(), row.size());
            for (size_t c = 0; c < C; c++) crows[i][c] = F::make(row.data() + c * F::DEG);
        }
        auto nodes = batch_parse(q.paths.data(), q.paths.size());
        if (info) { int tot = 0; for (auto& v : nodes) tot += (int)v.size(); info->batch_shapes.push_back({(int)nodes.size(), tot}); }
        if (batch_root(pos, leaves, nodes, ilog2(N)) != croot) throw Err("verify: constraint query Merkle proof failed");
    }
    // 6. DEEP composition at the queried positions (composer.cairo:17-316)
    T z_next = F::mulb(z, g), zC = f_pow<F>(z, C);
    T z_conj = F::conj(z);
    std::vector<T> evals(Q);
    for (size_t i = 0; i < Q; i++) {
        uint64_t x = gl_mul(GEN, gl_pow(gN, pos[i]));
        T xe = F::from(x);
        T s1 = F::zero(), s2 = F::zero(), s3 = F::zero();
        for (size_t c = 0; c < TW; c++) {
            T v = c < W ? F::from(trows[i][c]) : trow_aux[i][c - W];
            s1 = F::add(s1, F::mul(F::sub(v, ood_cur[c]), da[c]));
            s2 = F::add(s2, F::mul(F::sub(v, ood_next[c]), db[c]));
            // with the quadratic extension enabled, winter 0.4 adds the conjugate term for base-field
            // (main segment) columns: (T(x) - conj(T(z))) / (x - conj(z)) * gamma
            if (F::DEG > 1 && c < W) s3 = F::add(s3, F::mul(F::sub(v, F::conj(ood_cur[c])), dg[c]));
        }
        T t = F::add(F::mul(s1, F::inv(F::sub(xe, z))), F::mul(s2, F::inv(F::sub(xe, z_next))));
        if (F::DEG > 1) t = F::add(t, F::mul(s3, F::inv(F::sub(xe, z_conj))));
        T sc = F::zero();
        for (size_t c = 0; c < C; c++) sc = F::add(sc, F::mul(F::sub(crows[i][c], ood_h[c]), dc[c]));
        T cpart = F::mul(sc, F::inv(F::sub(xe, zC)));
        evals[i] = F::mul(F::add(t, cpart), F::add(lambda, F::mulb(mu, x)));
    }
    if (info) {
        info->z[0] = F::comp(z, 0); info->z[1] = F::DEG > 1 ? F::comp(z, 1) : 0;
        for (auto& a : alphas) info->fri_alphas.push_back(F::comp(a, 0));
        info->lambda = F::comp(lambda, 0); info->mu = F::comp(mu, 0);
        info->positions = pos;
        for (auto& e : evals) info->deep_evals.push_back(F::comp(e, 0));
    }
    // 7. FRI (fri_verifier.cairo:243-451). Offset stays 7 at every layer (fri_verifier.cairo:23,308).
    if ((int)pr.fri_layers.size() != layers) throw Err("verify: wrong number of FRI layers");
    std::vector<uint64_t> cur_pos = pos;
    std::vector<T> cur_eval = evals;
    uint64_t dom = N, omega = gN;
    for (int l = 0; l < layers; l++) {
        uint64_t rows = dom / Fd;
        std::vector<uint64_t> fpos = fold_positions(cur_pos, dom, Fd);
        const Proof::Q& q = pr.fri_layers[l];
        if (q.values.size() != fpos.size() * Fd * EB) throw Err("verify: bad FRI layer values length");
        std::vector<std::vector<T>> vals(fpos.size(), std::vector<T>(Fd));
        std::vector<Digest> leaves(fpos.size());
        for (size_t k = 0; k < fpos.size(); k++) {
            Col row;
            for (size_t c = 0; c < Fd * F::DEG; c++) { uint64_t v = Coin::le64(q.values.data() + (k * Fd * F::DEG + c) * 8); if (v >= P) throw Err("verify: non-canonical element"); row.push_back(v); }
            leaves[k] = hash_elements(row.data(), row.size());
            for (size_t j = 0; j < Fd; j++) vals[k][j] = F::make(row.data() + j * F::DEG);
        }
        auto nodes = batch_parse(q.paths.data(), q.paths.size());
        if (info) { int tot = 0; for (auto& v : nodes) tot += (int)v.size(); info->batch_shapes.push_back({(int)nodes.size(), tot}); }
        if (batch_root(fpos, leaves, nodes, ilog2(rows)) != roots[ri + l]) throw Err("verify: FRI layer Merkle proof failed");
        // consistency of the incoming evaluations with the opened rows (fri_verifier.cairo:298-303)
        for (size_t i = 0; i < cur_pos.size(); i++) {
            uint64_t fp = cur_pos[i] % rows, jj = cur_pos[i] / rows;
            size_t k = std::find(fpos.begin(), fpos.end(), fp) - fpos.begin();
            if (!F::eq(vals[k][jj], cur_eval[i])) throw Err("verify: FRI layer value mismatch");
        }
        // fold each opened row (fri_verifier.cairo:305-315, compute_folding_roots :218-228)
        std::vector<T> nxt(fpos.size());
        uint64_t wF = gl_pow(omega, rows);
        for (size_t k = 0; k < fpos.size(); k++) {
            uint64_t xe = gl_mul(GEN, gl_pow(omega, fpos[k]));
            std::vector<uint64_t> xs(Fd);
            uint64_t r = 1;
            for (size_t j = 0; j < Fd; j++) { xs[j] = gl_mul(xe, r); r = gl_mul(r, wF); }
            nxt[k] = lagrange_eval<F>(xs.data(), vals[k].data(), (int)Fd, alphas[l]);
        }
        cur_pos = fpos; cur_eval = nxt; dom = rows; omega = gl_pow(omega, Fd);
    }
    // remainder (channel.cairo:80-100, fri_verifier.cairo:261-265)
    if (pr.fri_remainder.size() != dom * EB) throw Err("verify: bad remainder length");
    std::vector<T> rem(dom);
    for (size_t i = 0; i < dom; i++) {
        uint64_t c0[2] = {0, 0};
        for (int k = 0; k < F::DEG; k++) { c0[k] = Coin::le64(pr.fri_remainder.data() + (i * F::DEG + k) * 8); if (c0[k] >= P) throw Err("verify: non-canonical element"); }
        rem[i] = F::make(c0);
    }
    for (size_t i = 0; i < cur_pos.size(); i++) if (!F::eq(rem[cur_pos[i]], cur_eval[i])) throw Err("verify: remainder value mismatch");
    {
        size_t rows = dom / Fd;
        std::vector<Digest> leaves(rows);
        for (size_t i = 0; i < rows; i++) {
            std::vector<T> row(Fd);
            for (size_t j = 0; j < Fd; j++) row[j] = rem[i + j * rows];
            leaves[i] = f_hash<F>(row.data(), Fd);
        }
        Digest rr = rows >= 2 ? MerkleTree(leaves).root() : leaves[0];
        if (rr != roots[ri + layers]) throw Err("verify: remainder commitment mismatch");
    }
    {   // remainder degree (winter-fri 0.4 verify_remainder): interpolate over <omega>, degree <= n / Fd^layers - 1
        size_t maxdeg_plus1 = n; for (int l = 0; l < layers; l++) maxdeg_plus1 /= Fd;
        if (maxdeg_plus1 == 0) maxdeg_plus1 = 1;   // folded past degree 0 (large blowup, small remainder): the remainder is a constant
        if (maxdeg_plus1 >= dom) throw Err("verify: remainder degree bound not valid");
        for (int k = 0; k < F::DEG; k++) {
            Col comp(dom);
            for (size_t i = 0; i < dom; i++) comp[i] = F::comp(rem[i], k);
            intt(comp.data(), dom);
            for (size_t i = maxdeg_plus1; i < dom; i++) if (comp[i] != 0) throw Err("verify: remainder degree too high");
        }
    }
}

template <class M> static void verify_with(const Bytes& proof_bytes, const Col& pub_elements, const M* air, VerifyInfo* info = nullptr) {
    Proof pr = Proof::parse(proof_bytes.data(), proof_bytes.size());
    if (pr.opt.field_ext == EXT_NONE) verify_impl<FB, M>(pr, pub_elements, air, info);
    else if (pr.opt.field_ext == EXT_QUADRATIC) verify_impl<FQ, M>(pr, pub_elements, air, info);
    else throw Err("verify: unsupported field extension");
}
static void verify(const Bytes& proof_bytes, const Col& pub_elements, AirKind kind, const FibAir* fib, VerifyInfo* info = nullptr) {
    OpaqueAir opaque;
    if (kind == AIR_FIB) verify_with<FibAir>(proof_bytes, pub_elements, fib, info);
    else verify_with<OpaqueAir>(proof_bytes, pub_elements, &opaque, info);
}

}  // namespace orc
