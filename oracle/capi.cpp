// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/gl.hpp header).
// Flat C entry points over the CPU restatement so tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg can call it through ctypes. Built into oracle/liboracle.so by oracle/Makefile. Nothing in aero_amd/ links
// or loads this library.
#include <omp.h>
#include <cstring>
#include <sstream>
#include "prover.hpp"

using namespace orc;

static thread_local std::string g_err;
static int fail(const std::exception& e) { g_err = e.what(); return -1; }
static std::string hexd(const Digest& d) { char b[65]; for (int i = 0; i < 32; i++) sprintf(b + 2 * i, "%02x", d.b[i]); return b; }

static ProverArtifacts<FB> g_art;   // intermediates of the last orc_prove_fib(..., keep_artifacts=1), base field
static ProverArtifacts<FQ> g_art_q; // same for the quadratic extension (only the artifacts listed in orc_artifact)
static bool g_art_is_q = false;

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }
void orc_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int orc_max_threads() { return omp_get_max_threads(); }
void orc_free(void* p) { free(p); }

// ---- field -------------------------------------------------------------------------------------
uint64_t orc_gl_add(uint64_t a, uint64_t b) { return gl_add(a, b); }
uint64_t orc_gl_sub(uint64_t a, uint64_t b) { return gl_sub(a, b); }
uint64_t orc_gl_mul(uint64_t a, uint64_t b) { return gl_mul(a, b); }
uint64_t orc_gl_mul_slow(uint64_t a, uint64_t b) { return gl_mul_slow(a, b); }
uint64_t orc_gl_inv(uint64_t a) { return gl_inv(a); }
uint64_t orc_gl_pow(uint64_t a, uint64_t e) { return gl_pow(a, e); }
uint64_t orc_gl_root_of_unity(int log_n) { return gl_root_of_unity(log_n); }
void orc_e2_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) { Fe2 r = e2_mul(Fe2{a[0], a[1]}, Fe2{b[0], b[1]}); out[0] = r.a0; out[1] = r.a1; }
void orc_e2_inv(const uint64_t a[2], uint64_t out[2]) { Fe2 r = e2_inv(Fe2{a[0], a[1]}); out[0] = r.a0; out[1] = r.a1; }

// ---- hashing / coin ------------------------------------------------------------------------------
void orc_blake2s(const uint8_t* data, size_t len, uint8_t out[32]) { Digest d = blake2s(data, len); memcpy(out, d.b, 32); }
void orc_hash_elements(const uint64_t* e, size_t n, uint8_t out[32]) { Digest d = hash_elements(e, n); memcpy(out, d.b, 32); }
// column-major (W columns of `rows` elements, column c at cols + c*rows) -> rows x 32 bytes
void orc_hash_rows(const uint64_t* cols, uint32_t W, size_t rows, uint8_t* out) {
    std::vector<Col> m(W);
    for (uint32_t c = 0; c < W; c++) m[c].assign(cols + (size_t)c * rows, cols + (size_t)(c + 1) * rows);
    auto d = hash_rows(m, rows);
    memcpy(out, d.data(), rows * 32);
}
// leaves (n x 32 B) -> all 2n node slots (slot 0 unused/zero, slot 1 = root, slots n.. = leaves)
int orc_merkle_nodes(const uint8_t* leaves, size_t n, uint8_t* nodes_out) {
    try {
        std::vector<Digest> l(n); memcpy(l.data(), leaves, n * 32);
        MerkleTree t(l);
        memcpy(nodes_out, t.t.data(), 2 * n * 32); memset(nodes_out, 0, 32);
        return 0;
    } catch (std::exception& e) { return fail(e); }
}
// batch opening bytes (u8 V, per vector u8 len + digests) for positions against a tree over `leaves`
int orc_batch_proof(const uint8_t* leaves, size_t n, const uint64_t* pos, size_t k, uint8_t* out, size_t cap, size_t* out_len) {
    try {
        std::vector<Digest> l(n); memcpy(l.data(), leaves, n * 32);
        MerkleTree t(l);
        Bytes b = batch_serialize(batch_prove(t, std::vector<uint64_t>(pos, pos + k)));
        *out_len = b.size();
        if (b.size() > cap) throw Err("buffer too small");
        memcpy(out, b.data(), b.size());
        return 0;
    } catch (std::exception& e) { return fail(e); }
}
// coin: state = 32-byte seed + counter, all ops explicit so host-side transcripts can be cross-checked
void orc_coin_new(const uint64_t* pub, size_t n, uint8_t seed[32]) { Coin c = Coin::from_pub_elements(pub, n); memcpy(seed, c.seed.b, 32); }
void orc_coin_reseed(uint8_t seed[32], const uint8_t d[32]) { Coin c; memcpy(c.seed.b, seed, 32); Digest x; memcpy(x.b, d, 32); c.reseed(x); memcpy(seed, c.seed.b, 32); }
void orc_coin_reseed_int(uint8_t seed[32], uint64_t v) { Coin c; memcpy(c.seed.b, seed, 32); c.reseed_int(v); memcpy(seed, c.seed.b, 32); }
int orc_coin_draw(const uint8_t seed[32], uint64_t* ctr, uint64_t* out) {
    try { Coin c; memcpy(c.seed.b, seed, 32); c.ctr = *ctr; *out = c.draw_base(); *ctr = c.ctr; return 0; } catch (std::exception& e) { return fail(e); }
}
int orc_coin_draw_integers(const uint8_t seed[32], uint64_t* ctr, size_t k, uint64_t domain, uint64_t* out) {
    try { Coin c; memcpy(c.seed.b, seed, 32); c.ctr = *ctr; auto v = c.draw_integers(k, domain); memcpy(out, v.data(), k * 8); *ctr = c.ctr; return 0; } catch (std::exception& e) { return fail(e); }
}
uint32_t orc_leading_zeros(const uint8_t seed[32], uint64_t nonce) { Coin c; memcpy(c.seed.b, seed, 32); return c.check_leading_zeros(nonce); }

// ---- polynomial stages -----------------------------------------------------------------------------
void orc_intt(uint64_t* a, size_t n) { intt(a, n, true); }
void orc_lde(const uint64_t* coeffs, size_t n, size_t blowup, uint64_t* out) { Col r = lde(coeffs, n, blowup, GEN); memcpy(out, r.data(), r.size() * 8); }
void orc_fib_trace(uint32_t W, int log_n, uint64_t* out) {
    auto t = fib_trace(W, log_n); size_t n = (size_t)1 << log_n;
    for (uint32_t c = 0; c < W; c++) memcpy(out + (size_t)c * n, t[c].data(), n * 8);
}
// one FRI fold (base field): values (dom) -> next (dom/fold), x-offset stays 7, domain generator w_dom
void orc_fri_fold(const uint64_t* values, size_t dom, uint32_t fold, uint64_t alpha, uint64_t* out) {
    size_t rows = dom / fold; uint64_t omega = gl_root_of_unity(ilog2(dom)), wF = gl_pow(omega, rows);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < rows; i++) {
        uint64_t xs[16], ys[16], xe = gl_mul(GEN, gl_pow(omega, i)), r = 1;
        for (uint32_t j = 0; j < fold; j++) { xs[j] = gl_mul(xe, r); r = gl_mul(r, wF); ys[j] = values[i + j * rows]; }
        out[i] = lagrange_eval<FB>(xs, ys, (int)fold, alpha);
    }
}

// ---- containers --------------------------------------------------------------------------------------
int orc_container_split(const uint8_t* data, size_t len, size_t* in_off, size_t* in_len, size_t* pf_off, size_t* pf_len) {
    try {
        Bytes a, b; container_split(data, len, a, b);
        *in_off = 8; *in_len = a.size(); *pf_off = 16 + a.size(); *pf_len = b.size();
        return 0;
    } catch (std::exception& e) { return fail(e); }
}
int orc_miden_pub_elements(const uint8_t* in, size_t len, uint64_t* out, size_t cap, size_t* n) {
    try {
        Col e = miden_pub_elements(Bytes(in, in + len)); *n = e.size();
        if (e.size() > cap) throw Err("buffer too small");
        memcpy(out, e.data(), e.size() * 8); return 0;
    } catch (std::exception& e) { return fail(e); }
}
// parse + re-serialise (layout round trip)
int orc_proof_roundtrip(const uint8_t* proof, size_t len, uint8_t* out, size_t cap, size_t* out_len) {
    try {
        Bytes b = Proof::parse(proof, len).to_bytes(); *out_len = b.size();
        if (b.size() > cap) throw Err("buffer too small");
        memcpy(out, b.data(), b.size()); return 0;
    } catch (std::exception& e) { return fail(e); }
}

// ---- verifier ------------------------------------------------------------------------------------------
// air_kind 0 = opaque (no OOD constraint check; what src/stark_verifier does), 1 = FibAir(W, log_n) with
// pub = results. If info/info_cap are given, a "key=value\n" dump of the transcript is written (golden tests).
int orc_verify(const uint8_t* proof, size_t len, const uint64_t* pub, size_t npub, int air_kind, uint32_t W, int log_n,
               char* info, size_t info_cap) {
    try {
        Col pe(pub, pub + npub);
        FibAir air; air.W = W; air.log_n = log_n; air.results = pe;
        VerifyInfo vi;
        verify(Bytes(proof, proof + len), pe, (AirKind)air_kind, air_kind == AIR_FIB ? &air : nullptr, &vi);
        if (info && info_cap) {
            std::ostringstream o;
            o << "coin_seed0=" << hexd(vi.coin_seed0) << "\ncoin_seed=" << hexd(vi.coin_seed) << "\npost_nonce_seed=" << hexd(vi.post_nonce_seed) << "\n";
            o << "roots="; for (size_t i = 0; i < vi.roots.size(); i++) o << (i ? "," : "") << hexd(vi.roots[i]); o << "\n";
            o << "z=" << vi.z[0] << "," << vi.z[1] << "\nlambda=" << vi.lambda << "\nmu=" << vi.mu << "\n";
            o << "fri_alphas="; for (size_t i = 0; i < vi.fri_alphas.size(); i++) o << (i ? "," : "") << vi.fri_alphas[i]; o << "\n";
            o << "positions="; for (size_t i = 0; i < vi.positions.size(); i++) o << (i ? "," : "") << vi.positions[i]; o << "\n";
            o << "deep="; for (size_t i = 0; i < vi.deep_evals.size(); i++) o << (i ? "," : "") << vi.deep_evals[i]; o << "\n";
            o << "batch_shapes="; for (size_t i = 0; i < vi.batch_shapes.size(); i++) o << (i ? ";" : "") << vi.batch_shapes[i].first << ":" << vi.batch_shapes[i].second; o << "\n";
            std::string s = o.str();
            if (s.size() + 1 > info_cap) throw Err("info buffer too small");
            memcpy(info, s.c_str(), s.size() + 1);
        }
        return 0;
    } catch (std::exception& e) { return fail(e); }
}

// ---- prover --------------------------------------------------------------------------------------------
// trace: column-major W x 2^log_n (or NULL to use the synthetic FibAir trace). opt: the 7 option bytes.
// Returns malloc'd proof (orc_free). pub_out receives W/2 results. times receives 12 doubles (StageTimes order).
int orc_prove_fib(const uint64_t* trace, uint32_t W, int log_n, const uint8_t opt7[7], uint8_t** proof, size_t* proof_len,
                  uint64_t* pub_out, double* times, int keep_artifacts) {
    try {
        size_t n = (size_t)1 << log_n;
        std::vector<Col> tr;
        if (trace) { tr.resize(W); for (uint32_t c = 0; c < W; c++) tr[c].assign(trace + (size_t)c * n, trace + (size_t)(c + 1) * n); }
        else tr = fib_trace(W, log_n);
        Options o{opt7[0], opt7[1], opt7[2], opt7[3], opt7[4], opt7[5], opt7[6]};
        Col pub; StageTimes tm; Bytes pf;
        if (keep_artifacts && o.field_ext == EXT_NONE) { g_art = ProverArtifacts<FB>(); g_art_is_q = false; pf = prove_fib<FB>(tr, log_n, o, &pub, &tm, &g_art); }
        else if (keep_artifacts && o.field_ext == EXT_QUADRATIC) { g_art_q = ProverArtifacts<FQ>(); g_art_is_q = true; pf = prove_fib<FQ>(tr, log_n, o, &pub, &tm, &g_art_q); }
        else pf = prove_fib_any(tr, log_n, o, &pub, &tm);
        *proof = (uint8_t*)malloc(pf.size()); memcpy(*proof, pf.data(), pf.size()); *proof_len = pf.size();
        if (pub_out) memcpy(pub_out, pub.data(), pub.size() * 8);
        if (times) { double t[12] = {tm.interpolate, tm.lde, tm.trace_commit, tm.constraints, tm.composition, tm.comp_commit, tm.ood, tm.deep, tm.fri, tm.grind, tm.queries, tm.total}; memcpy(times, t, sizeof t); }
        return 0;
    } catch (std::exception& e) { return fail(e); }
}
// FibAir with an auxiliary segment of A columns built from R coin elements (oracle/stark.hpp: FibAir::A, R).
int orc_prove_fib_aux(const uint64_t* trace, uint32_t W, int log_n, uint32_t A, uint32_t R, uint32_t D, const uint8_t opt7[7], uint8_t** proof,
                      size_t* proof_len, uint64_t* pub_out, double* times, int keep_artifacts) {
    try {
        size_t n = (size_t)1 << log_n;
        std::vector<Col> tr;
        if (trace) { tr.resize(W); for (uint32_t c = 0; c < W; c++) tr[c].assign(trace + (size_t)c * n, trace + (size_t)(c + 1) * n); }
        else tr = fib_trace(W, log_n);
        Options o{opt7[0], opt7[1], opt7[2], opt7[3], opt7[4], opt7[5], opt7[6]};
        Col pub; StageTimes tm;
        Bytes pf;
        if (keep_artifacts && o.field_ext == EXT_NONE) { g_art = ProverArtifacts<FB>(); g_art_is_q = false; pf = prove_fib<FB>(tr, log_n, o, &pub, &tm, &g_art, A, R, D); }
        else if (keep_artifacts && o.field_ext == EXT_QUADRATIC) { g_art_q = ProverArtifacts<FQ>(); g_art_is_q = true; pf = prove_fib<FQ>(tr, log_n, o, &pub, &tm, &g_art_q, A, R, D); }
        else pf = prove_fib_any(tr, log_n, o, &pub, &tm, A, R, D);
        *proof = (uint8_t*)malloc(pf.size()); memcpy(*proof, pf.data(), pf.size()); *proof_len = pf.size();
        if (pub_out) memcpy(pub_out, pub.data(), pub.size() * 8);
        if (times) { double t[12] = {tm.interpolate, tm.lde, tm.trace_commit, tm.constraints, tm.composition, tm.comp_commit, tm.ood, tm.deep, tm.fri, tm.grind, tm.queries, tm.total}; memcpy(times, t, sizeof t); }
        return 0;
    } catch (std::exception& e) { return fail(e); }
}
// Verify with the OOD constraint check for FibAir(W) + aux segment (A, R).
int orc_verify_fib_aux(const uint8_t* proof, size_t len, const uint64_t* pub, size_t npub, uint32_t W, int log_n, uint32_t A, uint32_t R, uint32_t D) {
    try {
        Col pe(pub, pub + npub);
        FibAir air; air.W = W; air.log_n = log_n; air.results = pe; air.A = A; air.R = A ? R : 0; air.D = D;
        verify(Bytes(proof, proof + len), pe, AIR_FIB, &air, nullptr);
        return 0;
    } catch (std::exception& e) { return fail(e); }
}
// ---- AIR-as-data (oracle/air.hpp): prove / verify against an AEROAIR program ------------------------------------------------
static ProgramAir load_air(const uint8_t* program, size_t plen, int log_n, const uint64_t* pub, size_t npub) {
    ProgramAir air = ProgramAir::parse(program, plen);
    air.bind(log_n, Col(pub, pub + npub));
    return air;
}
// trace: column-major W x 2^log_n; pub: the program's public inputs (they seed the coin)
int orc_prove_air(const uint8_t* program, size_t plen, const uint64_t* trace, uint32_t W, int log_n, const uint64_t* pub, size_t npub,
                  const uint8_t opt7[7], uint8_t** proof, size_t* proof_len, double* times, int keep_artifacts) {
    try {
        const size_t n = (size_t)1 << log_n;
        ProgramAir air = load_air(program, plen, log_n, pub, npub);
        if (W != air.W) throw Err("prove_air: trace width does not match the program");
        std::vector<Col> tr(W);
        for (uint32_t c = 0; c < W; c++) tr[c].assign(trace + (size_t)c * n, trace + (size_t)(c + 1) * n);
        Options o{opt7[0], opt7[1], opt7[2], opt7[3], opt7[4], opt7[5], opt7[6]};
        StageTimes tm; Bytes pf;
        if (o.field_ext == EXT_NONE) {
            if (keep_artifacts) { g_art = ProverArtifacts<FB>(); g_art_is_q = false; }
            pf = prove_model<FB, ProgramAir>(air, tr, log_n, o, &tm, keep_artifacts ? &g_art : nullptr);
        } else if (o.field_ext == EXT_QUADRATIC) {
            if (keep_artifacts) { g_art_q = ProverArtifacts<FQ>(); g_art_is_q = true; }
            pf = prove_model<FQ, ProgramAir>(air, tr, log_n, o, &tm, keep_artifacts ? &g_art_q : nullptr);
        } else throw Err("prove_air: unsupported field extension");
        *proof = (uint8_t*)malloc(pf.size()); memcpy(*proof, pf.data(), pf.size()); *proof_len = pf.size();
        if (times) { double t[12] = {tm.interpolate, tm.lde, tm.trace_commit, tm.constraints, tm.composition, tm.comp_commit, tm.ood, tm.deep, tm.fri, tm.grind, tm.queries, tm.total}; memcpy(times, t, sizeof t); }
        return 0;
    } catch (std::exception& e) { return fail(e); }
}
// full verification incl. the OOD constraint check evaluated from the program
int orc_verify_air(const uint8_t* proof, size_t len, const uint8_t* program, size_t plen, const uint64_t* pub, size_t npub, int log_n) {
    try {
        ProgramAir air = load_air(program, plen, log_n, pub, npub);
        verify_with<ProgramAir>(Bytes(proof, proof + len), air.pub, &air, nullptr);
        return 0;
    } catch (std::exception& e) { return fail(e); }
}
// out = { ce_blowup, numerator columns, transition constraints, assertions, main width, aux width, aux rands, nodes }
int orc_air_info(const uint8_t* program, size_t plen, int log_n, uint64_t out[8]) {
    try {
        ProgramAir air = ProgramAir::parse(program, plen);
        air.bind(log_n, Col(air.num_pub, 0));
        const uint64_t v[8] = {air.ce_blowup(), air.num_columns(), air.num_transition(), air.num_assertions(), air.W, air.A, air.R, air.nodes.size()};
        memcpy(out, v, sizeof v);
        return 0;
    } catch (std::exception& e) { return fail(e); }
}
// 0 when the main trace satisfies the program's main transition constraints (test helper for hand-made traces)
int orc_air_check_trace(const uint8_t* program, size_t plen, const uint64_t* trace, uint32_t W, int log_n, const uint64_t* pub, size_t npub) {
    try {
        const size_t n = (size_t)1 << log_n;
        ProgramAir air = load_air(program, plen, log_n, pub, npub);
        if (W != air.W) throw Err("check_trace: width mismatch");
        std::vector<Col> tr(W);
        for (uint32_t c = 0; c < W; c++) tr[c].assign(trace + (size_t)c * n, trace + (size_t)(c + 1) * n);
        const std::string why = air.check_main_trace(tr);
        if (!why.empty()) throw Err(why);
        return 0;
    } catch (std::exception& e) { return fail(e); }
}
// Fetch an intermediate of the last keep_artifacts prove. Column-major for matrices. Returns element count
// (u64 units; digests count 4 per digest) or -1.
long orc_artifact(const char* name, uint64_t* out, size_t cap) {
    std::string s(name);
    Col flat;
    auto put_cols = [&](const std::vector<Col>& m) { for (auto& c : m) flat.insert(flat.end(), c.begin(), c.end()); };
    if (g_art_is_q) {
        // quadratic extension: E-valued columns come out as component columns, column c * 2 + d (the layout of the
        // product's aero_eval_constraints_fib)
        if (s == "ce_cols") { for (auto& col : g_art_q.ce_cols) for (int d = 0; d < 2; d++) for (auto& v : col) flat.push_back(FQ::comp(v, d)); }
        else if (s == "cons_coeffs") flat = g_art_q.cons_coeffs;
        else if (s == "aux_rands") flat = g_art_q.aux_rands;
        else if (s == "aux_cols") put_cols(g_art_q.aux_cols);
        else if (s == "aux_lde") put_cols(g_art_q.aux_lde);
        else if (s == "trace_lde") put_cols(g_art_q.trace_lde);
        else if (s == "comp_polys") put_cols(g_art_q.comp_polys);
        else if (s == "comp_lde") put_cols(g_art_q.comp_lde);
        else return -1;
        if (flat.size() > cap) return -(long)flat.size() - 2;
        memcpy(out, flat.data(), flat.size() * 8);
        return (long)flat.size();
    }
    if (s == "trace_polys") put_cols(g_art.trace_polys);
    else if (s == "trace_lde") put_cols(g_art.trace_lde);
    else if (s == "trace_leaves") { flat.resize(g_art.trace_leaves.size() * 4); memcpy(flat.data(), g_art.trace_leaves.data(), flat.size() * 8); }
    else if (s == "ce_cols") put_cols(g_art.ce_cols);
    else if (s == "comp_polys") put_cols(g_art.comp_polys);
    else if (s == "comp_lde") put_cols(g_art.comp_lde);
    else if (s == "deep") flat = g_art.deep;
    else if (s.rfind("fri_layer_", 0) == 0) { size_t l = atoi(s.c_str() + 10); if (l >= g_art.fri_layers.size()) return -1; flat = g_art.fri_layers[l]; }
    else if (s == "ood_cur") flat = g_art.ood_cur;
    else if (s == "ood_next") flat = g_art.ood_next;
    else if (s == "ood_h") flat = g_art.ood_h;
    else if (s == "cons_coeffs") flat = g_art.cons_coeffs;
    else if (s == "aux_rands") flat = g_art.aux_rands;
    else if (s == "aux_cols") put_cols(g_art.aux_cols);
    else if (s == "aux_lde") put_cols(g_art.aux_lde);
    else return -1;
    if (flat.size() > cap) return -(long)flat.size() - 2;
    memcpy(out, flat.data(), flat.size() * 8);
    return (long)flat.size();
}

}  // extern "C"
