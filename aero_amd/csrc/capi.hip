// extern "C" boundary of libaero_stark.so (declarations + reference citations: include/aero_stark.h).
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

#include <atomic>
#include "capi_internal.hpp"
#include "../../include/aero_air.h"
#include "proof_format.hpp"
#include "stark_kernels.hpp"
#include "worker_messages.hpp"

thread_local std::string g_create_err;

namespace aero { Context* ctx_of(aero_ctx* c) { return c ? c->c : nullptr; } }   // for comm_rccl.hip

extern "C" {

int32_t aero_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int32_t aero_ctx_create(int32_t device_id, aero_ctx** out) {
    if (!out) return AERO_E_BAD_ARG;
    *out = nullptr;
    try {
        aero_ctx* h = new aero_ctx();
        try { h->keep = std::make_shared<Context>(device_id); h->c = h->keep.get(); } catch (...) { delete h; throw; }
        if (const char* e = getenv("AERO_SELF_VERIFY")) if (e[0] == '0' || e[0] == '1') h->self_verify = e[0] - '0';
        *out = h;
        return AERO_OK;
    } catch (const Error& e) { g_create_err = e.what(); return e.code; }
    catch (const std::exception& e) { g_create_err = e.what(); return AERO_E_INTERNAL; }
}
void aero_ctx_destroy(aero_ctx* ctx) {
    if (!ctx) return;
    delete ctx;   // the Context itself dies with the last matrix / tree that still references it
}
int32_t aero_ctx_set_self_verify(aero_ctx* ctx, int32_t mode) {
    if (!ctx || mode < AERO_SELF_VERIFY_AUTO || mode > AERO_SELF_VERIFY_ON) return AERO_E_BAD_ARG;
    ctx->self_verify = mode;
    return AERO_OK;
}
int32_t aero_selftest(aero_ctx* ctx, uint32_t samples, uint64_t seed) {
    return guard(ctx, [&] {
        REQUIRE(samples >= 64 && samples <= (1u << 22), "selftest: samples must be in [64, 2^22]");
        field_selftest(ctx->c, samples, seed);
    });
}
int32_t aero_ctx_synchronize(aero_ctx* ctx) {
    return guard(ctx, [&] { ctx->c->sync(); });
}
const char* aero_last_error(const aero_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }
void aero_free(void* p) { free(p); }

// ---- matrices ----------------------------------------------------------------------------------
int32_t aero_trace_upload(aero_ctx* ctx, const uint64_t* col_major, uint32_t width, uint32_t log_n, aero_matrix** out) {
    return guard(ctx, [&] {
        REQUIRE(col_major && out, "trace_upload: null argument");
        REQUIRE(width >= 1 && width <= 255 && log_n >= 3 && log_n <= 29, "trace_upload: width must be in [1,255] and log_n in [3,29]");
        size_t n = (size_t)1 << log_n;
        aero_matrix* h = new aero_matrix(ctx);
        try {
            h->m = Matrix(ctx->c, (int)width, n);
            AERO_HIP(hipMemcpyAsync(h->m.data.get(), col_major, (size_t)width * n * 8, hipMemcpyHostToDevice, ctx->c->stream));
            if (!all_canonical(ctx->c, h->m.data.get(), (size_t)width * n)) fail("trace_upload: trace holds a non-canonical field element (>= p)");
        } catch (...) { delete h; throw; }
        *out = h;
    });
}
int32_t aero_matrix_shape(const aero_matrix* m, uint32_t* cols, uint64_t* rows) {
    if (!m) return AERO_E_BAD_ARG;
    if (cols) *cols = (uint32_t)m->m.cols;
    if (rows) *rows = m->m.rows;
    return AERO_OK;
}
int32_t aero_matrix_device_ptr(const aero_matrix* m, uint64_t** dev_ptr_out) {
    if (!m || !dev_ptr_out) return AERO_E_BAD_ARG;
    *dev_ptr_out = m->m.data.get();
    return AERO_OK;
}
int32_t aero_matrix_download(aero_ctx* ctx, const aero_matrix* m, uint64_t* out) {
    return guard(ctx, [&] {
        REQUIRE(m && out, "matrix_download: null argument");
        AERO_HIP(hipMemcpyAsync(out, m->m.data.get(), (size_t)m->m.cols * m->m.rows * 8, hipMemcpyDeviceToHost, ctx->c->stream));
        ctx->c->sync();
    });
}
void aero_matrix_free(aero_ctx* ctx, aero_matrix* m) {
    (void)ctx;
    delete m;
}
int32_t aero_fib_trace(uint32_t width, uint32_t log_n, uint64_t* out) {
    if (!out || width < 2 || (width & 1) || log_n > 29) return AERO_E_BAD_ARG;
    size_t n = (size_t)1 << log_n;
    for (uint32_t k = 0; k < width / 2; k++) {
        uint64_t a = 1 + 2 * (uint64_t)k, b = 2 + 2 * (uint64_t)k;
        uint64_t* ca = out + (size_t)(2 * k) * n;
        uint64_t* cb = out + (size_t)(2 * k + 1) * n;
        for (size_t i = 0; i < n; i++) {
            ca[i] = a; cb[i] = b;
            uint64_t na = gl::add(a, b), nb = gl::add(b, na);
            a = na; b = nb;
        }
    }
    return AERO_OK;
}

// ---- stage 1 -----------------------------------------------------------------------------------
int32_t aero_interpolate_columns(aero_ctx* ctx, const aero_matrix* trace, aero_matrix** polys) {
    return guard(ctx, [&] {
        REQUIRE(trace && polys, "interpolate_columns: null argument");
        REQUIRE((trace->m.rows & (trace->m.rows - 1)) == 0 && trace->m.rows >= 8, "interpolate_columns: row count must be a power of two >= 8");
        Prover p(ctx->c, ProofOptions::with_96_bit_security());
        aero_matrix* h = new aero_matrix(ctx);
        try { h->m = p.interpolate_columns(trace->m.data.get(), (uint32_t)trace->m.cols, ilog2u(trace->m.rows)); ctx->c->sync(); ctx->c->scratch_reset(); }
        catch (...) { delete h; throw; }
        *polys = h;
    });
}
int32_t aero_evaluate_columns_over(aero_ctx* ctx, const aero_matrix* polys, uint32_t log_blowup, aero_matrix** lde) {
    return guard(ctx, [&] {
        REQUIRE(polys && lde, "evaluate_columns_over: null argument");
        REQUIRE(log_blowup >= 1 && log_blowup <= 7, "evaluate_columns_over: log_blowup must be in [1,7]");
        REQUIRE(ilog2u(polys->m.rows) + (int)log_blowup <= gl::TWO_ADICITY, "evaluate_columns_over: domain exceeds two-adicity");
        Prover p(ctx->c, ProofOptions::with_96_bit_security());
        aero_matrix* h = new aero_matrix(ctx);
        try { h->m = p.evaluate_columns_over(polys->m, (int)log_blowup); ctx->c->sync(); }
        catch (...) { delete h; throw; }
        *lde = h;
    });
}
int32_t aero_poly_eval(aero_ctx* ctx, const aero_matrix* polys, uint64_t z, uint64_t* out) {
    return guard(ctx, [&] {
        REQUIRE(polys && out, "poly_eval: null argument");
        REQUIRE(z < gl::P, "poly_eval: non-canonical field element");
        DevBuf<uint64_t> d(ctx->c, polys->m.cols);
        launch_eval_bitrev<gl::FB>(ctx->c, polys->m.data.get(), polys->m.rows, 0, polys->m.cols, 1, ilog2u(polys->m.rows),
                                   gl::mul(z, gl::inv(gl::GEN)), 0, 1, d.get());
        AERO_HIP(hipMemcpyAsync(out, d.get(), polys->m.cols * 8, hipMemcpyDeviceToHost, ctx->c->stream));
        ctx->c->sync();
        ctx->c->scratch_reset();
    });
}

// ---- hashing / Merkle ----------------------------------------------------------------------------
int32_t aero_hash_rows(aero_ctx* ctx, const uint64_t* rows_row_major, uint32_t width, uint64_t n_rows, uint8_t* digests_out) {
    return guard(ctx, [&] {
        REQUIRE(digests_out && (rows_row_major || n_rows == 0), "hash_rows: null argument");
        REQUIRE(width >= 1, "hash_rows: rows must have at least one element");
        if (n_rows == 0) return;
        // row-major host rows -> column-major device matrix (strided copy), then the coalesced row-hash kernel
        Matrix m(ctx->c, (int)width, n_rows);
        std::vector<uint64_t> cm((size_t)width * n_rows);
        for (uint64_t r = 0; r < n_rows; r++)
            for (uint32_t c = 0; c < width; c++) cm[(size_t)c * n_rows + r] = rows_row_major[r * width + c];
        AERO_HIP(hipMemcpyAsync(m.data.get(), cm.data(), cm.size() * 8, hipMemcpyHostToDevice, ctx->c->stream));
        DevBuf<Digest> d(ctx->c, n_rows);
        ctx->c->hash_rows(m.data.get(), n_rows, (int)width, n_rows, d.get());
        AERO_HIP(hipMemcpyAsync(digests_out, d.get(), n_rows * 32, hipMemcpyDeviceToHost, ctx->c->stream));
        ctx->c->sync();
    });
}
int32_t aero_hash_matrix_rows(aero_ctx* ctx, const aero_matrix* m, uint8_t* digests_out) {
    return guard(ctx, [&] {
        REQUIRE(m, "hash_matrix_rows: null argument");
        DevBuf<Digest> d(ctx->c, m->m.rows);
        ctx->c->hash_rows(m->m.data.get(), m->m.rows, m->m.cols, m->m.rows, d.get());
        if (digests_out) AERO_HIP(hipMemcpyAsync(digests_out, d.get(), m->m.rows * 32, hipMemcpyDeviceToHost, ctx->c->stream));
        ctx->c->sync();
    });
}
int32_t aero_merkle_from_leaves(aero_ctx* ctx, const uint8_t* leaves, uint64_t n, aero_tree** out, uint8_t root_out[32]) {
    return guard(ctx, [&] {
        REQUIRE(leaves && out, "merkle_from_leaves: null argument");
        REQUIRE(n >= 2 && (n & (n - 1)) == 0, "merkle_from_leaves: leaf count must be a power of two >= 2");
        aero_tree* h = new aero_tree(ctx);
        try {
            h->t = MerkleTree(ctx->c, n);
            AERO_HIP(hipMemcpyAsync(h->t.leaves(), leaves, n * 32, hipMemcpyHostToDevice, ctx->c->stream));
            ctx->c->merkle_build(h->t.nodes.get(), n);
            AERO_HIP(hipMemcpyAsync(&h->t.root_host, h->t.nodes.get() + 1, 32, hipMemcpyDeviceToHost, ctx->c->stream));
            ctx->c->sync();
        } catch (...) { delete h; throw; }
        if (root_out) memcpy(root_out, h->t.root_host.w, 32);
        *out = h;
    });
}
int32_t aero_merkle_commit_rows(aero_ctx* ctx, const aero_matrix* m, aero_tree** out, uint8_t root_out[32]) {
    return guard(ctx, [&] {
        REQUIRE(m && out, "merkle_commit_rows: null argument");
        REQUIRE(m->m.rows >= 2 && (m->m.rows & (m->m.rows - 1)) == 0, "merkle_commit_rows: row count must be a power of two >= 2");
        Prover p(ctx->c, ProofOptions::with_96_bit_security());
        aero_tree* h = new aero_tree(ctx);
        try { h->t = p.commit_to_rows(m->m); } catch (...) { delete h; throw; }
        if (root_out) memcpy(root_out, h->t.root_host.w, 32);
        *out = h;
    });
}
int32_t aero_merkle_open_batch(aero_ctx* ctx, const aero_tree* tree, const uint64_t* positions, uint32_t k, uint8_t* out, size_t cap,
                               size_t* out_len) {
    return guard(ctx, [&] {
        REQUIRE(tree && positions && out_len && k >= 1, "merkle_open_batch: null or empty argument");
        std::vector<uint64_t> pos(positions, positions + k);
        Bytes b = open_batch(ctx->c, tree->t, pos);
        *out_len = b.size();
        REQUIRE(out && cap >= b.size(), "merkle_open_batch: output buffer too small");
        memcpy(out, b.data(), b.size());
    });
}
int32_t aero_merkle_nodes(aero_ctx* ctx, const aero_tree* tree, uint8_t* out) {
    return guard(ctx, [&] {
        REQUIRE(tree && out, "merkle_nodes: null argument");
        AERO_HIP(hipMemcpyAsync(out, tree->t.nodes.get(), 2 * tree->t.n * 32, hipMemcpyDeviceToHost, ctx->c->stream));
        ctx->c->sync();
        memset(out, 0, 32);
    });
}
void aero_tree_free(aero_ctx* ctx, aero_tree* tree) {
    (void)ctx;
    delete tree;
}

}  // extern "C"

// ---- constraint evaluation -----------------------------------------------------------------------
// FibAir with its optional auxiliary segment (air == nullptr or aux_width == 0: the plain AIR). `aux_lde` = (A * DEG) x N component
// columns, `rands` = R elements of E; C = the AIR's constraint-evaluation blowup (2 / 4 / 8).
template <class F>
static void eval_constraints_air(Context* c, const Matrix& lde, const Matrix* aux_lde, const aero_fib_air* air, uint32_t log_blowup,
                                 const uint64_t* results, const uint64_t* rands_in, const uint64_t* coeffs, uint32_t frag, uint32_t nfrags,
                                 uint64_t* out_cols, uint64_t* frag_index_out) {
    typedef typename F::T T;
    const uint32_t W = (uint32_t)lde.cols;
    const uint32_t A = air ? air->aux_width : 0, R = A ? air->aux_rands : 0, D = A ? air->aux_degree : 2;
    FibAir shape;
    shape.width = W; shape.aux_width = A; shape.aux_rands = R; shape.aux_degree = D;
    const size_t N = lde.rows, B = (size_t)1 << log_blowup, n = N / B, C = shape.ce_blowup_factor(), ceN = C * n;
    REQUIRE(W >= 2 && !(W & 1), "eval_constraints: FibAir needs an even width");
    REQUIRE(!A || (D >= 2 && D <= 8 && R >= 1 && R <= 255 && A <= 255 - W), "eval_constraints: bad auxiliary segment shape");
    REQUIRE(!A || (aux_lde && rands_in && aux_lde->rows == N && (uint32_t)aux_lde->cols == A * F::DEG), "eval_constraints: auxiliary LDE / random elements missing or of the wrong shape");
    REQUIRE(B >= C && n >= 8, "eval_constraints: blowup smaller than the constraint-evaluation blowup, or trace too short");
    REQUIRE(nfrags >= 1 && (nfrags & (nfrags - 1)) == 0 && ceN / nfrags >= 1 && frag < nfrags, "eval_constraints: bad fragment spec");
    const size_t rows = ceN / nfrags, first = (size_t)frag * rows;
    const size_t nt = shape.num_transition_constraints(), na = shape.num_assertions();
    std::vector<T> ta(nt), tb(nt), ba(na), bb(na), rands(R);
    const uint64_t* p = coeffs;
    auto rd = [&]() { uint64_t c0 = *p++; uint64_t c1 = F::DEG > 1 ? *p++ : 0; REQUIRE(c0 < gl::P && c1 < gl::P, "eval_constraints: non-canonical coefficient"); return F::make(c0, c1); };
    for (size_t i = 0; i < nt; i++) { ta[i] = rd(); tb[i] = rd(); }
    for (size_t i = 0; i < na; i++) { ba[i] = rd(); bb[i] = rd(); }
    p = rands_in;
    for (uint32_t i = 0; i < R; i++) rands[i] = rd();
    std::vector<uint64_t> res(results, results + W / 2);
    auto up = [&](const void* src, size_t bytes) { void* d = c->scratch_alloc(bytes + 8); AERO_HIP(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, c->stream)); return d; };
    const int log_ce = ilog2u(ceN);
    NttTables* tce = c->ntt_tables(log_ce);
    FibConsArgs<F> a{};
    a.lde = lde.data.get(); a.N = N; a.W = W; a.C = (uint32_t)C; a.blowup = (uint32_t)B; a.ce_step = (uint32_t)(B / C); a.xmask = (uint32_t)C - 1;
    a.first = first; a.count = rows;
    a.ta = (const T*)up(ta.data(), nt * sizeof(T)); a.tb = (const T*)up(tb.data(), nt * sizeof(T));
    a.ba = (const T*)up(ba.data(), na * sizeof(T)); a.bb = (const T*)up(bb.data(), na * sizeof(T));
    a.results = (const uint64_t*)up(res.data(), res.size() * 8);
    a.tw_lo = tce->lo_fwd; a.tw_hi = tce->hi_fwd; a.twi_lo = tce->lo_inv; a.twi_hi = tce->hi_inv; a.tw_h = tce->h;
    a.offset = gl::GEN; a.gen_inv = gl::inv(gl::GEN); a.k7 = gl::pow(gl::GEN, ceN);
    std::vector<uint64_t> xn(C), zn(C), xnp(C);
    uint64_t g7n = gl::pow(gl::GEN, n), wC = gl::root_of_unity(ilog2u(C));
    for (size_t k = 0; k < C; k++) {
        const uint64_t xnk = gl::mul(g7n, gl::pow(wC, k));
        xnp[k] = gl::pow(xnk, C + 1 - D);                  // aux degree adjustment x^((C + 1 - D) n + (D - 2)): its x^n part
        xn[k] = gl::inv(xnk); zn[k] = gl::inv(gl::sub(xnk, 1));
    }
    a.xn_inv = (const uint64_t*)up(xn.data(), C * 8); a.zn_inv = (const uint64_t*)up(zn.data(), C * 8); a.xn = (const uint64_t*)up(xnp.data(), C * 8);
    a.aux = A ? aux_lde->data.get() : nullptr; a.A = A; a.R = R; a.D = D;
    a.rands = A ? (const T*)up(rands.data(), R * sizeof(T)) : nullptr;
    a.w_last = gl::pow(gl::root_of_unity(ilog2u(n)), n - 1);
    DevBuf<uint64_t> d_out(c, 3 * F::DEG * rows);
    a.out_cols = d_out.get();
    launch_fib_constraints<F>(c, a, 0);
    AERO_HIP(hipMemcpyAsync(out_cols, d_out.get(), 3 * F::DEG * rows * 8, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    c->scratch_reset();
    if (frag_index_out) *frag_index_out = first;
}
// auxiliary columns of the stand-in AIR from the main trace and the drawn elements: (A * DEG) x n component columns
template <class F> static void aux_columns_abi(aero_ctx* ctx, const Matrix& trace, const aero_fib_air* air, const uint64_t* rands_in, aero_matrix** out) {
    typedef typename F::T T;
    Context* c = ctx->c;
    const uint32_t W = (uint32_t)trace.cols, A = air->aux_width, R = air->aux_rands, D = air->aux_degree;
    REQUIRE(A >= 1 && A <= 255 - W && R >= 1 && R <= 255 && D >= 2 && D <= 8, "aux_columns: bad auxiliary segment shape");
    std::vector<T> rands(R);
    const uint64_t* p = rands_in;
    for (uint32_t i = 0; i < R; i++) {
        const uint64_t c0 = *p++, c1 = F::DEG > 1 ? *p++ : 0;
        REQUIRE(c0 < gl::P && c1 < gl::P, "aux_columns: non-canonical random element");
        rands[i] = F::make(c0, c1);
    }
    T* d_r = (T*)c->scratch_alloc(R * sizeof(T) + 8);
    AERO_HIP(hipMemcpyAsync(d_r, rands.data(), R * sizeof(T), hipMemcpyHostToDevice, c->stream));
    std::unique_ptr<aero_matrix> m(new aero_matrix(ctx));
    m->m = Matrix(c, (int)(A * F::DEG), trace.rows);
    launch_aux_columns<F>(c, trace.data.get(), trace.rows, W, A, R, D, d_r, m->m.data.get());
    c->sync();
    c->scratch_reset();
    *out = m.release();
}
extern "C" {

int32_t aero_eval_constraints_fib(aero_ctx* ctx, const aero_matrix* trace_lde, uint32_t log_blowup, const uint64_t* results,
                                  const uint64_t* coeffs, uint8_t field_extension, uint32_t fragment_offset, uint32_t num_fragments,
                                  uint64_t* out_cols, uint64_t* frag_index_out) {
    return aero_eval_constraints_air(ctx, trace_lde, nullptr, nullptr, log_blowup, results, nullptr, coeffs, field_extension, fragment_offset,
                                     num_fragments, out_cols, frag_index_out);
}
int32_t aero_eval_constraints_air(aero_ctx* ctx, const aero_matrix* trace_lde, const aero_matrix* aux_lde, const aero_fib_air* air,
                                  uint32_t log_blowup, const uint64_t* results, const uint64_t* rands, const uint64_t* coeffs,
                                  uint8_t field_extension, uint32_t fragment_offset, uint32_t num_fragments, uint64_t* out_cols,
                                  uint64_t* frag_index_out) {
    return guard(ctx, [&] {
        REQUIRE(trace_lde && results && coeffs && out_cols, "eval_constraints: null argument");
        REQUIRE(log_blowup >= 1 && log_blowup <= 7, "eval_constraints: log_blowup must be in [1,7]");
        const Matrix* aux = aux_lde ? &aux_lde->m : nullptr;
        if (field_extension == EXT_NONE) eval_constraints_air<gl::FB>(ctx->c, trace_lde->m, aux, air, log_blowup, results, rands, coeffs, fragment_offset, num_fragments, out_cols, frag_index_out);
        else if (field_extension == EXT_QUADRATIC) eval_constraints_air<gl::FQ>(ctx->c, trace_lde->m, aux, air, log_blowup, results, rands, coeffs, fragment_offset, num_fragments, out_cols, frag_index_out);
        else fail("eval_constraints: field extension must be 1 (None) or 2 (Quadratic)", ST_UNSUPPORTED);
    });
}
int32_t aero_aux_columns_fib(aero_ctx* ctx, const aero_matrix* trace, const aero_fib_air* air, const uint64_t* rands, uint8_t field_extension,
                             aero_matrix** aux_out) {
    return guard(ctx, [&] {
        REQUIRE(trace && air && rands && aux_out, "aux_columns: null argument");
        if (field_extension == EXT_NONE) aux_columns_abi<gl::FB>(ctx, trace->m, air, rands, aux_out);
        else if (field_extension == EXT_QUADRATIC) aux_columns_abi<gl::FQ>(ctx, trace->m, air, rands, aux_out);
        else fail("aux_columns: field extension must be 1 (None) or 2 (Quadratic)", ST_UNSUPPORTED);
    });
}

}  // extern "C"

// ---- composition polynomial / DEEP / FRI layers ------------------------------------------------------------------
template <class F> static void composition_poly_fib(aero_ctx* ctx, const uint64_t* numer_cols, uint32_t log_n, aero_matrix** out, size_t C = FibAir::plain_ce_blowup_factor()) {
    Context* c = ctx->c;
    const size_t n = (size_t)1 << log_n, ceN = C * n;
    const int log_ce = ilog2u(ceN);
    for (size_t i = 0; i < 3 * F::DEG * ceN; i++) REQUIRE(numer_cols[i] < gl::P, "composition_poly_fib: non-canonical element");
    DevBuf<uint64_t> d_cols(c, 3 * F::DEG * ceN);
    AERO_HIP(hipMemcpyAsync(d_cols.get(), numer_cols, 3 * F::DEG * ceN * 8, hipMemcpyHostToDevice, c->stream));
    auto m = new aero_matrix(ctx);
    std::unique_ptr<aero_matrix> guard_m(m);
    m->m = Matrix(c, (int)(C * F::DEG), n);     // [DEG][ce_n] = columns ordered [component][c]
    NttTables* tce = c->ntt_tables(log_ce);
    FibDivideArgs<F> a{};
    a.cols = d_cols.get(); a.ce_n = ceN; a.C = (uint32_t)C;
    a.tw_lo = tce->lo_fwd; a.tw_hi = tce->hi_fwd; a.tw_h = tce->h;
    a.offset = gl::GEN; a.w_last = gl::pow(gl::root_of_unity((int)log_n), n - 1);
    std::vector<uint64_t> zn(C);
    const uint64_t g7n = gl::pow(gl::GEN, n), wC = gl::root_of_unity(ilog2u(C));
    for (size_t k = 0; k < C; k++) zn[k] = gl::inv(gl::sub(gl::mul(g7n, gl::pow(wC, k)), 1));
    uint64_t* d_zn = (uint64_t*)c->scratch_alloc(C * 8 + 8);
    AERO_HIP(hipMemcpyAsync(d_zn, zn.data(), C * 8, hipMemcpyHostToDevice, c->stream));
    a.zn_inv = d_zn;
    for (int d = 0; d < F::DEG; d++) a.out_h[d] = m->m.data.get() + (size_t)d * ceN;
    launch_fib_divide<F>(c, a);
    ProofOptions po = ProofOptions::with_96_bit_security();
    Prover p(c, po);
    p.composition_from_evaluations(m->m.data.get(), F::DEG, log_ce, ilog2u(C), gl::GEN);
    c->sync();
    c->scratch_reset();
    *out = guard_m.release();
}
template <class F> static typename F::T rd_elem(const uint64_t*& p, const char* what) {
    uint64_t c0 = *p++, c1 = F::DEG > 1 ? *p++ : 0;
    if (c0 >= gl::P || c1 >= gl::P) fail(std::string(what) + ": non-canonical element");
    return F::make(c0, c1);
}
template <class F>
static void deep_compose_abi(aero_ctx* ctx, const Matrix& tlde, const Matrix& clde, uint32_t log_blowup, const uint64_t* z, const uint64_t* ood_frame,
                             const uint64_t* ood_evals, const uint64_t* coeffs, aero_matrix** out) {
    Context* c = ctx->c;
    const size_t N = tlde.rows, W = tlde.cols;
    REQUIRE(clde.rows == N && clde.cols % F::DEG == 0 && clde.cols > 0, "deep_compose: composition LDE shape does not match");
    const size_t C = clde.cols / F::DEG;
    REQUIRE((N >> log_blowup) >= 8 && ((N >> log_blowup) << log_blowup) == N, "deep_compose: LDE rows are not trace_length << log_blowup");
    const int log_n = ilog2u(N >> log_blowup);
    DeepInputs<F> in;
    in.z = rd_elem<F>(z, "deep_compose");
    in.ood_cur.resize(W); in.ood_next.resize(W); in.ood_h.resize(C);
    in.da.resize(W); in.db.resize(W); in.dg.resize(W); in.dc.resize(C);
    for (size_t i = 0; i < W; i++) in.ood_cur[i] = rd_elem<F>(ood_frame, "deep_compose");
    for (size_t i = 0; i < W; i++) in.ood_next[i] = rd_elem<F>(ood_frame, "deep_compose");
    for (size_t i = 0; i < C; i++) in.ood_h[i] = rd_elem<F>(ood_evals, "deep_compose");
    for (size_t i = 0; i < W; i++) { in.da[i] = rd_elem<F>(coeffs, "deep_compose"); in.db[i] = rd_elem<F>(coeffs, "deep_compose"); in.dg[i] = rd_elem<F>(coeffs, "deep_compose"); }
    for (size_t i = 0; i < C; i++) in.dc[i] = rd_elem<F>(coeffs, "deep_compose");
    in.lambda = rd_elem<F>(coeffs, "deep_compose"); in.mu = rd_elem<F>(coeffs, "deep_compose");
    ProofOptions po = ProofOptions::with_96_bit_security();
    Prover p(c, po);
    auto m = new aero_matrix(ctx);
    std::unique_ptr<aero_matrix> guard_m(m);
    m->m.data = p.deep_compose<F>(tlde.data.get(), clde.data.get(), nullptr, (uint32_t)W, 0, (uint32_t)C, log_n, (int)log_blowup, gl::GEN, in);
    m->m.rows = N; m->m.cols = F::DEG;
    c->sync();
    c->scratch_reset();
    *out = guard_m.release();
}
extern "C" {
int32_t aero_composition_poly_fib(aero_ctx* ctx, const uint64_t* numer_cols, uint32_t log_n, uint8_t field_extension, aero_matrix** comp_polys) {
    return guard(ctx, [&] {
        REQUIRE(numer_cols && comp_polys, "composition_poly_fib: null argument");
        REQUIRE(log_n >= 3 && log_n <= 28, "composition_poly_fib: log_n out of range");
        if (field_extension == EXT_NONE) composition_poly_fib<gl::FB>(ctx, numer_cols, log_n, comp_polys);
        else if (field_extension == EXT_QUADRATIC) composition_poly_fib<gl::FQ>(ctx, numer_cols, log_n, comp_polys);
        else fail("composition_poly_fib: field extension must be 1 (None) or 2 (Quadratic)", ST_UNSUPPORTED);
    });
}
int32_t aero_composition_poly_air(aero_ctx* ctx, const uint64_t* numer_cols, uint32_t log_n, uint32_t num_columns, uint8_t field_extension,
                                  aero_matrix** comp_polys) {
    return guard(ctx, [&] {
        REQUIRE(numer_cols && comp_polys, "composition_poly_air: null argument");
        REQUIRE(log_n >= 3 && log_n <= 28, "composition_poly_air: log_n out of range");
        REQUIRE(num_columns == 2 || num_columns == 4 || num_columns == 8, "composition_poly_air: 2, 4 or 8 composition columns");
        if (field_extension == EXT_NONE) composition_poly_fib<gl::FB>(ctx, numer_cols, log_n, comp_polys, num_columns);
        else if (field_extension == EXT_QUADRATIC) composition_poly_fib<gl::FQ>(ctx, numer_cols, log_n, comp_polys, num_columns);
        else fail("composition_poly_air: field extension must be 1 (None) or 2 (Quadratic)", ST_UNSUPPORTED);
    });
}
int32_t aero_deep_compose(aero_ctx* ctx, const aero_matrix* trace_lde, const aero_matrix* comp_lde, uint32_t log_blowup, uint8_t field_extension,
                          const uint64_t* z, const uint64_t* ood_frame, const uint64_t* ood_evals, const uint64_t* coeffs, aero_matrix** deep_evals) {
    return guard(ctx, [&] {
        REQUIRE(trace_lde && comp_lde && z && ood_frame && ood_evals && coeffs && deep_evals, "deep_compose: null argument");
        REQUIRE(log_blowup >= 1 && log_blowup <= 7, "deep_compose: log_blowup must be in [1,7]");
        if (field_extension == EXT_NONE) deep_compose_abi<gl::FB>(ctx, trace_lde->m, comp_lde->m, log_blowup, z, ood_frame, ood_evals, coeffs, deep_evals);
        else if (field_extension == EXT_QUADRATIC) deep_compose_abi<gl::FQ>(ctx, trace_lde->m, comp_lde->m, log_blowup, z, ood_frame, ood_evals, coeffs, deep_evals);
        else fail("deep_compose: field extension must be 1 (None) or 2 (Quadratic)", ST_UNSUPPORTED);
    });
}
int32_t aero_fri_build_layers(aero_ctx* ctx, const aero_matrix* evals, const aero_proof_options* o, const uint8_t seed_in[32], uint8_t* roots_out,
                              size_t roots_cap, uint32_t* num_roots, uint8_t seed_out[32], aero_fri** out) {
    return guard(ctx, [&] {
        REQUIRE(evals && o && seed_in && roots_out && num_roots && seed_out && out, "fri_build_layers: null argument");
        ProofOptions po{o->num_queries, o->blowup_factor, o->grinding_factor, o->hash_fn, o->field_extension, o->fri_folding_factor, o->fri_log_max_remainder};
        Prover p(ctx->c, po);
        const int deg = po.field_extension == EXT_QUADRATIC ? 2 : 1;
        const Matrix& m = evals->m;
        REQUIRE(m.cols == deg, "fri_build_layers: evaluations must have one column per extension component");
        REQUIRE(m.rows >= 2 && (m.rows & (m.rows - 1)) == 0, "fri_build_layers: domain size must be a power of two");
        const int layers = num_fri_layers(m.rows, po.fri_folding_factor, 1ull << po.fri_log_max_remainder);
        REQUIRE(roots_cap >= (size_t)(layers + 1) * 32, "fri_build_layers: roots buffer too small");
        HostCoin coin;
        memcpy(coin.seed.w, seed_in, 32);
        DevBuf<uint64_t> copy(ctx->c, (size_t)deg * m.rows);     // the layers own their evaluations
        AERO_HIP(hipMemcpyAsync(copy.get(), m.data.get(), (size_t)deg * m.rows * 8, hipMemcpyDeviceToDevice, ctx->c->stream));
        auto f = new aero_fri(ctx);
        std::unique_ptr<aero_fri> guard_f(f);
        f->opt = po;
        Bytes roots;
        if (deg == 1) f->fl = p.fri_build_layers<gl::FB>(std::move(copy), m.rows, coin, &roots);
        else f->fl = p.fri_build_layers<gl::FQ>(std::move(copy), m.rows, coin, &roots);
        memcpy(roots_out, roots.data(), roots.size());
        *num_roots = (uint32_t)(roots.size() / 32);
        memcpy(seed_out, coin.seed.w, 32);
        ctx->c->scratch_reset();
        *out = guard_f.release();
    });
}
int32_t aero_fri_open(aero_ctx* ctx, const aero_fri* fri, const uint64_t* positions, uint32_t k, uint8_t** out, size_t* out_len) {
    return guard(ctx, [&] {
        REQUIRE(fri && positions && out && out_len && k > 0, "fri_open: null argument");
        std::vector<uint64_t> pos(positions, positions + k);
        for (uint64_t q : pos) REQUIRE(q < fri->fl.lde_size, "fri_open: position out of range");
        Prover p(ctx->c, fri->opt);
        Bytes b = fri->fl.deg == 1 ? p.fri_open<gl::FB>(fri->fl, pos) : p.fri_open<gl::FQ>(fri->fl, pos);
        uint8_t* buf = (uint8_t*)malloc(b.size());
        if (!buf) throw std::bad_alloc();
        memcpy(buf, b.data(), b.size());
        *out = buf; *out_len = b.size();
    });
}
void aero_fri_free(aero_ctx* ctx, aero_fri* fri) {
    if (!fri) return;
    if (ctx && ctx->c) (void)hipSetDevice(ctx->c->device);
    delete fri;
}
}  // extern "C"

extern "C" {

// ---- FRI / grinding ---------------------------------------------------------------------------------
int32_t aero_fri_fold(aero_ctx* ctx, const uint64_t* values, uint64_t dom, uint32_t fold, uint64_t alpha, uint64_t* out) {
    return guard(ctx, [&] {
        REQUIRE(values && out, "fri_fold: null argument");
        REQUIRE(fold == 2 || fold == 4 || fold == 8 || fold == 16, "fri_fold: folding factor must be 2, 4, 8 or 16");
        REQUIRE(dom >= fold && (dom & (dom - 1)) == 0 && ilog2u(dom) <= gl::TWO_ADICITY, "fri_fold: domain must be a power of two >= fold");
        REQUIRE(alpha < gl::P, "fri_fold: non-canonical alpha");
        Context* c = ctx->c;
        const size_t rows = dom / fold;
        DevBuf<uint64_t> d_in(c, dom), d_out(c, rows);
        AERO_HIP(hipMemcpyAsync(d_in.get(), values, dom * 8, hipMemcpyHostToDevice, c->stream));
        NttTables* td = c->ntt_tables(ilog2u(dom));
        FoldArgs<gl::FB> a{};
        a.in[0] = a.in[1] = d_in.get(); a.out[0] = a.out[1] = d_out.get();
        a.rows = rows; a.fold = (int)fold; a.alpha = alpha;
        a.twi_lo = td->lo_inv; a.twi_hi = td->hi_inv; a.tw_h = td->h;
        a.gen_inv = gl::inv(gl::GEN); a.fold_inv = gl::inv(fold);
        uint64_t wFi = gl::inv(gl::root_of_unity(ilog2u(fold)));
        for (uint32_t m = 0; m < fold; m++) a.dft[m] = gl::pow(wFi, m);
        launch_fri_fold<gl::FB>(c, a);
        AERO_HIP(hipMemcpyAsync(out, d_out.get(), rows * 8, hipMemcpyDeviceToHost, c->stream));
        c->sync();
    });
}
int32_t aero_grind(aero_ctx* ctx, const uint8_t seed[32], uint32_t bits, uint64_t* nonce_out) {
    return guard(ctx, [&] {
        REQUIRE(seed && nonce_out, "grind: null argument");
        REQUIRE(bits <= 32, "grind: more than 32 bits of grinding is not supported");
        Digest s;
        memcpy(s.w, seed, 32);
        *nonce_out = run_grind(ctx->c, s, bits);
    });
}

}  // extern "C"

// ---- whole proof ----------------------------------------------------------------------------------------
static void do_prove(aero_ctx* ctx, const uint64_t* trace_dev, uint32_t width, int log_n, const aero_proof_options* o, uint8_t** proof,
                     size_t* proof_len, uint64_t* pub_out, const aero_comm* comm = nullptr, uint32_t aux_width = 0, uint32_t aux_rands = 0,
                     uint32_t aux_degree = 2, const uint64_t* trace_host = nullptr, unsigned int* verdict = nullptr) {
    REQUIRE(o && proof && proof_len, "prove: null argument");
    *proof = nullptr; *proof_len = 0;       // whatever goes wrong below, the caller is not left with a stale pointer
    ProofOptions po{o->num_queries, o->blowup_factor, o->grinding_factor, o->hash_fn, o->field_extension, o->fri_folding_factor, o->fri_log_max_remainder};
    Prover p(ctx->c, po);
    p.set_aux_segment(aux_width, aux_rands, aux_degree);
    if (trace_host) {
        if (ctx->c->landed.dev && !(comm && comm->world > 1)) {
            if (ctx->c->landed.bytes != ((size_t)width << log_n) * 8) fail("prove: the landed trace does not have this call's shape", ST_INTERNAL);
            trace_dev = ctx->c->landed.dev; p.set_landed_trace(trace_host, ctx->c->landed.ready, verdict); }
        else p.set_host_trace(trace_host, verdict);
    }
    if (comm) p.set_comm(shard_comm_of(comm, "prove_fib_sharded"));
    p.collect_stage_times = ctx->stage_timing;
    if (ctx->concurrent_peers) p.h2d_pipeline = false;      // column groups on a second stream: no gain under other proofs (profiles/r5_h2d.md)
    std::vector<uint64_t> pub;
    Bytes b = p.prove(trace_dev, width, log_n, &pub);
    ctx->last_ms = p.last_stage_ms;
    // prove-then-verify (main.rs:47, proving_worker.rs:196-203); a trace that failed the canonical-form check is the caller's error, reported by it
    if (self_verify_wanted(ctx, comm) && !(verdict && *verdict != 0)) {
        const aero_fib_air a{aux_width, aux_width ? aux_rands : 0, aux_width ? aux_degree : 2};
        run_self_verify(ctx, b, pub, &a, nullptr, (uint32_t)log_n, *o);
    }
    uint8_t* buf = (uint8_t*)malloc(b.size());
    if (!buf) throw std::bad_alloc();
    memcpy(buf, b.data(), b.size());
    *proof = buf; *proof_len = b.size();
    if (pub_out) memcpy(pub_out, pub.data(), pub.size() * 8);
}
extern "C" {

int32_t aero_prove_fib(aero_ctx* ctx, const aero_matrix* trace, const aero_proof_options* options, uint8_t** proof, size_t* proof_len,
                       uint64_t* pub_out) {
    return guard(ctx, [&] {
        REQUIRE(trace, "prove_fib: null trace");
        REQUIRE((trace->m.rows & (trace->m.rows - 1)) == 0, "prove_fib: trace length must be a power of two");
        do_prove(ctx, trace->m.data.get(), (uint32_t)trace->m.cols, ilog2u(trace->m.rows), options, proof, proof_len, pub_out);
    });
}
int32_t aero_prove_fib_sharded(aero_ctx* ctx, const aero_comm* comm, const aero_matrix* trace, const aero_proof_options* options,
                               uint8_t** proof, size_t* proof_len, uint64_t* pub_out) {
    return guard(ctx, [&] {
        REQUIRE(comm, "prove_fib_sharded: null comm");
        REQUIRE(trace, "prove_fib_sharded: null trace");
        REQUIRE((trace->m.rows & (trace->m.rows - 1)) == 0, "prove_fib_sharded: trace length must be a power of two");
        do_prove(ctx, trace->m.data.get(), (uint32_t)trace->m.cols, ilog2u(trace->m.rows), options, proof, proof_len, pub_out, comm);
    });
}
int32_t aero_commit_trace_sharded(aero_ctx* ctx, const aero_comm* comm, const aero_matrix* trace, const aero_proof_options* o, uint8_t root_out[32],
                                  uint8_t* subtree_roots_out) {
    return guard(ctx, [&] {
        REQUIRE(trace && o && root_out, "commit_trace_sharded: null argument");
        REQUIRE((trace->m.rows & (trace->m.rows - 1)) == 0, "commit_trace_sharded: trace length must be a power of two");
        ProofOptions po{o->num_queries, o->blowup_factor, o->grinding_factor, o->hash_fn, o->field_extension, o->fri_folding_factor, o->fri_log_max_remainder};
        Prover p(ctx->c, po);
        if (comm) p.set_comm(shard_comm_of(comm, "commit_trace_sharded"));
        p.trace_commit_only = true;
        const Bytes b = p.prove(trace->m.data.get(), (uint32_t)trace->m.cols, ilog2u(trace->m.rows), nullptr);
        const size_t G = comm ? (size_t)comm->world : 1;
        REQUIRE(b.size() == 32 * (1 + G), "commit_trace_sharded: unexpected result size");
        memcpy(root_out, b.data(), 32);
        if (subtree_roots_out) memcpy(subtree_roots_out, b.data() + 32, 32 * G);
    });
}
int32_t aero_prove_fib_aux(aero_ctx* ctx, const aero_comm* comm, const aero_matrix* trace, uint32_t aux_width, uint32_t aux_rands,
                           const aero_proof_options* options, uint8_t** proof, size_t* proof_len, uint64_t* pub_out) {
    return guard(ctx, [&] {
        REQUIRE(trace, "prove_fib_aux: null trace");
        REQUIRE((trace->m.rows & (trace->m.rows - 1)) == 0, "prove_fib_aux: trace length must be a power of two");
        do_prove(ctx, trace->m.data.get(), (uint32_t)trace->m.cols, ilog2u(trace->m.rows), options, proof, proof_len, pub_out, comm, aux_width, aux_rands);
    });
}
int32_t aero_prove_fib_air(aero_ctx* ctx, const aero_comm* comm, const aero_matrix* trace, const aero_fib_air* air, const aero_proof_options* options,
                           uint8_t** proof, size_t* proof_len, uint64_t* pub_out) {
    return guard(ctx, [&] {
        REQUIRE(trace && air, "prove_fib_air: null argument");
        REQUIRE((trace->m.rows & (trace->m.rows - 1)) == 0, "prove_fib_air: trace length must be a power of two");
        do_prove(ctx, trace->m.data.get(), (uint32_t)trace->m.cols, ilog2u(trace->m.rows), options, proof, proof_len, pub_out, comm, air->aux_width,
                 air->aux_rands, air->aux_width ? air->aux_degree : 2);
    });
}
}  // extern "C"
// Trace in host memory -> proof: the host-to-device copy is enqueued on the context's stream, the canonical-form check runs
// behind it WITHOUT a stream synchronisation of its own (its verdict is read after the proof, which is discarded when the
// trace held an element >= p), then the proof. Pinned host memory (aero_host_register) makes the copy a true async DMA.
static void prove_from_host(aero_ctx* ctx, const uint64_t* trace_col_major, uint32_t width, uint32_t log_n, const aero_fib_air* air,
                            const aero_proof_options* options, uint8_t** proof, size_t* proof_len, uint64_t* pub_out, const aero_comm* comm = nullptr) {
    REQUIRE(trace_col_major, "prove_fib_host: null trace");
    REQUIRE(width >= 2 && width <= 254 && log_n >= 3 && log_n <= 29, "prove_fib_host: bad shape");
    Context* c = ctx->c;
    const size_t n = (size_t)1 << log_n;
    unsigned int* verdict = c->pinned_word();
    *verdict = 0;
    // no device copy of the trace outlives the proof: the columns go straight into the interpolation buffer (Prover::set_host_trace;
    // with an auxiliary segment a copy is kept until its columns are built)
    const uint32_t A = air ? air->aux_width : 0;
    (void)n; (void)c;
    do_prove(ctx, nullptr, width, (int)log_n, options, proof, proof_len, pub_out, comm, A, A ? air->aux_rands : 0, A ? air->aux_degree : 2, trace_col_major, verdict);
    if (*verdict != 0) {   // the proof's own synchronisations have long passed the check
        free(*proof);
        *proof = nullptr; *proof_len = 0;
        fail("prove_fib_host: trace holds a non-canonical field element (>= p)");
    }
}
extern "C" {
int32_t aero_prove_fib_host(aero_ctx* ctx, const uint64_t* trace_col_major, uint32_t width, uint32_t log_n, const aero_proof_options* options,
                            uint8_t** proof, size_t* proof_len, uint64_t* pub_out) {
    return guard(ctx, [&] { prove_from_host(ctx, trace_col_major, width, log_n, nullptr, options, proof, proof_len, pub_out); });
}
int32_t aero_prove_fib_air_host(aero_ctx* ctx, const uint64_t* trace_col_major, uint32_t width, uint32_t log_n, const aero_fib_air* air,
                                const aero_proof_options* options, uint8_t** proof, size_t* proof_len, uint64_t* pub_out) {
    return guard(ctx, [&] { prove_from_host(ctx, trace_col_major, width, log_n, air, options, proof, proof_len, pub_out); });
}
int32_t aero_prove_fib_sharded_host(aero_ctx* ctx, const aero_comm* comm, const uint64_t* trace_col_major, uint32_t width, uint32_t log_n,
                                    const aero_fib_air* air, const aero_proof_options* options, uint8_t** proof, size_t* proof_len, uint64_t* pub_out) {
    return guard(ctx, [&] { prove_from_host(ctx, trace_col_major, width, log_n, air, options, proof, proof_len, pub_out, comm); });
}
int32_t aero_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return AERO_E_BAD_ARG;
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); g_create_err = std::string("hipHostRegister: ") + hipGetErrorString(e); return e == hipErrorOutOfMemory ? AERO_E_OOM : AERO_E_HIP; }
    return AERO_OK;
}
int32_t aero_host_unregister(void* p) {
    if (!p) return AERO_E_BAD_ARG;
    hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { (void)hipGetLastError(); g_create_err = std::string("hipHostUnregister: ") + hipGetErrorString(e); return AERO_E_HIP; }
    return AERO_OK;
}
// Pinned host memory owned by the library's runtime (hipHostMalloc): DMA-able from the start, nothing of the caller's address space is
// registered after the fact (no user-pointer mapping that has to follow the kernel's page migrations).
int32_t aero_host_alloc(size_t bytes, void** out) {
    if (!out || !bytes) return AERO_E_BAD_ARG;
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; g_create_err = std::string("hipHostMalloc: ") + hipGetErrorString(e); return e == hipErrorOutOfMemory ? AERO_E_OOM : AERO_E_HIP; }
    return AERO_OK;
}
int32_t aero_host_free(void* p) {
    if (!p) return AERO_OK;
    hipError_t e = hipHostFree(p);
    if (e != hipSuccess) { (void)hipGetLastError(); g_create_err = std::string("hipHostFree: ") + hipGetErrorString(e); return AERO_E_HIP; }
    return AERO_OK;
}
int32_t aero_proof_container(const uint8_t* inputs, size_t inputs_len, const uint8_t* proof, size_t proof_len, uint8_t** out, size_t* out_len) {
    if (!out || !out_len || (!inputs && inputs_len) || (!proof && proof_len)) return AERO_E_BAD_ARG;
    size_t total = 16 + inputs_len + proof_len;
    uint8_t* b = (uint8_t*)malloc(total);
    if (!b) return AERO_E_OOM;
    uint64_t l = inputs_len;
    memcpy(b, &l, 8);
    if (inputs_len) memcpy(b + 8, inputs, inputs_len);
    l = proof_len;
    memcpy(b + 8 + inputs_len, &l, 8);
    if (proof_len) memcpy(b + 16 + inputs_len, proof, proof_len);
    *out = b; *out_len = total;
    return AERO_OK;
}

// ---- trace files ---------------------------------------------------------------------------------------------
// "AEROTRC" + format version byte, u32 width, u32 log_n, u32 air_id, u32 aux_width, u32 aux_rands, u32 aux_degree (all
// little-endian), then width * 2^log_n u64 little-endian in column-major order (include/aero_stark.h).
namespace {
const char TRACE_MAGIC[8] = {'A', 'E', 'R', 'O', 'T', 'R', 'C', 1};
struct TraceHeader { uint32_t width, log_n, air_id, aux_width, aux_rands, aux_degree; };
struct FileCloser { FILE* f; ~FileCloser() { if (f) fclose(f); } };
void read_header(FILE* f, const char* path, TraceHeader* h) {
    char magic[8];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, TRACE_MAGIC, 8) != 0) fail(std::string("trace file: ") + path + " is not an AEROTRC version-1 file");
    uint32_t w[6];
    if (fread(w, 4, 6, f) != 6) fail("trace file: truncated header");
    *h = TraceHeader{w[0], w[1], w[2], w[3], w[4], w[5]};
    if (h->width < 1 || h->width > 255 || h->log_n < 3 || h->log_n > 29) fail("trace file: width must be in [1,255] and log_n in [3,29]");
}
}  // namespace

int32_t aero_trace_file_write(const char* path, const uint64_t* col_major, uint32_t width, uint32_t log_n, uint32_t air_id, const aero_fib_air* air) {
    if (!path || !col_major || width < 1 || width > 255 || log_n < 3 || log_n > 29) return AERO_E_BAD_ARG;
    FileCloser fc{fopen(path, "wb")};
    if (!fc.f) { g_create_err = std::string("trace file: cannot create ") + path; return AERO_E_BAD_ARG; }
    const uint32_t w[6] = {width, log_n, air_id, air ? air->aux_width : 0, air ? air->aux_rands : 0, air ? air->aux_degree : 2};
    const size_t count = (size_t)width << log_n;
    if (fwrite(TRACE_MAGIC, 1, 8, fc.f) != 8 || fwrite(w, 4, 6, fc.f) != 6 || fwrite(col_major, 8, count, fc.f) != count) {
        g_create_err = std::string("trace file: short write to ") + path;
        return AERO_E_INTERNAL;
    }
    return AERO_OK;
}
int32_t aero_trace_file_info(const char* path, uint32_t* width, uint32_t* log_n, uint32_t* air_id, aero_fib_air* air) {
    if (!path) return AERO_E_BAD_ARG;
    try {
        FileCloser fc{fopen(path, "rb")};
        if (!fc.f) fail(std::string("trace file: cannot open ") + path);
        TraceHeader h;
        read_header(fc.f, path, &h);
        if (width) *width = h.width;
        if (log_n) *log_n = h.log_n;
        if (air_id) *air_id = h.air_id;
        if (air) *air = aero_fib_air{h.aux_width, h.aux_rands, h.aux_degree};
        return AERO_OK;
    } catch (const Error& e) { g_create_err = e.what(); return e.code; }
}
// The file is streamed through two pinned buffers: while chunk i travels to the device, chunk i + 1 is read from disk.
int32_t aero_trace_file_load(aero_ctx* ctx, const char* path, aero_matrix** trace_out, uint32_t* air_id, aero_fib_air* air) {
    return guard(ctx, [&] {
        REQUIRE(path && trace_out, "trace_file_load: null argument");
        *trace_out = nullptr;
        Context* c = ctx->c;
        FileCloser fc{fopen(path, "rb")};
        if (!fc.f) fail(std::string("trace file: cannot open ") + path);
        TraceHeader h;
        read_header(fc.f, path, &h);
        // 0 = the built-in FibAir; 2 = the constraint set travels as an AEROAIR program next to the file (include/aero_air.h:
        // aero_prove_air takes the matrix); 1 = Miden's ProcessorAir by name, whose constraint set this library does not contain
        if (h.air_id != AERO_AIR_FIB && h.air_id != 2u /* AERO_AIR_PROGRAM */)
            fail("trace file: AIR id " + std::to_string(h.air_id) + " is not built into this library (0 = FibAir, 2 = a constraint program handed to "
                 "aero_prove_air; Miden's ProcessorAir, id 1, is not in the reference mount: export it as a program and write the file with id 2)", ST_UNSUPPORTED);
        const size_t count = (size_t)h.width << h.log_n;
        std::unique_ptr<aero_matrix> m(new aero_matrix(ctx));
        m->m = Matrix(c, (int)h.width, (size_t)1 << h.log_n);
        const size_t chunk = (size_t)4 << 20;   // elements per chunk (32 MiB)
        uint64_t* pin[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};
        auto cleanup = [&] { for (int i = 0; i < 2; i++) { if (pin[i]) (void)hipHostFree(pin[i]); if (done[i]) (void)hipEventDestroy(done[i]); } };
        try {
            for (int i = 0; i < 2; i++) { AERO_HIP(hipHostMalloc((void**)&pin[i], std::min(chunk, count) * 8, hipHostMallocDefault)); AERO_HIP(hipEventCreate(&done[i])); }
            size_t off = 0;
            for (int k = 0; off < count; k ^= 1) {
                const size_t n = std::min(chunk, count - off);
                AERO_HIP(hipEventSynchronize(done[k]));                   // the previous copy out of this buffer has finished
                if (fread(pin[k], 8, n, fc.f) != n) fail("trace file: fewer elements than width * 2^log_n");
                AERO_HIP(hipMemcpyAsync(m->m.data.get() + off, pin[k], n * 8, hipMemcpyHostToDevice, c->stream));
                AERO_HIP(hipEventRecord(done[k], c->stream));
                off += n;
            }
            if (fgetc(fc.f) != EOF) fail("trace file: trailing bytes after the trace");
            if (!all_canonical(c, m->m.data.get(), count)) fail("trace file: the trace holds a non-canonical field element (>= p)");
        } catch (...) { (void)hipStreamSynchronize(c->stream); cleanup(); throw; }
        cleanup();
        if (air_id) *air_id = h.air_id;
        if (air) *air = aero_fib_air{h.aux_width, h.aux_rands, h.aux_degree};
        *trace_out = m.release();
    });
}

// ---- instrumentation ------------------------------------------------------------------------------------
int32_t aero_set_stage_timing(aero_ctx* ctx, int32_t enable) {
    if (!ctx) return AERO_E_BAD_ARG;
    ctx->stage_timing = enable != 0;
    return AERO_OK;
}
int32_t aero_last_stage_ms(const aero_ctx* ctx, double out[12]) {
    if (!ctx || !out) return AERO_E_BAD_ARG;
    const StageMs& m = ctx->last_ms;
    double v[12] = {m.interpolate, m.lde, m.trace_commit, m.constraints, m.composition, m.comp_commit, m.ood, m.deep, m.fri, m.grind, m.queries, m.total};
    memcpy(out, v, sizeof v);
    return AERO_OK;
}
int32_t aero_set_kernel_timing(aero_ctx* ctx, int32_t enable, const char* only_kernel) {
    return guard(ctx, [&] {
        if (!enable && ctx->c->kernel_timing) ctx->c->kt_report();
        ctx->c->kernel_timing = enable != 0;
        ctx->c->kt_filter = (enable && only_kernel) ? only_kernel : "";
    });
}
int32_t aero_kernel_timing_report(aero_ctx* ctx, char* buf, size_t cap) {
    return guard(ctx, [&] {
        REQUIRE(buf && cap > 0, "kernel_timing_report: null buffer");
        std::string s = ctx->c->kt_report();
        REQUIRE(s.size() + 1 <= cap, "kernel_timing_report: buffer too small");
        memcpy(buf, s.c_str(), s.size() + 1);
    });
}
int32_t aero_memory_stats(const aero_ctx* ctx, uint64_t* in_use, uint64_t* peak) {
    if (!ctx || !ctx->c) return AERO_E_BAD_ARG;
    if (in_use) *in_use = ctx->c->bytes_in_use;
    if (peak) *peak = ctx->c->bytes_peak;
    return AERO_OK;
}

}  // extern "C"

// ---- pool: several proofs in flight on one device, one worker thread per context ---------------------------------------
struct aero_pool {
    struct Slot {
        aero_ctx* ctx = nullptr;
        std::thread th;
        // job (written by the caller under `mu`, read by the worker)
        const aero_matrix* trace = nullptr;
        const uint64_t* host_trace = nullptr;   // host-trace batch (aero_pool_prove_fib_host): copied in before every proof
        // queue mode (aero_pool_prove_*_queue): this slot's share of a queue of DIFFERENT host traces, in order; result k of the slot goes to
        // entry queue_out[k] of the caller's arrays (the reference deals its batches the same way: batch_idx % concurrency, pool.rs:105-124)
        std::vector<const uint64_t*> queue;
        std::vector<uint32_t> queue_out;
        uint8_t* proof = nullptr;
        size_t proof_len = 0;
        std::vector<uint64_t> pub;
        int32_t status = 0;
        bool has_job = false;
    };
    std::vector<std::unique_ptr<Slot>> slots;
    std::shared_ptr<Context::CopyGate> gate = std::make_shared<Context::CopyGate>();   // orders the slots' host-to-device copies (aero_internal.hpp)
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    uint64_t generation = 0;       // bumped per batch
    int pending = 0;
    bool stop = false;
    // batch parameters
    aero_fib_air air{0, 0, 2};
    const aero_air* program = nullptr;     // batch of a program AIR (aero_pool_prove_air*): proven through aero_prove_air[_host]
    std::vector<uint64_t> program_pub;
    aero_proof_options opt{};
    uint32_t rounds = 1;
    uint32_t host_width = 0, host_log_n = 0;
    // queue mode: the caller's result arrays (one entry per queued trace) and, for programs, one statement per trace
    uint8_t** q_proofs = nullptr;
    size_t* q_lens = nullptr;
    uint64_t* q_pubs = nullptr;                 // built-in AIR: width / 2 results per trace
    const uint64_t* q_program_pubs = nullptr;   // programs: n_pub elements per trace (null: program_pub for all)

    int numa_node = -1;                     // of the pool's device (numa.hip); -1 = unknown
    std::atomic<uint32_t> pinned{0};        // worker threads bound to that node's CPUs

    void worker(Slot* s) {
        // this thread issues every launch and every copy of its slot: it runs on the GPU's own socket where the box tells which that is
        if (numa_node >= 0 && bind_thread_to_node(numa_node) > 0) pinned++;
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return stop || (generation != seen && s->has_job); });
                if (stop) return;
                seen = generation;
            }
            int32_t rc = AERO_OK;
            uint8_t* out = nullptr;
            size_t len = 0;
            std::vector<uint64_t> pub;
            // an exception that leaves a thread body ends the whole host process (std::terminate): whatever is thrown here
            // (the entry points below catch their own) becomes the slot's status
            try {
            const bool queued = !s->queue.empty();
            const uint32_t rounds = queued ? (uint32_t)s->queue.size() : this->rounds;        // shadows the batch parameter
            auto trace_of = [&](uint32_t r) { return queued ? s->queue[r] : s->host_trace; };
            pub.resize((size_t)((s->host_trace || queued) ? host_width : (uint32_t)s->trace->m.cols) / 2);
            // Prefetch (round 5): a slot whose stream sits behind a 10 ms copy is a slot that does not compute - eight slots of 2^20 x 72
            // proofs behaved like four or five (3.5 against 4.2 G cells/s resident with the link at half its rate, profiles/r5_h2d.md).
            // With several rounds ahead the copy of trace r + 1 runs on the copy stream, into the other of two landing buffers, WHILE
            // trace r is proven; every trace still crosses the link inside the call. Narrow traces (< 32 MiB) keep the in-stream copy:
            // their copy is short and hides behind the other proofs as it is. AERO_POOL_PREFETCH_MIN_MB overrides the size, 0 = never.
            static const double prefetch_mb = getenv("AERO_POOL_PREFETCH_MIN_MB") ? atof(getenv("AERO_POOL_PREFETCH_MIN_MB")) : 32.0;
            uint32_t trace_w = host_width;
            if (program) { uint32_t info[16] = {0}; if (aero_air_info(program, info) == AERO_OK) trace_w = info[0]; }      // main_width
            const size_t trace_bytes = (s->host_trace || queued) ? ((size_t)trace_w << host_log_n) * 8 : 0;
            const bool prefetch = (s->host_trace || queued) && rounds > 1 && prefetch_mb > 0 && (double)trace_bytes >= prefetch_mb * 1048576.0;
            Context* const c = s->ctx->c;
            DevBuf<uint64_t> land[2];
            hipStream_t cs = nullptr;
            // Declared behind the landing buffers, so it runs BEFORE they return to the context's allocator on every way out of this block
            // (an exception included): nothing of this job is left on the copy stream, the context offers no landed trace any more, and the
            // pool's gate does not keep pointing at an event of this context.
            struct PrefetchEnd {
                Context* c; hipStream_t* cs; Context::CopyGate* gate; bool on;
                ~PrefetchEnd() {
                    if (!on) return;
                    c->landed = Context::LandedTrace{};
                    if (*cs) (void)hipStreamSynchronize(*cs);
                    std::lock_guard<std::mutex> lk(gate->mu);      // a fired event is a no-op to wait on, but it is this context's: leave none behind
                    if (gate->last == c->sync_event(32) || gate->last == c->sync_event(33)) gate->last = nullptr;
                }
            } prefetch_end{c, &cs, gate.get(), prefetch};
            auto send = [&](uint32_t r) {       // the copy of round r's trace, behind whatever the copy stream still carries
                // one copy at a time per pool, in the order the slots ask (Context::CopyGate): the first fills then start the slots one copy
                // apart and they stay out of phase - proofs that run in lockstep queue their latency-bound stages behind each other
                std::lock_guard<std::mutex> lk(gate->mu);
                if (gate->last) AERO_HIP(hipStreamWaitEvent(cs, gate->last, 0));
                AERO_HIP(hipMemcpyAsync(land[r & 1].get(), trace_of(r), trace_bytes, hipMemcpyHostToDevice, cs));
                AERO_HIP(hipEventRecord(c->sync_event(32 + (r & 1)), cs));
                gate->last = c->sync_event(32 + (r & 1));
            };
            if (prefetch) {
                rc = guard(s->ctx, [&] {
                    land[0] = DevBuf<uint64_t>(c, trace_bytes / 8); land[1] = DevBuf<uint64_t>(c, trace_bytes / 8);
                    cs = c->get_copy_stream();
                    send(0);
                });
            }
            for (uint32_t r = 0; r < rounds && rc == AERO_OK; r++) {
                if (out) { free(out); out = nullptr; }
                if (prefetch) {
                    // round r - 1 has returned (the calls below are synchronous): its landing buffer is free for round r + 1
                    if (r + 1 < rounds) rc = guard(s->ctx, [&] { send(r + 1); });
                    if (rc != AERO_OK) break;
                    c->landed.dev = land[r & 1].get(); c->landed.ready = c->sync_event(32 + (r & 1)); c->landed.bytes = trace_bytes;
                }
                const uint64_t* ht = trace_of(r);
                if (program) {
                    const uint32_t np = (uint32_t)program_pub.size();
                    const uint64_t* pb = (queued && q_program_pubs) ? q_program_pubs + (size_t)s->queue_out[r] * np : program_pub.data();
                    if (ht) rc = aero_prove_air_host(s->ctx, program, ht, host_log_n, pb, np, &opt, &out, &len);
                    else rc = aero_prove_air(s->ctx, nullptr, program, s->trace, pb, np, &opt, &out, &len);
                }
                else if (ht) rc = aero_prove_fib_air_host(s->ctx, ht, host_width, host_log_n, &air, &opt, &out, &len, pub.data());
                else rc = aero_prove_fib_air(s->ctx, nullptr, s->trace, &air, &opt, &out, &len, pub.data());
                c->landed = Context::LandedTrace{};
                if (queued && rc == AERO_OK) {        // every proof of a queue goes back to the caller, in the caller's order
                    const uint32_t o = s->queue_out[r];
                    q_proofs[o] = out; q_lens[o] = len; out = nullptr; len = 0;
                    if (q_pubs && !program) memcpy(q_pubs + (size_t)o * (host_width / 2), pub.data(), (size_t)(host_width / 2) * 8);
                }
            }
            } catch (const std::bad_alloc&) { rc = AERO_E_OOM; s->ctx->err = "pool worker: host allocation failed"; }
            catch (...) { rc = AERO_E_INTERNAL; s->ctx->err = "pool worker: unexpected exception"; }
            {
                std::lock_guard<std::mutex> lk(mu);
                s->status = rc; s->proof = out; s->proof_len = len; s->pub = std::move(pub); s->has_job = false;
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
};

extern "C" {
int32_t aero_pool_create(int32_t device_id, uint32_t slots, aero_pool** out) {
    if (!out || slots == 0 || slots > 64) return AERO_E_BAD_ARG;
    *out = nullptr;
    std::unique_ptr<aero_pool> p(new (std::nothrow) aero_pool());
    if (!p) return AERO_E_OOM;
    for (uint32_t i = 0; i < slots; i++) {
        aero_ctx* c = nullptr;
        int32_t rc = aero_ctx_create(device_id, &c);
        if (rc != AERO_OK) {
            for (auto& s : p->slots) aero_ctx_destroy(s->ctx);
            return rc;
        }
        p->slots.emplace_back(new aero_pool::Slot());
        p->slots.back()->ctx = c;
        c->c->copy_gate = p->gate;
    }
    p->numa_node = numa_node_of_device(device_id);
    for (auto& s : p->slots) s->th = std::thread(&aero_pool::worker, p.get(), s.get());
    *out = p.release();
    return AERO_OK;
}
void aero_pool_destroy(aero_pool* pool) {
    if (!pool) return;
    {
        std::lock_guard<std::mutex> lk(pool->mu);
        pool->stop = true;
    }
    pool->cv_job.notify_all();
    for (auto& s : pool->slots) if (s->th.joinable()) s->th.join();
    for (auto& s : pool->slots) aero_ctx_destroy(s->ctx);
    delete pool;
}
int32_t aero_pool_set_self_verify(aero_pool* pool, int32_t mode) {
    if (!pool || mode < AERO_SELF_VERIFY_AUTO || mode > AERO_SELF_VERIFY_ON) return AERO_E_BAD_ARG;
    std::lock_guard<std::mutex> lk(pool->mu);
    if (pool->pending) return AERO_E_BAD_ARG;      // a batch is running
    for (auto& s : pool->slots) s->ctx->self_verify = mode;
    return AERO_OK;
}
int32_t aero_pool_placement(const aero_pool* pool, int32_t* node_out, uint32_t* pinned_out) {
    if (!pool) return AERO_E_BAD_ARG;
    if (node_out) *node_out = pool->numa_node;
    if (pinned_out) *pinned_out = pool->pinned.load();
    return AERO_OK;
}
uint32_t aero_pool_slots(const aero_pool* pool) { return pool ? (uint32_t)pool->slots.size() : 0; }
aero_ctx* aero_pool_ctx(aero_pool* pool, uint32_t slot) { return (pool && slot < pool->slots.size()) ? pool->slots[slot]->ctx : nullptr; }
static int32_t pool_run(aero_pool* pool, const aero_matrix* const* traces, const uint64_t* const* host_traces, uint32_t width, uint32_t log_n,
                        uint32_t count, const aero_fib_air* air, const aero_proof_options* options, uint32_t rounds, uint8_t** proofs,
                        size_t* proof_lens, uint64_t* pubs, const aero_air* program = nullptr, const uint64_t* program_pub = nullptr, uint32_t n_pub = 0) {
    // a program's evaluation kernel is built HERE, once, before the workers start: no slot meets a compilation inside its proof
    // (a refusal by hiprtc is not an error: the proofs run interpreted)
    if (program) (void)aero_air_prepare(program, traces ? (uint32_t)ilog2u(traces[0]->m.rows) : log_n, options, 1);
    {
        std::lock_guard<std::mutex> lk(pool->mu);
        pool->program = program;
        pool->program_pub.assign(program_pub, program_pub + (program ? n_pub : 0));
        pool->air = air ? *air : aero_fib_air{0, 0, 2};
        pool->opt = *options;
        pool->rounds = rounds;
        pool->host_width = width; pool->host_log_n = log_n;
        for (uint32_t i = 0; i < count; i++) {
            pool->slots[i]->trace = traces ? traces[i] : nullptr;
            pool->slots[i]->host_trace = host_traces ? host_traces[i] : nullptr;
            pool->slots[i]->ctx->concurrent_peers = count > 1;
            pool->slots[i]->has_job = true;
        }
        pool->pending = (int)count;
        pool->generation++;
    }
    pool->cv_job.notify_all();
    {
        std::unique_lock<std::mutex> lk(pool->mu);
        pool->cv_done.wait(lk, [&] { return pool->pending == 0; });
        // the batch is over: a proof proven on one of these contexts on its own (single-proof latency) pipelines its copy again
        for (uint32_t i = 0; i < count; i++) pool->slots[i]->ctx->concurrent_peers = false;
    }
    int32_t first = AERO_OK;
    size_t poff = 0;
    for (uint32_t i = 0; i < count; i++) {
        aero_pool::Slot& s = *pool->slots[i];
        proofs[i] = s.proof; proof_lens[i] = s.proof_len;
        if (pubs && s.status == AERO_OK) memcpy(pubs + poff, s.pub.data(), s.pub.size() * 8);
        poff += (size_t)(traces ? (uint32_t)traces[i]->m.cols : width) / 2;
        if (first == AERO_OK && s.status != AERO_OK) first = s.status;
        s.proof = nullptr;
    }
    return first;
}
int32_t aero_pool_prove_fib(aero_pool* pool, const aero_matrix* const* traces, uint32_t count, const aero_fib_air* air,
                            const aero_proof_options* options, uint32_t rounds, uint8_t** proofs, size_t* proof_lens, uint64_t* pubs) {
    if (!pool || !traces || !options || !proofs || !proof_lens || count == 0 || count > pool->slots.size() || rounds == 0) return AERO_E_BAD_ARG;
    for (uint32_t i = 0; i < count; i++)
        if (!traces[i] || traces[i]->keep.get() != pool->slots[i]->ctx->c) return AERO_E_BAD_ARG;   // must live on slot i's context
    return pool_run(pool, traces, nullptr, 0, 0, count, air, options, rounds, proofs, proof_lens, pubs);
}
int32_t aero_pool_prove_fib_host(aero_pool* pool, const uint64_t* const* host_traces, uint32_t width, uint32_t log_n, uint32_t count,
                                 const aero_fib_air* air, const aero_proof_options* options, uint32_t rounds, uint8_t** proofs,
                                 size_t* proof_lens, uint64_t* pubs) {
    if (!pool || !host_traces || !options || !proofs || !proof_lens || count == 0 || count > pool->slots.size() || rounds == 0) return AERO_E_BAD_ARG;
    if (width < 2 || width > 254 || log_n < 3 || log_n > 29) return AERO_E_BAD_ARG;
    for (uint32_t i = 0; i < count; i++) if (!host_traces[i]) return AERO_E_BAD_ARG;
    return pool_run(pool, nullptr, host_traces, width, log_n, count, air, options, rounds, proofs, proof_lens, pubs);
}
// the same for an AIR given as a constraint program (include/aero_air.h): one program and one statement for the whole batch
int32_t aero_pool_prove_air(aero_pool* pool, const aero_air* air, const aero_matrix* const* traces, uint32_t count, const uint64_t* pub, uint32_t n_pub,
                            const aero_proof_options* options, uint32_t rounds, uint8_t** proofs, size_t* proof_lens) {
    if (!pool || !air || !traces || !options || !proofs || !proof_lens || (n_pub && !pub) || count == 0 || count > pool->slots.size() || rounds == 0) return AERO_E_BAD_ARG;
    for (uint32_t i = 0; i < count; i++)
        if (!traces[i] || traces[i]->keep.get() != pool->slots[i]->ctx->c) return AERO_E_BAD_ARG;
    return pool_run(pool, traces, nullptr, 0, 0, count, nullptr, options, rounds, proofs, proof_lens, nullptr, air, pub, n_pub);
}
int32_t aero_pool_prove_air_host(aero_pool* pool, const aero_air* air, const uint64_t* const* host_traces, uint32_t log_n, uint32_t count,
                                 const uint64_t* pub, uint32_t n_pub, const aero_proof_options* options, uint32_t rounds, uint8_t** proofs,
                                 size_t* proof_lens) {
    if (!pool || !air || !host_traces || !options || !proofs || !proof_lens || (n_pub && !pub) || count == 0 || count > pool->slots.size() || rounds == 0) return AERO_E_BAD_ARG;
    if (log_n < 3 || log_n > 29) return AERO_E_BAD_ARG;
    for (uint32_t i = 0; i < count; i++) if (!host_traces[i]) return AERO_E_BAD_ARG;
    return pool_run(pool, nullptr, host_traces, 2, log_n, count, nullptr, options, rounds, proofs, proof_lens, nullptr, air, pub, n_pub);
}
// A queue of DIFFERENT host traces of one shape through the pool: trace t goes to slot t mod slots (the reference's pool deals its batches the
// same way, pool.rs:105-124), every slot proves its share in order - copying its next trace while it proves the current one when they are large
// enough to be worth it - and EVERY proof comes back: proofs[t] (malloc'd, aero_free), proof_lens[t], pubs + t * width / 2.
static int32_t pool_run_queue(aero_pool* pool, const uint64_t* const* host_traces, uint32_t n_traces, uint32_t width, uint32_t log_n, const aero_fib_air* air,
                              const aero_proof_options* options, uint8_t** proofs, size_t* proof_lens, uint64_t* pubs, const aero_air* program,
                              const uint64_t* program_pubs, uint32_t n_pub) {
    const uint32_t S = (uint32_t)pool->slots.size(), used = n_traces < S ? n_traces : S;
    for (uint32_t t = 0; t < n_traces; t++) { proofs[t] = nullptr; proof_lens[t] = 0; }
    if (program) (void)aero_air_prepare(program, log_n, options, 1);
    {
        std::lock_guard<std::mutex> lk(pool->mu);
        pool->program = program;
        pool->program_pub.assign(n_pub, 0);
        if (program && program_pubs) pool->program_pub.assign(program_pubs, program_pubs + n_pub);
        pool->q_program_pubs = program ? program_pubs : nullptr;
        pool->air = air ? *air : aero_fib_air{0, 0, 2};
        pool->opt = *options;
        pool->rounds = 1;
        pool->host_width = width; pool->host_log_n = log_n;
        pool->q_proofs = proofs; pool->q_lens = proof_lens; pool->q_pubs = pubs;
        for (uint32_t i = 0; i < used; i++) {
            aero_pool::Slot& s = *pool->slots[i];
            s.trace = nullptr; s.host_trace = nullptr;
            s.queue.clear(); s.queue_out.clear();
            for (uint32_t t = i; t < n_traces; t += S) { s.queue.push_back(host_traces[t]); s.queue_out.push_back(t); }
            s.ctx->concurrent_peers = used > 1;
            s.has_job = true;
        }
        pool->pending = (int)used;
        pool->generation++;
    }
    pool->cv_job.notify_all();
    {
        std::unique_lock<std::mutex> lk(pool->mu);
        pool->cv_done.wait(lk, [&] { return pool->pending == 0; });
        for (uint32_t i = 0; i < used; i++) { pool->slots[i]->ctx->concurrent_peers = false; pool->slots[i]->queue.clear(); pool->slots[i]->queue_out.clear(); }
        pool->q_proofs = nullptr; pool->q_lens = nullptr; pool->q_pubs = nullptr; pool->q_program_pubs = nullptr;
    }
    int32_t first = AERO_OK;
    for (uint32_t i = 0; i < used; i++) {
        aero_pool::Slot& s = *pool->slots[i];
        if (first == AERO_OK && s.status != AERO_OK) first = s.status;
        if (s.proof) { free(s.proof); s.proof = nullptr; }      // a slot that failed mid-queue may hold a last buffer
    }
    if (first != AERO_OK)        // all or nothing: the caller gets no partial result set to sort out
        for (uint32_t t = 0; t < n_traces; t++) if (proofs[t]) { free(proofs[t]); proofs[t] = nullptr; proof_lens[t] = 0; }
    return first;
}
int32_t aero_pool_prove_fib_queue(aero_pool* pool, const uint64_t* const* host_traces, uint32_t n_traces, uint32_t width, uint32_t log_n,
                                  const aero_fib_air* air, const aero_proof_options* options, uint8_t** proofs, size_t* proof_lens, uint64_t* pubs) {
    if (!pool || !host_traces || !options || !proofs || !proof_lens || n_traces == 0 || n_traces > (1u << 24)) return AERO_E_BAD_ARG;
    if (width < 2 || width > 254 || log_n < 3 || log_n > 29) return AERO_E_BAD_ARG;
    for (uint32_t t = 0; t < n_traces; t++) if (!host_traces[t]) return AERO_E_BAD_ARG;
    return pool_run_queue(pool, host_traces, n_traces, width, log_n, air, options, proofs, proof_lens, pubs, nullptr, nullptr, 0);
}
int32_t aero_pool_prove_air_queue(aero_pool* pool, const aero_air* air, const uint64_t* const* host_traces, uint32_t n_traces, uint32_t log_n,
                                  const uint64_t* pubs_per_trace, uint32_t n_pub, const aero_proof_options* options, uint8_t** proofs, size_t* proof_lens) {
    if (!pool || !air || !host_traces || !options || !proofs || !proof_lens || (n_pub && !pubs_per_trace) || n_traces == 0 || n_traces > (1u << 24)) return AERO_E_BAD_ARG;
    if (log_n < 3 || log_n > 29) return AERO_E_BAD_ARG;
    for (uint32_t t = 0; t < n_traces; t++) if (!host_traces[t]) return AERO_E_BAD_ARG;
    return pool_run_queue(pool, host_traces, n_traces, 2, log_n, nullptr, options, proofs, proof_lens, nullptr, air, pubs_per_trace, n_pub);
}
// ---- the reference's worker seam at the message level (worker_messages.hpp) ------------------------------------------------------
// hashing_worker.rs:12-26: every row of the work item -> Blake2s_256::hash_elements, answered in row order with the batch index.
int32_t aero_worker_hash_rows(aero_ctx* ctx, const uint8_t* work_item, size_t work_item_len, uint8_t** result, size_t* result_len) {
    return guard(ctx, [&] {
        REQUIRE(work_item && result && result_len, "worker_hash_rows: null argument");
        *result = nullptr; *result_len = 0;
        // the rows are hashed where they lie in the message: one scan of the row headers on the host, the bytes go to the device
        // as they are (every field of this message is a u64), one lane per row
        std::vector<uint64_t> offs;
        const uint64_t batch_idx = wm::scan_hashing_work_item(work_item, work_item_len, offs);
        const size_t k = offs.size();
        std::vector<uint8_t> digests(32 * k);
        if (k) {
            Context* c = ctx->c;
            const size_t words = work_item_len / 8;
            DevBuf<uint64_t> d_msg(c, words), d_offs(c, k);
            DevBuf<Digest> out(c, k);
            AERO_HIP(hipMemcpyAsync(d_msg.get(), work_item, work_item_len, hipMemcpyHostToDevice, c->stream));
            AERO_HIP(hipMemcpyAsync(d_offs.get(), offs.data(), k * 8, hipMemcpyHostToDevice, c->stream));
            c->hash_message_rows(MsgSrc{d_msg.get(), d_offs.get()}, k, out.get());
            AERO_HIP(hipMemcpyAsync(digests.data(), out.get(), k * sizeof(Digest), hipMemcpyDeviceToHost, c->stream));
            c->sync();
        }
        const std::vector<uint8_t> msg = wm::emit_hashing_result(batch_idx, digests.data(), k);
        uint8_t* buf = (uint8_t*)malloc(msg.size() ? msg.size() : 1);
        if (!buf) throw std::bad_alloc();
        memcpy(buf, msg.data(), msg.size());
        *result = buf; *result_len = msg.size();
    });
}
}  // extern "C"
