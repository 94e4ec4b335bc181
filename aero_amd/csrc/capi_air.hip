// extern "C" boundary for program AIRs (declarations + reference citations: include/aero_air.h).
#include "../../include/aero_air.h"
#include "air_host.hpp"
#include "capi_internal.hpp"
#include "proof_format.hpp"
#include "stark_kernels.hpp"
#include "worker_messages.hpp"

using namespace aero;

static void put_err(char* err, size_t cap, const std::string& s) {
    if (err && cap) { const size_t k = std::min(cap - 1, s.size()); memcpy(err, s.data(), k); err[k] = 0; }
}
static ProofOptions to_options(const aero_proof_options* o) {
    return ProofOptions{o->num_queries, o->blowup_factor, o->grinding_factor, o->hash_fn, o->field_extension, o->fri_folding_factor, o->fri_log_max_remainder};
}
static uint8_t* to_malloc(const std::vector<uint8_t>& b, size_t* len) {
    uint8_t* buf = (uint8_t*)malloc(b.size() ? b.size() : 1);
    if (!buf) throw std::bad_alloc();
    if (!b.empty()) memcpy(buf, b.data(), b.size());
    *len = b.size();
    return buf;
}
template <class F> static std::vector<typename F::T> read_elems(const uint64_t* p, size_t count, const char* what) {
    std::vector<typename F::T> v(count);
    for (size_t i = 0; i < count; i++) {
        const uint64_t c0 = *p++, c1 = F::DEG > 1 ? *p++ : 0;
        if (c0 >= gl::P || c1 >= gl::P) fail(std::string(what) + ": non-canonical element");
        v[i] = F::make(c0, c1);
    }
    return v;
}
static std::vector<uint64_t> read_pub(const air::Program& p, const uint64_t* pub, uint32_t n_pub, const char* what) {
    if (n_pub != p.num_pub || (!pub && n_pub)) fail(std::string(what) + ": the program takes " + std::to_string(p.num_pub) + " public inputs");
    std::vector<uint64_t> v(pub, pub + n_pub);
    for (uint64_t x : v) if (x >= gl::P) fail(std::string(what) + ": non-canonical public input");
    return v;
}

extern "C" {

int32_t aero_air_load(const uint8_t* program, size_t len, aero_air** out, char* err, size_t err_cap) {
    if (!out) return AERO_E_BAD_ARG;
    *out = nullptr;
    try {
        if (!program) fail("air program: null pointer");
        std::unique_ptr<aero_air> a(new aero_air());
        a->prog = air::load(program, len);
        a->bytes.assign(program, program + len);
        *out = a.release();
        put_err(err, err_cap, "");
        return AERO_OK;
    } catch (const Error& e) { put_err(err, err_cap, e.what()); return e.code; }
    catch (const std::bad_alloc&) { put_err(err, err_cap, "host allocation failed"); return AERO_E_OOM; }
    catch (const std::exception& e) { put_err(err, err_cap, e.what()); return AERO_E_INTERNAL; }
}
void aero_air_free(aero_air* air) { delete air; }

int32_t aero_air_fib_program(uint32_t width, const aero_fib_air* desc, uint8_t** program, size_t* len) {
    if (!program || !len) return AERO_E_BAD_ARG;
    try {
        const std::vector<uint8_t> b = air::fib_program(width, desc ? desc->aux_width : 0, desc ? desc->aux_rands : 0, desc && desc->aux_width ? desc->aux_degree : 2);
        *program = to_malloc(b, len);
        return AERO_OK;
    } catch (const Error& e) { g_create_err = e.what(); return e.code; }
    catch (const std::bad_alloc&) { return AERO_E_OOM; }
}
int32_t aero_air_synth_vm_program(uint32_t log_n, uint32_t pairs, uint32_t aux, uint32_t rands, uint8_t** program, size_t* len) {
    if (!program || !len) return AERO_E_BAD_ARG;
    try {
        const std::vector<uint8_t> b = air::synth_vm_program(log_n, pairs, aux, rands);
        *program = to_malloc(b, len);
        return AERO_OK;
    } catch (const Error& e) { g_create_err = e.what(); return e.code; }
    catch (const std::bad_alloc&) { return AERO_E_OOM; }
}
int32_t aero_air_synth_vm_trace(uint32_t log_n, uint32_t pairs, uint64_t* trace_out, uint64_t* pub_out) {
    if (!trace_out || log_n < 4 || log_n > 29 || pairs < 1 || 20 + 2 * pairs > 254) return AERO_E_BAD_ARG;
    air::synth_vm_trace(log_n, pairs, trace_out, pub_out);
    return AERO_OK;
}
int32_t aero_air_info(const aero_air* air, uint32_t out[16]) {
    if (!air || !out) return AERO_E_BAD_ARG;
    const air::Program& p = air->prog;
    const uint32_t v[16] = {p.W, p.A, p.R, p.num_pub, p.exemptions, p.n_main_trans, (uint32_t)p.trans.size() - p.n_main_trans,
                            (uint32_t)p.masserts.size(), (uint32_t)p.aasserts.size(), p.ce_blowup, (uint32_t)p.periodic.size(), (uint32_t)p.nodes.size(),
                            (uint32_t)p.cons_code.size(), p.cons_slotsB, p.cons_slotsE, p.has_builders() ? 1u : 0u};
    memcpy(out, v, sizeof v);
    return AERO_OK;
}
int32_t aero_air_num_divisors(const aero_air* air, uint32_t log_n, uint32_t* out) {
    if (!air || !out) return AERO_E_BAD_ARG;
    try { *out = (uint32_t)air::instantiate(air->prog, (int)log_n).num_columns(); return AERO_OK; }
    catch (const Error& e) { g_create_err = e.what(); return e.code; }
}

int32_t aero_air_jit_compile(const aero_air* air, uint32_t log_n, uint32_t field_extension, int32_t fused) {
    if (!air || (field_extension != 1 && field_extension != 2)) return AERO_E_BAD_ARG;
    try {
        const air::Instance in = air::instantiate(air->prog, (int)log_n);
        std::string err;
        if (air_jit_compile_only(air->prog, in, (int)field_extension, fused ? 1 : 0, &err)) return AERO_OK;
        g_create_err = err;
        return AERO_E_UNSUPPORTED;
    } catch (const Error& e) { g_create_err = e.what(); return e.code; }
    catch (const std::bad_alloc&) { return AERO_E_OOM; }
}
int32_t aero_air_prepare(const aero_air* air, uint32_t log_n, const aero_proof_options* options, uint32_t world) {
    if (!air || !options || world == 0 || (world & (world - 1)) || log_n < 3 || log_n > 29) return AERO_E_BAD_ARG;
    if (options->field_extension != 1 && options->field_extension != 2) return AERO_E_BAD_ARG;
    if (const char* e = getenv("AERO_AIR_JIT")) if (e[0] == '0') return AERO_OK;      // interpreter: nothing to build
    try {
        // rows of one evaluation launch: the constraint domain (ce_blowup * n), or a rank's coset of the LDE when that is smaller
        const size_t n = (size_t)1 << log_n, ce = (size_t)air->prog.ce_blowup * n, shard = (size_t)options->blowup_factor * n / world;
        std::string err;
        if (air_jit_prepare(air->prog, (int)log_n, (int)options->field_extension, world > 1 && shard < ce ? shard : ce, &err)) return AERO_OK;
        g_create_err = err;
        return AERO_E_UNSUPPORTED;
    } catch (const Error& e) { g_create_err = e.what(); return e.code; }
    catch (const std::bad_alloc&) { return AERO_E_OOM; }
    catch (const std::exception& e) { g_create_err = e.what(); return AERO_E_INTERNAL; }
}
int32_t aero_air_jit_source(const aero_air* air, uint32_t log_n, uint32_t field_extension, int32_t fused, uint8_t** source, size_t* len) {
    if (!air || !source || !len || (field_extension != 1 && field_extension != 2)) return AERO_E_BAD_ARG;
    try {
        const std::string s = air_jit_source(air->prog, air::instantiate(air->prog, (int)log_n), (int)field_extension, fused ? 1 : 0);
        *source = to_malloc(std::vector<uint8_t>(s.begin(), s.end()), len);
        return AERO_OK;
    } catch (const Error& e) { g_create_err = e.what(); return e.code; }
    catch (const std::bad_alloc&) { return AERO_E_OOM; }
}

}  // extern "C"

// ---- whole proof -----------------------------------------------------------------------------------------------------------
static void prove_program(aero_ctx* ctx, const aero_comm* comm, const aero_air* air, const uint64_t* trace_dev, const uint64_t* trace_host,
                          unsigned int* verdict, uint32_t width, int log_n, const uint64_t* pub, uint32_t n_pub, const aero_proof_options* o,
                          uint8_t** proof, size_t* proof_len) {
    REQUIRE(air && o && proof && proof_len, "prove_air: null argument");
    *proof = nullptr; *proof_len = 0;
    const air::Program& p = air->prog;
    const std::vector<uint64_t> pubv = read_pub(p, pub, n_pub, "prove_air");
    REQUIRE(width == p.W, "prove_air: the trace does not have the program's main width");
    if (p.A && !p.has_builders()) fail("prove_air: the program does not say how its auxiliary columns are built (no aux builders)", ST_UNSUPPORTED);
    Prover pr(ctx->c, to_options(o));
    pr.set_program(&p, pubv);
    if (trace_host) {
        if (ctx->c->landed.dev && !(comm && comm->world > 1)) {
            if (ctx->c->landed.bytes != ((size_t)width << log_n) * 8) fail("prove_air: the landed trace does not have this call's shape", ST_INTERNAL);
            trace_dev = ctx->c->landed.dev; pr.set_landed_trace(trace_host, ctx->c->landed.ready, verdict); }
        else pr.set_host_trace(trace_host, verdict);
    }
    if (comm) pr.set_comm(shard_comm_of(comm, "prove_air"));
    pr.collect_stage_times = ctx->stage_timing;
    if (ctx->concurrent_peers) pr.h2d_pipeline = false;      // column groups on a second stream: no gain under other proofs (profiles/r5_h2d.md)
    const Bytes b = pr.prove(trace_dev, width, log_n, nullptr);
    ctx->last_ms = pr.last_stage_ms;
    if (self_verify_wanted(ctx, comm) && !(verdict && *verdict != 0)) run_self_verify(ctx, b, pubv, nullptr, &p, (uint32_t)log_n, *o);
    *proof = to_malloc(b, proof_len);
}

extern "C" {

int32_t aero_prove_air(aero_ctx* ctx, const aero_comm* comm, const aero_air* air, const aero_matrix* trace, const uint64_t* pub, uint32_t n_pub,
                       const aero_proof_options* options, uint8_t** proof, size_t* proof_len) {
    return guard(ctx, [&] {
        REQUIRE(trace, "prove_air: null trace");
        REQUIRE((trace->m.rows & (trace->m.rows - 1)) == 0, "prove_air: trace length must be a power of two");
        prove_program(ctx, comm, air, trace->m.data.get(), nullptr, nullptr, (uint32_t)trace->m.cols, ilog2u(trace->m.rows), pub, n_pub, options, proof, proof_len);
    });
}
static int32_t prove_air_from_host(aero_ctx* ctx, const aero_comm* comm, const aero_air* air, const uint64_t* trace_col_major, uint32_t log_n,
                                   const uint64_t* pub, uint32_t n_pub, const aero_proof_options* options, uint8_t** proof, size_t* proof_len);
int32_t aero_prove_air_host(aero_ctx* ctx, const aero_air* air, const uint64_t* trace_col_major, uint32_t log_n, const uint64_t* pub, uint32_t n_pub,
                            const aero_proof_options* options, uint8_t** proof, size_t* proof_len) {
    return prove_air_from_host(ctx, nullptr, air, trace_col_major, log_n, pub, n_pub, options, proof, proof_len);
}
// one proof over the ranks of `comm`, every rank holding (or mapping) the same host trace: rank k copies only its share of the
// main columns (and the few the auxiliary builders read), as aero_prove_fib_sharded_host does
int32_t aero_prove_air_sharded_host(aero_ctx* ctx, const aero_comm* comm, const aero_air* air, const uint64_t* trace_col_major, uint32_t log_n,
                                    const uint64_t* pub, uint32_t n_pub, const aero_proof_options* options, uint8_t** proof, size_t* proof_len) {
    if (!comm) return AERO_E_BAD_ARG;
    if (air && options && comm->world >= 1 && log_n >= 3 && log_n <= 29) (void)aero_air_prepare(air, log_n, options, (uint32_t)comm->world);   // the ranks' kernel, once, ahead of the proof
    return prove_air_from_host(ctx, comm, air, trace_col_major, log_n, pub, n_pub, options, proof, proof_len);
}
static int32_t prove_air_from_host(aero_ctx* ctx, const aero_comm* comm, const aero_air* air, const uint64_t* trace_col_major, uint32_t log_n,
                                   const uint64_t* pub, uint32_t n_pub, const aero_proof_options* options, uint8_t** proof, size_t* proof_len) {
    return guard(ctx, [&] {
        REQUIRE(air && trace_col_major && proof && proof_len, "prove_air_host: null argument");
        REQUIRE(log_n >= 3 && log_n <= 29, "prove_air_host: log_n must be in [3, 29]");
        Context* c = ctx->c;
        const uint32_t W = air->prog.W;
        const size_t n = (size_t)1 << log_n;
        unsigned int* verdict = c->pinned_word();
        *verdict = 0;
        // the columns go straight into the interpolation buffer (Prover::set_host_trace); with auxiliary builders a device copy of the
        // main segment is kept until the auxiliary columns are built
        (void)n; (void)c;
        prove_program(ctx, comm, air, nullptr, trace_col_major, verdict, W, (int)log_n, pub, n_pub, options, proof, proof_len);
        if (*verdict != 0) {
            free(*proof);
            *proof = nullptr; *proof_len = 0;
            fail("prove_air_host: trace holds a non-canonical field element (>= p)");
        }
    });
}

}  // extern "C"

// ---- the constraint seam ---------------------------------------------------------------------------------------------------
template <class F>
static void eval_constraints_program(Context* c, const air::Program& p, const Matrix& lde, const Matrix* aux_lde, uint32_t log_blowup,
                                     const std::vector<uint64_t>& pub, const uint64_t* rands_in, const uint64_t* coeffs, uint32_t frag, uint32_t nfrags,
                                     uint64_t* out_cols, uint64_t* frag_index_out) {
    typedef typename F::T T;
    const size_t N = lde.rows, B = (size_t)1 << log_blowup, n = N / B, C = p.ce_blowup, ceN = C * n;
    REQUIRE((uint32_t)lde.cols == p.W, "eval_constraints: the trace LDE does not have the program's main width");
    REQUIRE(n * B == N && n >= 8 && (n & (n - 1)) == 0, "eval_constraints: LDE rows are not trace_length << log_blowup");
    REQUIRE(B >= C, "eval_constraints: blowup smaller than the program's constraint-evaluation blowup");
    REQUIRE(!p.A || (aux_lde && rands_in && aux_lde->rows == N && (uint32_t)aux_lde->cols == p.A * F::DEG), "eval_constraints: auxiliary LDE / random elements missing or of the wrong shape");
    REQUIRE(nfrags >= 1 && (nfrags & (nfrags - 1)) == 0 && ceN / nfrags >= 1 && frag < nfrags, "eval_constraints: bad fragment spec");
    const air::Instance in = air::instantiate(p, ilog2u(n));
    const size_t rows = ceN / nfrags, first = (size_t)frag * rows, ncols = in.num_columns() * F::DEG;
    AirCoeffs<F> cc;
    {
        const size_t nt = p.num_transition(), na = p.num_assertions();
        const std::vector<T> all = read_elems<F>(coeffs, 2 * (nt + na), "eval_constraints");
        for (size_t i = 0; i < nt; i++) { cc.ta.push_back(all[2 * i]); cc.tb.push_back(all[2 * i + 1]); }
        for (size_t i = 0; i < na; i++) { cc.ba.push_back(all[2 * (nt + i)]); cc.bb.push_back(all[2 * (nt + i) + 1]); }
    }
    const std::vector<T> rands = p.R ? read_elems<F>(rands_in, p.R, "eval_constraints") : std::vector<T>();
    AirGeometry g;
    g.lde = lde.data.get(); g.aux = p.A ? aux_lde->data.get() : nullptr; g.frame_rows = N; g.split_log = 0;
    g.rows = ceN; g.first = first; g.count = rows; g.offset = gl::GEN;
    DevBuf<uint64_t> d_out(c, ncols * rows);
    air_eval_constraints<F>(c, p, in, g, cc, pub.data(), rands.data(), 0, d_out.get(), nullptr);
    AERO_HIP(hipMemcpyAsync(out_cols, d_out.get(), ncols * rows * 8, hipMemcpyDeviceToHost, c->stream));
    c->sync();
    c->scratch_reset();
    if (frag_index_out) *frag_index_out = first;
}
template <class F>
static void aux_columns_program(aero_ctx* ctx, const air::Program& p, const Matrix& trace, const std::vector<uint64_t>& pub, const uint64_t* rands_in, aero_matrix** out) {
    typedef typename F::T T;
    Context* c = ctx->c;
    REQUIRE((uint32_t)trace.cols == p.W && (trace.rows & (trace.rows - 1)) == 0 && trace.rows >= 8, "aux_columns: the trace does not match the program");
    REQUIRE(p.A > 0, "aux_columns: the program has no auxiliary segment");
    const std::vector<T> rands = read_elems<F>(rands_in, p.R, "aux_columns");
    std::unique_ptr<aero_matrix> m(new aero_matrix(ctx));
    m->m = Matrix(c, (int)(p.A * F::DEG), trace.rows);
    air_build_aux<F>(c, p, trace.data.get(), ilog2u(trace.rows), pub.data(), rands.data(), m->m.data.get());
    c->sync();
    c->scratch_reset();
    *out = m.release();
}
template <class F>
static void composition_poly_program(aero_ctx* ctx, const air::Program& p, const uint64_t* numer_cols, uint32_t log_n, aero_matrix** out) {
    Context* c = ctx->c;
    const air::Instance in = air::instantiate(p, (int)log_n);
    const size_t n = (size_t)1 << log_n, C = p.ce_blowup, ceN = C * n, ncols = in.num_columns() * F::DEG;
    const int log_ce = ilog2u(ceN);
    for (size_t i = 0; i < ncols * ceN; i++) REQUIRE(numer_cols[i] < gl::P, "composition_poly: non-canonical element");
    DevBuf<uint64_t> d_cols(c, ncols * ceN);
    AERO_HIP(hipMemcpyAsync(d_cols.get(), numer_cols, ncols * ceN * 8, hipMemcpyHostToDevice, c->stream));
    std::unique_ptr<aero_matrix> m(new aero_matrix(ctx));
    m->m = Matrix(c, (int)(C * F::DEG), n);
    uint64_t* oh[2] = {m->m.data.get(), m->m.data.get() + (F::DEG > 1 ? ceN : 0)};
    air_divide_columns<F>(c, p, in, d_cols.get(), ceN, gl::GEN, oh);
    Prover pr(c, ProofOptions::with_96_bit_security());
    pr.composition_from_evaluations(m->m.data.get(), F::DEG, log_ce, ilog2u(C), gl::GEN);
    c->sync();
    c->scratch_reset();
    *out = m.release();
}

extern "C" {

int32_t aero_eval_constraints_program(aero_ctx* ctx, const aero_air* air, const aero_matrix* trace_lde, const aero_matrix* aux_lde, uint32_t log_blowup,
                                      const uint64_t* pub, uint32_t n_pub, const uint64_t* rands, const uint64_t* coeffs, uint8_t field_extension,
                                      uint32_t fragment_offset, uint32_t num_fragments, uint64_t* out_cols, uint64_t* frag_index_out) {
    return guard(ctx, [&] {
        REQUIRE(air && trace_lde && coeffs && out_cols, "eval_constraints: null argument");
        REQUIRE(log_blowup >= 1 && log_blowup <= 7, "eval_constraints: log_blowup must be in [1,7]");
        const std::vector<uint64_t> pubv = read_pub(air->prog, pub, n_pub, "eval_constraints");
        const Matrix* aux = aux_lde ? &aux_lde->m : nullptr;
        if (field_extension == EXT_NONE) eval_constraints_program<gl::FB>(ctx->c, air->prog, trace_lde->m, aux, log_blowup, pubv, rands, coeffs, fragment_offset, num_fragments, out_cols, frag_index_out);
        else if (field_extension == EXT_QUADRATIC) eval_constraints_program<gl::FQ>(ctx->c, air->prog, trace_lde->m, aux, log_blowup, pubv, rands, coeffs, fragment_offset, num_fragments, out_cols, frag_index_out);
        else fail("eval_constraints: field extension must be 1 (None) or 2 (Quadratic)", ST_UNSUPPORTED);
    });
}
int32_t aero_aux_columns_program(aero_ctx* ctx, const aero_air* air, const aero_matrix* trace, const uint64_t* pub, uint32_t n_pub, const uint64_t* rands,
                                 uint8_t field_extension, aero_matrix** aux_out) {
    return guard(ctx, [&] {
        REQUIRE(air && trace && rands && aux_out, "aux_columns: null argument");
        const std::vector<uint64_t> pubv = read_pub(air->prog, pub, n_pub, "aux_columns");
        if (field_extension == EXT_NONE) aux_columns_program<gl::FB>(ctx, air->prog, trace->m, pubv, rands, aux_out);
        else if (field_extension == EXT_QUADRATIC) aux_columns_program<gl::FQ>(ctx, air->prog, trace->m, pubv, rands, aux_out);
        else fail("aux_columns: field extension must be 1 (None) or 2 (Quadratic)", ST_UNSUPPORTED);
    });
}
// `Trace::validate(&air)` (what a debug build of the reference's prover runs inside commit_to_trace_and_validate, proving_worker.rs:323-332)
int32_t aero_air_validate_trace(aero_ctx* ctx, const aero_air* air, const aero_matrix* trace, const aero_matrix* aux, const uint64_t* pub, uint32_t n_pub,
                                const uint64_t* rands, uint8_t field_extension, uint64_t* first_failure) {
    return guard(ctx, [&] {
        REQUIRE(air && trace && first_failure, "validate_trace: null argument");
        const air::Program& p = air->prog;
        const std::vector<uint64_t> pubv = read_pub(p, pub, n_pub, "validate_trace");
        REQUIRE((uint32_t)trace->m.cols == p.W, "validate_trace: the trace does not have the program's main width");
        REQUIRE(trace->m.rows >= 8 && (trace->m.rows & (trace->m.rows - 1)) == 0, "validate_trace: trace length must be a power of two >= 8");
        const int deg = field_extension == EXT_QUADRATIC ? 2 : 1;
        REQUIRE(field_extension == EXT_NONE || field_extension == EXT_QUADRATIC, "validate_trace: field extension must be 1 (None) or 2 (Quadratic)");
        if (aux) {
            REQUIRE(p.A > 0 && (uint32_t)aux->m.cols == p.A * (uint32_t)deg && aux->m.rows == trace->m.rows, "validate_trace: auxiliary matrix must hold aux_width * degree component columns of the trace's length");
            REQUIRE(rands, "validate_trace: the auxiliary constraints need the random elements");
        }
        const air::Instance in = air::instantiate(p, ilog2u(trace->m.rows));
        std::vector<uint64_t> zero((size_t)p.R * 2 + 2, 0);
        const uint64_t* rv = rands ? rands : zero.data();
        if (deg == 1) *first_failure = air_validate_trace<gl::FB>(ctx->c, p, in, trace->m.data.get(), aux ? aux->m.data.get() : nullptr, pubv.data(), rv);
        else *first_failure = air_validate_trace<gl::FQ>(ctx->c, p, in, trace->m.data.get(), aux ? aux->m.data.get() : nullptr, pubv.data(), reinterpret_cast<const gl::E2*>(rv));
    });
}
int32_t aero_composition_poly_program(aero_ctx* ctx, const aero_air* air, const uint64_t* numer_cols, uint32_t log_n, uint8_t field_extension,
                                      aero_matrix** comp_polys) {
    return guard(ctx, [&] {
        REQUIRE(air && numer_cols && comp_polys, "composition_poly: null argument");
        REQUIRE(log_n >= 3 && log_n <= 28, "composition_poly: log_n out of range");
        if (field_extension == EXT_NONE) composition_poly_program<gl::FB>(ctx, air->prog, numer_cols, log_n, comp_polys);
        else if (field_extension == EXT_QUADRATIC) composition_poly_program<gl::FQ>(ctx, air->prog, numer_cols, log_n, comp_polys);
        else fail("composition_poly: field extension must be 1 (None) or 2 (Quadratic)", ST_UNSUPPORTED);
    });
}

// constraints_worker.rs:14-79: the work item's trace LDE (main columns + one auxiliary segment), composition coefficients, auxiliary
// random elements and fragment -> the fragment's numerator columns, one per divisor. The AIR is the program (the message carries
// none: the reference's worker hard-wires ProcessorAir); PUB operands read `pub`, or by default the elements of the Miden
// PublicInputs inside the message (program hash || stack inputs || outputs.stack || overflow addresses).
int32_t aero_worker_eval_constraints(aero_ctx* ctx, const uint8_t* work_item, size_t work_item_len, const aero_air* air, const uint64_t* pub,
                                     uint32_t n_pub, uint8_t** result, size_t* result_len) {
    return guard(ctx, [&] {
        REQUIRE(work_item && air && result && result_len, "worker_eval_constraints: null argument");
        *result = nullptr; *result_len = 0;
        const air::Program& p = air->prog;
        const wm::ConstraintWorkItem w = wm::parse_constraint_work_item(work_item, work_item_len);
        const uint32_t W = w.main_width, A = w.aux_width;
        REQUIRE(W == p.W && A == p.A && (!A || w.aux_rands == p.R), "worker_eval_constraints: the trace layout of the message is not the program's");
        REQUIRE(w.main_cols.size() == W, "worker_eval_constraints: the main segment does not have the width the layout names");
        REQUIRE(w.trace_len >= 8 && (w.trace_len & (w.trace_len - 1)) == 0 && w.trace_len <= ((uint64_t)1 << 29), "worker_eval_constraints: trace length must be a power of two in [8, 2^29]");
        REQUIRE(w.blowup >= 2 && w.blowup <= 128 && (w.blowup & (w.blowup - 1)) == 0 && w.blowup == w.options[1], "worker_eval_constraints: blowup of the LDE and of the proof options disagree");
        if (w.options[4] != EXT_NONE) fail("worker_eval_constraints: the message carries base-field coefficients (field extension must be None)", ST_UNSUPPORTED);
        const size_t N = (size_t)w.trace_len * w.blowup;
        for (const auto& col : w.main_cols) REQUIRE(col.n == N, "worker_eval_constraints: an LDE column is not trace_length * blowup long");
        if (A) {
            REQUIRE(w.aux_segments.size() == 1 && w.aux_segments[0].size() == A, "worker_eval_constraints: expected one auxiliary segment of the layout's width");
            REQUIRE(w.aux_rand_elements.size() == 1 && w.aux_rand_elements[0].size() == w.aux_rands, "worker_eval_constraints: auxiliary random elements do not match the layout");
            for (const auto& col : w.aux_segments[0]) REQUIRE(col.n == N, "worker_eval_constraints: an auxiliary LDE column is not trace_length * blowup long");
        } else {
            for (const auto& seg : w.aux_segments) REQUIRE(seg.empty(), "worker_eval_constraints: auxiliary columns without an auxiliary layout");
        }
        REQUIRE(w.n_transition == p.num_transition() && w.n_boundary == p.num_assertions(), "worker_eval_constraints: coefficient counts do not match the program (one pair per transition constraint, one per assertion)");
        std::vector<uint64_t> pubv;
        if (pub || n_pub) pubv = read_pub(p, pub, n_pub, "worker_eval_constraints");
        else {
            fmt::MidenInputs mi;
            try { mi = fmt::parse_miden_inputs(w.public_inputs.data(), w.public_inputs.size()); }
            catch (const std::exception& e) { fail(std::string("worker message: public inputs: ") + e.what()); }
            for (auto* v : {&mi.hash, &mi.stack_inputs, &mi.out_stack, &mi.overflow}) pubv.insert(pubv.end(), v->begin(), v->end());
            REQUIRE(pubv.size() == p.num_pub, "worker_eval_constraints: the message's public inputs do not have the number of elements the program reads");
            for (uint64_t v : pubv) REQUIRE(v < gl::P, "worker_eval_constraints: non-canonical public input");
        }
        REQUIRE(w.num_fragments >= 1 && w.num_fragments <= ((uint64_t)1 << 30) && w.fragment_offset < w.num_fragments, "worker_eval_constraints: bad fragment");
        Context* c = ctx->c;
        Matrix lde(c, (int)W, N), aux;
        // the columns go to the device straight from the message; Felt::new's reduction of raw values happens there
        for (uint32_t col = 0; col < W; col++)
            AERO_HIP(hipMemcpyAsync(lde.data.get() + (size_t)col * N, w.main_cols[col].data, N * 8, hipMemcpyHostToDevice, c->stream));
        reduce_canonical(c, lde.data.get(), (size_t)W * N);
        if (A) {
            aux = Matrix(c, (int)A, N);
            for (uint32_t col = 0; col < A; col++)
                AERO_HIP(hipMemcpyAsync(aux.data.get() + (size_t)col * N, w.aux_segments[0][col].data, N * 8, hipMemcpyHostToDevice, c->stream));
            reduce_canonical(c, aux.data.get(), (size_t)A * N);
        }
        const size_t ceN = (size_t)p.ce_blowup * (size_t)w.trace_len;
        REQUIRE(ceN % w.num_fragments == 0, "worker_eval_constraints: the fragments do not divide the constraint domain");
        const size_t rows = ceN / (size_t)w.num_fragments;
        const size_t ncols = air::instantiate(p, ilog2u(w.trace_len)).num_columns();
        std::vector<uint64_t> cols(ncols * rows);
        uint64_t first = 0;
        eval_constraints_program<gl::FB>(c, p, lde, A ? &aux : nullptr, (uint32_t)ilog2u(w.blowup), pubv, A ? w.aux_rand_elements[0].data() : nullptr,
                                         w.coeffs.data(), (uint32_t)w.fragment_offset, (uint32_t)w.num_fragments, cols.data(), &first);
        const std::vector<uint8_t> msg = wm::emit_constraint_result(first, w.num_fragments, cols.data(), ncols, rows);
        *result = to_malloc(msg, result_len);
    });
}

}  // extern "C"
