// Device interpreter of compiled AEROAIR programs (air_program.hpp) for gfx950.
//
// Reference seam replaced: `ConstraintEvaluator::evaluate_fragment(&trace_lde, &domain, &mut fragment)` for ANY `Air`
// (/root/reference/aero-sdk/miden-wasm/src/constraints_worker.rs:32-59; divisor list and fragment stitching
// proving_worker.rs:374-437) and the `build_aux_segment` step of `commit_to_trace_and_validate` (proving_worker.rs:323-332).
//
// Execution model. One lane = one row of the constraint-evaluation domain; one wavefront = one workgroup, so control flow is
// the PROGRAM's control flow and uniform by construction: instruction words, scalar operands (constants, public inputs, random
// elements, folded nodes), composition coefficients and table offsets are the same address in every lane and are fetched with
// scalar loads; the opcode dispatch is a scalar branch. Per-lane state: the row's frame (read straight from the column-major LDE,
// lane <-> row, coalesced; the trace is never staged), and a register file in LDS - slot s of lane l at word s * 64 + l, one
// 8-byte bank-conflict-free access per operand - sized by the host's register allocation, not by the program length. Transition
// constraints accumulate into two registers per degree group (sum alpha_k t_k, sum beta_k t_k); the group's degree adjustment
// x^adj is ONE lookup in the two-level table of the domain generator (x^adj = offset^adj * w^(s adj mod rows), offset^adj folded
// into the beta coefficients on the host). Boundary divisors x^a - b are batch-inverted over the K rows a lane owns
// (Montgomery's trick through LDS, any number of divisors) before the rows are interpreted, so the fused division costs one
// field inversion per K rows.
#include "air_kernels.hpp"

namespace aero {

using gl::FB;
using gl::FQ;
using namespace air;

constexpr int AIR_WG = 64;

__device__ __forceinline__ uint64_t air_tw(const uint64_t* __restrict__ lo, const uint64_t* __restrict__ hi, uint64_t e, int h) {
    return gl::mul(lo[e & ((1ull << h) - 1)], hi[e >> h]);
}

template <class F> struct AirRow {
    typedef typename F::T T;
    const uint64_t* main;
    const uint64_t* aux;
    size_t stride, r, rn;
    uint32_t s;                  // row index for the periodic tables
    const uint64_t* scalB;
    const T* scalE;
    const uint64_t* ptab;
    const uint32_t *p_off, *p_mask;
    uint64_t *ldsB, *ldsE;       // already offset by the lane
};
template <class F> __device__ __forceinline__ uint64_t air_fetch_b(const AirRow<F>& c, uint32_t kind, uint32_t idx) {
    switch (kind) {
        case D_SLOT_B: return c.ldsB[idx * AIR_WG];
        case D_MAIN_CUR: return c.main[(size_t)idx * c.stride + c.r];
        case D_MAIN_NXT: return c.main[(size_t)idx * c.stride + c.rn];
        case D_PERIODIC: return c.ptab[c.p_off[idx] + (c.s & c.p_mask[idx])];
        default: return c.scalB[idx];
    }
}
template <class F> __device__ __forceinline__ typename F::T air_fetch_e(const AirRow<F>& c, uint32_t kind, uint32_t idx) {
    switch (kind) {
        case D_SLOT_E: return F::make(c.ldsE[(idx * F::DEG) * AIR_WG], F::DEG > 1 ? c.ldsE[(idx * F::DEG + 1) * AIR_WG] : 0);
        case D_AUX_CUR: { const size_t o = (size_t)(idx * F::DEG) * c.stride + c.r; return F::make(c.aux[o], F::DEG > 1 ? c.aux[o + c.stride] : 0); }
        case D_AUX_NXT: { const size_t o = (size_t)(idx * F::DEG) * c.stride + c.rn; return F::make(c.aux[o], F::DEG > 1 ? c.aux[o + c.stride] : 0); }
        case D_SCAL_E: return c.scalE[idx];
        default: return F::from(air_fetch_b<F>(c, kind, idx));
    }
}
template <class F> __device__ __forceinline__ void air_store_e(const AirRow<F>& c, uint32_t slot, typename F::T v) {
    c.ldsE[(slot * F::DEG) * AIR_WG] = F::comp(v, 0);
    if (F::DEG > 1) c.ldsE[(slot * F::DEG + 1) * AIR_WG] = F::comp(v, 1);
}
// the arithmetic opcodes (OP_ADD_B .. OP_MULB_E)
template <class F> __device__ __forceinline__ void air_arith(const AirRow<F>& c, uint32_t op, uint32_t ka, uint32_t kb, uint32_t dst, uint32_t ia, uint32_t ib) {
    typedef typename F::T T;
    if (op <= OP_MUL_B) {
        const uint64_t a = air_fetch_b<F>(c, ka, ia), b = air_fetch_b<F>(c, kb, ib);
        c.ldsB[dst * AIR_WG] = op == OP_ADD_B ? gl::add(a, b) : op == OP_SUB_B ? gl::sub(a, b) : gl::mul(a, b);
    } else if (op == OP_MULB_E) {
        const T a = air_fetch_e<F>(c, ka, ia);
        air_store_e<F>(c, dst, F::mulb(a, air_fetch_b<F>(c, kb, ib)));
    } else {
        const T a = air_fetch_e<F>(c, ka, ia), b = air_fetch_e<F>(c, kb, ib);
        air_store_e<F>(c, dst, op == OP_ADD_E ? F::add(a, b) : op == OP_SUB_E ? F::sub(a, b) : F::mul(a, b));
    }
}

// MODE 0: numerator columns (the reference's ConstraintEvaluationTable seam); MODE 1: divided by the divisors and summed (H)
template <class F, int MODE> __global__ __launch_bounds__(AIR_WG) void air_constraints_kernel(AirConsArgs<F> a, uint32_t K) {
    typedef typename F::T T;
    extern __shared__ uint64_t air_lds[];
    const uint32_t lane = threadIdx.x;
    const size_t nthreads = a.count / K;
    size_t t = (size_t)blockIdx.x * AIR_WG + lane;
    const bool active = t < nthreads;
    if (!active) t = nthreads - 1;        // the wavefront stays convergent: idle lanes shadow the last row and store nothing
    const uint64_t rmask = a.rows - 1;
    const uint32_t nb = a.n_bgroups;
    AirRow<F> c;
    c.main = a.lde; c.aux = a.aux; c.stride = a.N; c.scalB = a.scalB; c.scalE = a.scalE; c.ptab = a.ptab; c.p_off = a.p_off; c.p_mask = a.p_mask;
    c.ldsB = air_lds + lane;
    c.ldsE = air_lds + (size_t)a.slotsB * AIR_WG + lane;
    uint64_t* const ldsD = air_lds + ((size_t)a.slotsB + (size_t)a.slotsE * F::DEG) * AIR_WG + lane;   // K * nb inverse divisors
    uint64_t* const ldsP = ldsD + (size_t)K * nb * AIR_WG;                                              // prefix products
    if (MODE == 1 && nb) {
        // boundary divisors x^a - b of the K rows, inverted with ONE field inversion
        uint64_t run = 1;
        for (uint32_t q = 0; q < K; q++) {
            const uint64_t s = a.first + t + (uint64_t)q * nthreads;
            for (uint32_t j = 0; j < nb; j++) {
                const AirBGroupDev& g = a.bgroups[j];
                const uint64_t d = gl::sub(gl::mul(g.ha, air_tw(a.tw_lo, a.tw_hi, (s * g.a_exp) & rmask, a.tw_h)), g.b);
                const uint32_t i = q * nb + j;
                ldsD[i * AIR_WG] = d;
                ldsP[i * AIR_WG] = run;
                run = gl::mul(run, d);
            }
        }
        uint64_t ia = gl::inv(run);
        for (uint32_t i = K * nb; i-- > 0;) {
            const uint64_t d = ldsD[i * AIR_WG];
            ldsD[i * AIR_WG] = gl::mul(ia, ldsP[i * AIR_WG]);
            ia = gl::mul(ia, d);
        }
    }
    const uint4* const code = reinterpret_cast<const uint4*>(a.code);
#pragma unroll 1
    for (uint32_t q = 0; q < K; q++) {
        const uint64_t s = a.first + t + (uint64_t)q * nthreads;
        size_t r = (size_t)s * a.ce_step;
        size_t rn = (r + a.blowup) & (a.N - 1);
        if (a.split_log) {
            const size_t part_len = a.N >> a.split_log, pm = ((size_t)1 << a.split_log) - 1;
            r = (r & pm) * part_len + (r >> a.split_log);
            rn = (rn & pm) * part_len + (rn >> a.split_log);
        }
        c.r = r; c.rn = rn; c.s = (uint32_t)s;
        // ---- transition constraints: the program
        T acc_a = F::zero(), acc_b = F::zero(), total = F::zero();
        uint4 I = code[0];
#pragma unroll 1
        for (uint32_t pc = 1;; pc++) {
            const uint4 In = code[pc];            // the stream ends with two END words: the look-ahead stays inside it
            const uint32_t op = I.x & 0xff, ka = (I.x >> 8) & 0xf, kb = (I.x >> 12) & 0xf;
            if (op == OP_END) break;
            if (op <= OP_MULB_E) {
                air_arith<F>(c, op, ka, kb, I.y, I.z, I.w);
            } else if (op == OP_EMIT_B) {
                const uint64_t v = air_fetch_b<F>(c, ka, I.z);
                acc_a = F::add(acc_a, F::mulb(a.ta[I.y], v));
                acc_b = F::add(acc_b, F::mulb(a.tb[I.y], v));
            } else if (op == OP_EMIT_E) {
                const T v = air_fetch_e<F>(c, ka, I.z);
                acc_a = F::add(acc_a, F::mul(a.ta[I.y], v));
                acc_b = F::add(acc_b, F::mul(a.tb[I.y], v));
            } else {                              // OP_GROUP_END
                total = F::add(total, F::mulb(acc_b, air_tw(a.tw_lo, a.tw_hi, (s * a.dg_exp[I.y]) & rmask, a.tw_h)));
                acc_b = F::zero();
            }
            I = In;
        }
        total = F::add(total, acc_a);
        const size_t o = (size_t)(s - a.first);
        T h = F::zero();
        if (MODE == 0) {
            if (active) for (int d = 0; d < F::DEG; d++) a.out_cols[(size_t)d * a.count + o] = F::comp(total, d);
        } else {
            // 1 / ((x^n - 1) / prod (x - w^(n-i))) = prod (x - w^(n-i)) * (x^n - 1)^-1
            const uint64_t x = gl::mul(a.offset, air_tw(a.tw_lo, a.tw_hi, s & rmask, a.tw_h));
            uint64_t tdiv = a.zn_inv[s & a.xmask];
            for (uint32_t i = 0; i < a.n_exempt; i++) tdiv = gl::mul(tdiv, gl::sub(x, a.exempt[i]));
            h = F::mulb(total, tdiv);
        }
        // ---- boundary constraints, one group per divisor
        for (uint32_t j = 0; j < nb; j++) {
            const AirBGroupDev& g = a.bgroups[j];
            T sa = F::zero(), sb = F::zero();
            for (uint32_t m = g.m0; m < g.m0 + g.count; m++) {
                const BoundaryMember bm = a.members[m];
                if (!bm.aux && !bm.val_ext) {
                    const uint64_t d = gl::sub(a.lde[(size_t)bm.col * a.N + r], a.scalB[bm.val_idx]);
                    sa = F::add(sa, F::mulb(a.ba[bm.coef], d));
                    sb = F::add(sb, F::mulb(a.bb[bm.coef], d));
                } else {
                    const T v = bm.aux ? air_fetch_e<F>(c, D_AUX_CUR, bm.col) : F::from(a.lde[(size_t)bm.col * a.N + r]);
                    const T d = F::sub(v, bm.val_ext ? a.scalE[bm.val_idx] : F::from(a.scalB[bm.val_idx]));
                    sa = F::add(sa, F::mul(a.ba[bm.coef], d));
                    sb = F::add(sb, F::mul(a.bb[bm.coef], d));
                }
            }
            const T gnum = F::add(sa, F::mulb(sb, air_tw(a.tw_lo, a.tw_hi, (s * g.adj_exp) & rmask, a.tw_h)));
            if (MODE == 0) {
                if (active) for (int d = 0; d < F::DEG; d++) a.out_cols[(size_t)((1 + j) * F::DEG + d) * a.count + o] = F::comp(gnum, d);
            } else {
                h = F::add(h, F::mulb(gnum, ldsD[(q * nb + j) * AIR_WG]));
            }
        }
        if (MODE == 1 && active) for (int d = 0; d < F::DEG; d++) a.out_h[d][s] = F::comp(h, d);
    }
}

static void air_set_lds(const void* kern, size_t bytes) {
    if (bytes > 160 * 1024) fail("air program: the interpreter's register file does not fit the 160 KiB of LDS", ST_UNSUPPORTED);
    if (bytes > 48 * 1024) AERO_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

template <class F> bool launch_air_constraints(Context* ctx, const AirConsArgs<F>& a, int mode) {
    const size_t cnt = a.count;
    const size_t in_cols = (size_t)a.W + (size_t)a.A * F::DEG;
    const size_t slots = (size_t)a.slotsB + (size_t)a.slotsE * F::DEG;
    if (mode == 0) {
        const size_t lds = (slots ? slots : 1) * AIR_WG * 8;
        air_set_lds((const void*)air_constraints_kernel<F, 0>, lds);
        AERO_LAUNCH(ctx, "air_constraints_kernel", cnt * 8 * (in_cols + (1 + a.n_bgroups) * F::DEG), (air_constraints_kernel<F, 0>),
                    dim3((unsigned)((cnt + AIR_WG - 1) / AIR_WG)), dim3(AIR_WG), lds, a, 1u);
    } else {
        // rows per lane sharing one batched inversion: 4 while the inverse-divisor block stays small, 1 when the count does not divide
        uint32_t K = cnt % 4 == 0 && cnt >= 4 * AIR_WG ? 4 : 1;
        if (a.n_bgroups > 8) K = 1;
        const size_t lds = (slots + 2 * (size_t)K * a.n_bgroups + 1) * AIR_WG * 8;
        if (lds > 160 * 1024) return false;
        air_set_lds((const void*)air_constraints_kernel<F, 1>, lds);
        AERO_LAUNCH(ctx, "air_constraints_kernel", cnt * 8 * (in_cols + F::DEG), (air_constraints_kernel<F, 1>),
                    dim3((unsigned)((cnt / K + AIR_WG - 1) / AIR_WG)), dim3(AIR_WG), lds, a, K);
    }
    ctx->check_launch("air_constraints");
    return true;
}
template bool launch_air_constraints<FB>(Context*, const AirConsArgs<FB>&, int);
template bool launch_air_constraints<FQ>(Context*, const AirConsArgs<FQ>&, int);

// ------------------------------------------------------------------------------------------------
// Unfused division (stage entry point; fallback for programs with many boundary divisors): one inversion per row over the
// product of its boundary divisors.
constexpr int AIR_MAX_BGROUPS = 64;
template <class F> __global__ __launch_bounds__(256) void air_divide_kernel(AirDivideArgs<F> a) {
    typedef typename F::T T;
    const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.rows) return;
    const uint64_t rmask = a.rows - 1;
    const uint64_t x = gl::mul(a.offset, air_tw(a.tw_lo, a.tw_hi, s & rmask, a.tw_h));
    uint64_t tdiv = a.zn_inv[s & a.xmask];
    for (uint32_t i = 0; i < a.n_exempt; i++) tdiv = gl::mul(tdiv, gl::sub(x, a.exempt[i]));
    auto col = [&](uint32_t j) { return F::make(a.cols[(size_t)(j * F::DEG) * a.rows + s], F::DEG > 1 ? a.cols[(size_t)(j * F::DEG + 1) * a.rows + s] : 0); };
    T h = F::mulb(col(0), tdiv);
    uint64_t den[AIR_MAX_BGROUPS], pre[AIR_MAX_BGROUPS];
    uint64_t run = 1;
    for (uint32_t j = 0; j < a.n_bgroups; j++) {
        const AirBGroupDev& g = a.bgroups[j];
        den[j] = gl::sub(gl::mul(g.ha, air_tw(a.tw_lo, a.tw_hi, (s * g.a_exp) & rmask, a.tw_h)), g.b);
        pre[j] = run;
        run = gl::mul(run, den[j]);
    }
    uint64_t ia = gl::inv(run);
    for (uint32_t j = a.n_bgroups; j-- > 0;) {
        h = F::add(h, F::mulb(col(1 + j), gl::mul(ia, pre[j])));
        ia = gl::mul(ia, den[j]);
    }
    for (int d = 0; d < F::DEG; d++) a.out_h[d][s] = F::comp(h, d);
}
template <class F> void launch_air_divide(Context* ctx, const AirDivideArgs<F>& a) {
    if (a.n_bgroups > AIR_MAX_BGROUPS) fail("air program: more than 64 boundary divisors", ST_UNSUPPORTED);
    AERO_LAUNCH(ctx, "air_divide_kernel", a.rows * 8 * F::DEG * (2 + a.n_bgroups), (air_divide_kernel<F>), dim3((unsigned)((a.rows + 255) / 256)), dim3(256), 0, a);
    ctx->check_launch("air_divide");
}
template void launch_air_divide<FB>(Context*, const AirDivideArgs<FB>&);
template void launch_air_divide<FQ>(Context*, const AirDivideArgs<FQ>&);

// ------------------------------------------------------------------------------------------------
// Auxiliary segment. Pass 1: the builder program per trace row -> factor columns (2c = numerator factor, 2c + 1 = denominator
// factor of aux column c). Pass 2: exclusive prefix products over the rows (per-block totals, scan of the totals, apply), the
// denominators inverted with one inversion per 8 rows.
template <class F> __global__ __launch_bounds__(AIR_WG) void air_aux_factors_kernel(AirAuxArgs<F> a, uint64_t* fac) {
    typedef typename F::T T;
    extern __shared__ uint64_t air_lds[];
    const uint32_t lane = threadIdx.x;
    size_t i = (size_t)blockIdx.x * AIR_WG + lane;
    const bool active = i < a.n;
    if (!active) i = a.n - 1;
    AirRow<F> c;
    c.main = a.trace; c.aux = nullptr; c.stride = a.n; c.r = i; c.rn = (i + 1) & (a.n - 1); c.s = (uint32_t)i;
    c.scalB = a.scalB; c.scalE = a.scalE; c.ptab = a.ptab; c.p_off = a.p_off; c.p_mask = a.p_mask;
    c.ldsB = air_lds + lane;
    c.ldsE = air_lds + (size_t)a.slotsB * AIR_WG + lane;
    const uint4* const code = reinterpret_cast<const uint4*>(a.code);
    uint4 I = code[0];
#pragma unroll 1
    for (uint32_t pc = 1;; pc++) {
        const uint4 In = code[pc];
        const uint32_t op = I.x & 0xff, ka = (I.x >> 8) & 0xf, kb = (I.x >> 12) & 0xf;
        if (op == OP_END) break;
        if (op <= OP_MULB_E) {
            air_arith<F>(c, op, ka, kb, I.y, I.z, I.w);
        } else {                                  // OP_OUT_B / OP_OUT_E
            const T v = op == OP_OUT_B ? F::from(air_fetch_b<F>(c, ka, I.z)) : air_fetch_e<F>(c, ka, I.z);
            if (active) for (int d = 0; d < F::DEG; d++) fac[((size_t)I.y * F::DEG + d) * a.n + i] = F::comp(v, d);
        }
        I = In;
    }
}
constexpr int SCAN_K = 8;
template <class F> __device__ __forceinline__ typename F::T air_wg_scan_mul(typename F::T v, typename F::T* sh) {   // inclusive, 256 lanes
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        typename F::T u = sh[t];
        if (t >= off) u = F::mul(sh[t - off], u);
        __syncthreads();
        sh[t] = u;
        __syncthreads();
    }
    return sh[t];
}
template <class F> __device__ __forceinline__ typename F::T air_fac_at(const uint64_t* fac, size_t n, uint32_t j, size_t i) {
    if (i >= n) return F::one();
    const size_t o = (size_t)(j * F::DEG) * n + i;
    return F::make(fac[o], F::DEG > 1 ? fac[o + n] : 0);
}
// grid (blocks, 2A): product of each block's rows of factor column j
template <class F> __global__ __launch_bounds__(256) void air_fac_totals_kernel(const uint64_t* fac, size_t n, const uint8_t* has_den, typename F::T* totals) {
    typedef typename F::T T;
    __shared__ T sh[256];
    const uint32_t j = blockIdx.y;
    if ((j & 1) && !has_den[j >> 1]) return;
    const size_t first = ((size_t)blockIdx.x * 256 + threadIdx.x) * SCAN_K;
    T p = F::one();
#pragma unroll
    for (int k = 0; k < SCAN_K; k++) p = F::mul(p, air_fac_at<F>(fac, n, j, first + k));
    const T inc = air_wg_scan_mul<F>(p, sh);
    if (threadIdx.x == 255) totals[(size_t)j * gridDim.x + blockIdx.x] = inc;
}
// in place: totals[j][b] <- product of totals[j][0..b) (one workgroup per factor column)
template <class F> __global__ __launch_bounds__(256) void air_scan_totals_kernel(typename F::T* totals, uint32_t nblk, const uint8_t* has_den) {
    typedef typename F::T T;
    __shared__ T sh[256];
    const uint32_t j = blockIdx.x;
    if ((j & 1) && !has_den[j >> 1]) return;
    T* row = totals + (size_t)j * nblk;
    const uint32_t per = (nblk + 255) / 256, lo = threadIdx.x * per;
    T p = F::one();
    for (uint32_t i = lo; i < lo + per && i < nblk; i++) p = F::mul(p, row[i]);
    air_wg_scan_mul<F>(p, sh);
    T run = threadIdx.x == 0 ? F::one() : sh[threadIdx.x - 1];
    for (uint32_t i = lo; i < lo + per && i < nblk; i++) { const T v = row[i]; row[i] = run; run = F::mul(run, v); }
}
// grid (blocks, A): column c = init * prefix(num) / prefix(den)
template <class F> __global__ __launch_bounds__(256) void air_fac_apply_kernel(const uint64_t* fac, size_t n, const uint8_t* has_den, const typename F::T* init,
                                                                            const typename F::T* totals, uint64_t* out) {
    typedef typename F::T T;
    __shared__ T sh[256];
    const uint32_t c = blockIdx.y;
    const size_t first = ((size_t)blockIdx.x * 256 + threadIdx.x) * SCAN_K;
    T fn[SCAN_K];
    T p = F::one();
#pragma unroll
    for (int k = 0; k < SCAN_K; k++) { fn[k] = air_fac_at<F>(fac, n, 2 * c, first + k); p = F::mul(p, fn[k]); }
    air_wg_scan_mul<F>(p, sh);
    T run = F::mul(init[c], totals[(size_t)(2 * c) * gridDim.x + blockIdx.x]);
    if (threadIdx.x) run = F::mul(run, sh[threadIdx.x - 1]);
    T val[SCAN_K];
#pragma unroll
    for (int k = 0; k < SCAN_K; k++) { val[k] = run; run = F::mul(run, fn[k]); }
    if (has_den[c]) {
        __syncthreads();
        T fd[SCAN_K];
        T pd = F::one();
#pragma unroll
        for (int k = 0; k < SCAN_K; k++) { fd[k] = air_fac_at<F>(fac, n, 2 * c + 1, first + k); pd = F::mul(pd, fd[k]); }
        air_wg_scan_mul<F>(pd, sh);
        T rd = totals[(size_t)(2 * c + 1) * gridDim.x + blockIdx.x];
        if (threadIdx.x) rd = F::mul(rd, sh[threadIdx.x - 1]);
        // rd = prefix before this lane's first row; 1 / prefix_k = (1 / prefix_8) * fd[k] * ... * fd[7]
        T inv = F::inv(F::mul(rd, pd));
#pragma unroll
        for (int k = SCAN_K - 1; k >= 0; k--) { inv = F::mul(inv, fd[k]); val[k] = F::mul(val[k], inv); }
    }
#pragma unroll
    for (int k = 0; k < SCAN_K; k++)
        if (first + k < n)
            for (int d = 0; d < F::DEG; d++) out[((size_t)c * F::DEG + d) * n + first + k] = F::comp(val[k], d);
}
template <class F> void launch_air_aux(Context* ctx, const AirAuxArgs<F>& a, const std::vector<uint8_t>& has_den_host) {
    typedef typename F::T T;
    const size_t n = a.n;
    const uint32_t A = a.A;
    uint64_t* fac = (uint64_t*)ctx->scratch_alloc((size_t)2 * A * F::DEG * n * 8);
    const size_t slots = (size_t)a.slotsB + (size_t)a.slotsE * F::DEG;
    const size_t lds = (slots ? slots : 1) * AIR_WG * 8;
    air_set_lds((const void*)air_aux_factors_kernel<F>, lds);
    AERO_LAUNCH(ctx, "air_aux_kernel", (size_t)n * 8 * (a.W + 2 * A * F::DEG), (air_aux_factors_kernel<F>), dim3((unsigned)((n + AIR_WG - 1) / AIR_WG)), dim3(AIR_WG), lds, a, fac);
    const uint32_t nblk = (uint32_t)((n + (size_t)SCAN_K * 256 - 1) / ((size_t)SCAN_K * 256));
    T* totals = (T*)ctx->scratch_alloc(sizeof(T) * (size_t)2 * A * nblk);
    (void)has_den_host;
    AERO_LAUNCH(ctx, "air_aux_kernel", (size_t)A * n * 8 * F::DEG, (air_fac_totals_kernel<F>), dim3(nblk, 2 * A), dim3(256), 0, fac, n, a.has_den, totals);
    AERO_LAUNCH(ctx, "air_aux_kernel", 0, (air_scan_totals_kernel<F>), dim3(2 * A), dim3(256), 0, totals, nblk, a.has_den);
    AERO_LAUNCH(ctx, "air_aux_kernel", (size_t)A * n * 8 * 2 * F::DEG, (air_fac_apply_kernel<F>), dim3(nblk, A), dim3(256), 0, fac, n, a.has_den, a.init, totals, a.out);
    ctx->check_launch("air_aux");
}
template void launch_air_aux<FB>(Context*, const AirAuxArgs<FB>&, const std::vector<uint8_t>&);
template void launch_air_aux<FQ>(Context*, const AirAuxArgs<FQ>&, const std::vector<uint8_t>&);

}  // namespace aero
