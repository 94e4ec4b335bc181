// Per-row STARK kernels for gfx950: constraint evaluation, composition division, out-of-domain evaluation,
// DEEP composition, FRI folding, proof-of-work grinding and query gathers.
//
// Reference seams replaced (the bodies are winter-prover / winter-fri 0.4, absent from the mount; call sites):
//   constraint evaluation ... /root/reference/aero-sdk/miden-wasm/src/constraints_worker.rs:32-76
//                             (ConstraintEvaluator::evaluate_fragment -> per-divisor numerator columns),
//                             proving_worker.rs:374-437 (divisor list, fragment stitching)
//   composition / OOD / DEEP / FRI / grind / queries ... proving_worker.rs:344-352 (prove_after_constraint_eval);
//                             formulas mirrored by the verifier: src/stark_verifier/composer.cairo:48-316,
//                             fri/fri_verifier.cairo:243-340, crypto/random.cairo:282-316
// All kernels are one-thread-per-row(-group), column-major coalesced reads, no inter-workgroup communication.
// Field inversions are batched per thread (Montgomery trick over the thread's K rows).
#include "aero_internal.hpp"
#include "stark_kernels.hpp"
#include "dft_small.hpp"

namespace aero {

using gl::FB;
using gl::FQ;

__device__ __forceinline__ uint64_t tw2(const uint64_t* __restrict__ lo, const uint64_t* __restrict__ hi, uint32_t e, int h) {
    return gl::mul(lo[e & ((1u << h) - 1)], hi[e >> h]);
}
template <class F> __device__ __forceinline__ typename F::T ld(const uint64_t* const* comp, size_t i) {
    return F::make(comp[0][i], F::DEG > 1 ? comp[1][i] : 0);
}

// ------------------------------------------------------------------------------------------------
// FibAir constraint evaluation over the constraint-evaluation domain (ce_n = C*n points, row s <-> LDE row
// s*ce_step, x = 7 w_ce^s). Frame = (LDE row r, LDE row r + blowup mod N).
//   MODE 0: write the three per-divisor numerator columns (the reference's ConstraintEvaluationTable seam)
//   MODE 1: additionally divide by the divisors and write H(x) = sum_i col_i / div_i (fused; columns not stored)
// WIDE: the sums over the columns as 160-bit integers reduced once (gl::Wide) - pays from a few columns on (2^20 x 72: 0.87 -> 0.56 ms,
// F_p^2 0.50 -> 0.28 ms at 2^18 x 72) and costs registers that a 2-column trace has better use for (0.093 -> 0.126 ms).
template <class F, int MODE, int K, bool WIDE>
__global__ __launch_bounds__(256) void fib_constraints_kernel(FibConsArgs<F> a) {
    typedef typename F::T T;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t nthreads = a.count / K;       // a.count rows handled by this launch (fragment)
    if (t >= nthreads) return;
    T num[K][3];
    uint64_t xs[K];
#pragma unroll
    for (int q = 0; q < K; q++) {
        const size_t s = a.first + t + (size_t)q * nthreads;   // ce step
        size_t r = s * a.ce_step;
        size_t rn = (r + a.blowup) & (a.N - 1);
        if (a.split_log) {      // physical position of (compact) row j: part j mod 2^k, index j / 2^k
            const size_t part_len = a.N >> a.split_log, pm = ((size_t)1 << a.split_log) - 1;
            r = (r & pm) * part_len + (r >> a.split_log);
            rn = (rn & pm) * part_len + (rn >> a.split_log);
        }
        // degree adjustments without exponentiation: x^ce_n = 7^ce_n is constant on the coset
        //   x^adj_t = x^(ce_n - 1)     = K7 * x^-1
        //   x^adj_b = x^(ce_n - n + 1) = K7 * x * (x^n)^-1,   x^n = 7^n * w_C^(s mod C)
        // - functions of the domain point only: with the per-shape table (MODE 1) they are two loads, and x itself is needed by the auxiliary group only
        uint64_t x = 0, xt, xb;
        const bool tab = MODE == 1 && a.inv_tab != nullptr;
        if (!tab || a.A) x = gl::mul(a.offset, tw2(a.tw_lo, a.tw_hi, (uint32_t)s, a.tw_h));       // h w_ce^s
        if (tab) { xt = a.inv_tab[3 * a.inv_tab_n + s]; xb = a.inv_tab[4 * a.inv_tab_n + s]; }
        else {
            const uint64_t xinv = gl::mul(a.gen_inv, tw2(a.twi_lo, a.twi_hi, (uint32_t)s, a.tw_h));    // (h w_ce^s)^-1
            xt = gl::mul(a.k7, xinv);
            xb = gl::mul(gl::mul(a.k7, x), a.xn_inv[s & a.xmask]);
        }
        xs[q] = x;
        T acc = F::zero(), g0 = F::zero(), g1 = F::zero();
        if (WIDE) {
            // sum_k (alpha_k + beta_k x^adj) c_k = sum alpha_k c_k + x^adj sum beta_k c_k: the two sums per divisor are accumulated as
            // 160-bit integers (gl::Wide) and reduced once - 9 instructions per term and sum instead of two field multiplications and additions
            gl::Wide wa[F::DEG], wb[F::DEG], w0a[F::DEG], w0b[F::DEG], w1a[F::DEG], w1b[F::DEG];
#pragma unroll
            for (int d = 0; d < F::DEG; d++) wa[d] = wb[d] = w0a[d] = w0b[d] = w1a[d] = w1b[d] = gl::wzero();
            for (uint32_t k = 0; k < a.W / 2; k++) {
                const uint64_t ca = a.lde[(size_t)(2 * k) * a.N + r], cb = a.lde[(size_t)(2 * k + 1) * a.N + r];
                const uint64_t na = a.lde[(size_t)(2 * k) * a.N + rn], nb = a.lde[(size_t)(2 * k + 1) * a.N + rn];
                const uint64_t t0 = gl::sub(na, gl::add(ca, cb));
                const uint64_t t1 = gl::sub(nb, gl::add(cb, na));
                const uint64_t b0 = gl::sub(ca, 1 + 2 * (uint64_t)k), b1 = gl::sub(cb, 2 + 2 * (uint64_t)k), b2 = gl::sub(cb, a.results[k]);
#pragma unroll
                for (int d = 0; d < F::DEG; d++) {
                    gl::wmac(wa[d], F::comp(a.ta[2 * k], d), t0);      gl::wmac(wb[d], F::comp(a.tb[2 * k], d), t0);
                    gl::wmac(wa[d], F::comp(a.ta[2 * k + 1], d), t1);  gl::wmac(wb[d], F::comp(a.tb[2 * k + 1], d), t1);
                    gl::wmac(w0a[d], F::comp(a.ba[2 * k], d), b0);     gl::wmac(w0b[d], F::comp(a.bb[2 * k], d), b0);
                    gl::wmac(w0a[d], F::comp(a.ba[2 * k + 1], d), b1); gl::wmac(w0b[d], F::comp(a.bb[2 * k + 1], d), b1);
                    gl::wmac(w1a[d], F::comp(a.ba[a.W + k], d), b2);   gl::wmac(w1b[d], F::comp(a.bb[a.W + k], d), b2);
                }
            }
            auto folded = [](const gl::Wide (&w)[F::DEG]) { return F::make(gl::wreduce(w[0]), F::DEG > 1 ? gl::wreduce(w[F::DEG - 1]) : 0); };
            acc = F::add(folded(wa), F::mulb(folded(wb), xt));
            g0 = F::add(folded(w0a), F::mulb(folded(w0b), xb));
            g1 = F::add(folded(w1a), F::mulb(folded(w1b), xb));
        } else {
            for (uint32_t k = 0; k < a.W / 2; k++) {
                const uint64_t ca = a.lde[(size_t)(2 * k) * a.N + r], cb = a.lde[(size_t)(2 * k + 1) * a.N + r];
                const uint64_t na = a.lde[(size_t)(2 * k) * a.N + rn], nb = a.lde[(size_t)(2 * k + 1) * a.N + rn];
                const uint64_t t0 = gl::sub(na, gl::add(ca, cb));
                const uint64_t t1 = gl::sub(nb, gl::add(cb, na));
                acc = F::add(acc, F::mulb(F::add(a.ta[2 * k], F::mulb(a.tb[2 * k], xt)), t0));
                acc = F::add(acc, F::mulb(F::add(a.ta[2 * k + 1], F::mulb(a.tb[2 * k + 1], xt)), t1));
                g0 = F::add(g0, F::mulb(F::add(a.ba[2 * k], F::mulb(a.bb[2 * k], xb)), gl::sub(ca, 1 + 2 * (uint64_t)k)));
                g0 = F::add(g0, F::mulb(F::add(a.ba[2 * k + 1], F::mulb(a.bb[2 * k + 1], xb)), gl::sub(cb, 2 + 2 * (uint64_t)k)));
                g1 = F::add(g1, F::mulb(F::add(a.ba[a.W + k], F::mulb(a.bb[a.W + k], xb)), gl::sub(cb, a.results[k])));
            }
        }
        if (a.A) {
            // auxiliary transition constraints (degree 2): adjustment x^n, constant on each of the C cosets of <w_n>
            uint64_t xx = a.xn[s & a.xmask];
            for (uint32_t e = 2; e < a.D; e++) xx = gl::mul(xx, x);
            for (uint32_t c = 0; c < a.A; c++) {
                const size_t o = (size_t)(c * F::DEG) * a.N;
                const T pc = F::make(a.aux[o + r], F::DEG > 1 ? a.aux[o + a.N + r] : 0);
                const T pn = F::make(a.aux[o + rn], F::DEG > 1 ? a.aux[o + a.N + rn] : 0);
                const uint64_t m = a.lde[(size_t)(c % a.W) * a.N + r];
                const T f1 = F::add(a.rands[c % a.R], F::from(m));
                T f = f1;
                for (uint32_t e = 2; e < a.D; e++) f = F::mul(f, f1);
                const T t = F::sub(pn, F::mul(pc, f));
                acc = F::add(acc, F::mul(F::add(a.ta[a.W + c], F::mulb(a.tb[a.W + c], xx)), t));
                const uint32_t bi = a.W + a.W / 2 + c;
                g0 = F::add(g0, F::mul(F::add(a.ba[bi], F::mulb(a.bb[bi], xb)), F::sub(pc, F::one())));
            }
        }
        num[q][0] = acc; num[q][1] = g0; num[q][2] = g1;
        if (MODE == 0) {
            const size_t o = s - a.first;
#pragma unroll
            for (int c = 0; c < 3; c++)
                for (int d = 0; d < F::DEG; d++) a.out_cols[(size_t)(c * F::DEG + d) * a.count + o] = F::comp(num[q][c], d);
        }
    }
    if (MODE == 1) {
        uint64_t den[2 * K];
        if (a.inv_tab) {
            // (x - 1)^-1, (x - w_n^(n-1))^-1 from the per-domain table: 16 bytes per row instead of 27 multiplications
#pragma unroll
            for (int q = 0; q < K; q++) {
                const size_t s = a.first + t + (size_t)q * nthreads;
                den[2 * q] = a.inv_tab[s];
                den[2 * q + 1] = a.inv_tab[a.inv_tab_n + s];
            }
        } else {
        // batch-invert (x - 1), (x - w_n^(n-1)) for the K points
        uint64_t pre[2 * K];
        uint64_t run = 1;
#pragma unroll
        for (int q = 0; q < K; q++) {
            den[2 * q] = gl::sub(xs[q], 1);
            den[2 * q + 1] = gl::sub(xs[q], a.w_last);
        }
#pragma unroll
        for (int i = 0; i < 2 * K; i++) { pre[i] = run; run = gl::mul(run, den[i]); }
        uint64_t ia = gl::inv(run);
#pragma unroll
        for (int i = 2 * K - 1; i >= 0; i--) { uint64_t v = gl::mul(ia, pre[i]); ia = gl::mul(ia, den[i]); den[i] = v; }
        }
#pragma unroll
        for (int q = 0; q < K; q++) {
            const size_t s = a.first + t + (size_t)q * nthreads;
            // 1 / ((x^n - 1) / (x - w^(n-1))) = (x - w^(n-1)) * zinv[s mod C]
            const uint64_t tdiv = a.inv_tab ? a.inv_tab[2 * a.inv_tab_n + s] : gl::mul(gl::sub(xs[q], a.w_last), a.zn_inv[s & a.xmask]);
            T h = F::mulb(num[q][0], tdiv);
            h = F::add(h, F::mulb(num[q][1], den[2 * q]));
            h = F::add(h, F::mulb(num[q][2], den[2 * q + 1]));
            for (int d = 0; d < F::DEG; d++) a.out_h[d][s] = F::comp(h, d);
        }
    }
}

// Everything the constraint kernel needs of a domain point x_s = offset * w_rows^s that does not depend on the proof, out[k * rows + s]:
//   k = 0: (x - 1)^-1   1: (x - w_last)^-1   2: (x - w_last) * zn_inv[s & xmask] (inverse of the transition divisor)
//   3: k7 * x^-1 (x^adj of the transition group)   4: k7 * x * xn_inv[s & xmask] (x^adj of the boundary groups)
// four points per thread share one inversion
struct FibTableArgs {
    uint64_t* out; size_t rows; uint64_t offset, gen_inv, k7, w_last;
    const uint64_t *tw_lo, *tw_hi, *twi_lo, *twi_hi; int tw_h;
    const uint64_t *xn_inv, *zn_inv; uint32_t xmask;
};
__global__ __launch_bounds__(256) void fib_inverse_table_kernel(FibTableArgs a) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = a.rows / 4;
    if (t >= nthreads) return;
    uint64_t den[8], pre[8], xs[4], run = 1;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint64_t x = gl::mul(a.offset, tw2(a.tw_lo, a.tw_hi, (uint32_t)(t + (size_t)q * nthreads), a.tw_h));
        xs[q] = x;
        den[2 * q] = gl::sub(x, 1);
        den[2 * q + 1] = gl::sub(x, a.w_last);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) { pre[i] = run; run = gl::mul(run, den[i]); }
    uint64_t ia = gl::inv(run);
#pragma unroll
    for (int i = 7; i >= 0; i--) { const uint64_t v = gl::mul(ia, pre[i]); ia = gl::mul(ia, den[i]); den[i] = v; }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const size_t s = t + (size_t)q * nthreads;
        a.out[s] = den[2 * q];
        a.out[a.rows + s] = den[2 * q + 1];
        a.out[2 * a.rows + s] = gl::mul(gl::sub(xs[q], a.w_last), a.zn_inv[s & a.xmask]);
        a.out[3 * a.rows + s] = gl::mul(a.k7, gl::mul(a.gen_inv, tw2(a.twi_lo, a.twi_hi, (uint32_t)s, a.tw_h)));
        a.out[4 * a.rows + s] = gl::mul(gl::mul(a.k7, xs[q]), a.xn_inv[s & a.xmask]);
    }
}
template <class F> void launch_fib_inverse_table(Context* ctx, uint64_t* out, const FibConsArgs<F>& c) {
    FibTableArgs a{out, c.count, c.offset, c.gen_inv, c.k7, c.w_last, c.tw_lo, c.tw_hi, c.twi_lo, c.twi_hi, c.tw_h, c.xn_inv, c.zn_inv, c.xmask};
    AERO_LAUNCH(ctx, "fib_inverse_table_kernel", 0, fib_inverse_table_kernel, dim3((unsigned)((a.rows / 4 + 255) / 256)), dim3(256), 0, a);
    ctx->check_launch("fib_inverse_table");
}
template void launch_fib_inverse_table<FB>(Context*, uint64_t*, const FibConsArgs<FB>&);
template void launch_fib_inverse_table<FQ>(Context*, uint64_t*, const FibConsArgs<FQ>&);

template <class F> void launch_fib_constraints(Context* ctx, const FibConsArgs<F>& a, int mode) {
    size_t cnt = a.count;
    const size_t in_cols = (size_t)a.W + (size_t)a.A * F::DEG;
    const bool wide = a.W >= 8;
    const dim3 g1((unsigned)((cnt + 255) / 256)), g4((unsigned)((cnt / 4 + 255) / 256));
    if (mode == 0) {
        if (wide) AERO_LAUNCH(ctx, "fib_constraints_kernel", cnt * 8 * (in_cols + 3 * F::DEG), (fib_constraints_kernel<F, 0, 1, true>), g1, dim3(256), 0, a);
        else AERO_LAUNCH(ctx, "fib_constraints_kernel", cnt * 8 * (in_cols + 3 * F::DEG), (fib_constraints_kernel<F, 0, 1, false>), g1, dim3(256), 0, a);
    } else if (cnt % 4 == 0) {   // 4 rows per thread share one batched inversion (measured best of 4 / 2 / 1 in both fields)
        // with the 160-bit sums 2 rows per thread: 6 sums x 5 limbs per row are live, 4 rows cost the occupancy (2^20 x 72: 0.81 ms with 4
        // rows, 0.56 with 2, 0.54 with 1; 2^20 x 8: 0.195 / 0.176 / 0.232)
        if (wide) AERO_LAUNCH(ctx, "fib_constraints_kernel", cnt * 8 * (in_cols + F::DEG), (fib_constraints_kernel<F, 1, 2, true>), dim3((unsigned)((cnt / 2 + 255) / 256)), dim3(256), 0, a);
        // with the divisor inverses from the per-shape table nothing is shared between a thread's rows: one row per thread (2^21 rows x 2 columns:
        // 56 / 50 / 46 us for 4 / 2 / 1 rows, 89 us with the per-thread inversions; profiles/r5_cons_inv_table.txt)
        else if (a.inv_tab) AERO_LAUNCH(ctx, "fib_constraints_kernel", cnt * 8 * (in_cols + F::DEG + 5), (fib_constraints_kernel<F, 1, 1, false>), g1, dim3(256), 0, a);
        else AERO_LAUNCH(ctx, "fib_constraints_kernel", cnt * 8 * (in_cols + F::DEG), (fib_constraints_kernel<F, 1, 4, false>), g4, dim3(256), 0, a);
    } else {
        if (wide) AERO_LAUNCH(ctx, "fib_constraints_kernel", cnt * 8 * (in_cols + F::DEG), (fib_constraints_kernel<F, 1, 1, true>), g1, dim3(256), 0, a);
        else AERO_LAUNCH(ctx, "fib_constraints_kernel", cnt * 8 * (in_cols + F::DEG), (fib_constraints_kernel<F, 1, 1, false>), g1, dim3(256), 0, a);
    }
    ctx->check_launch("fib_constraints");
}
template void launch_fib_constraints<FB>(Context*, const FibConsArgs<FB>&, int);
template void launch_fib_constraints<FQ>(Context*, const FibConsArgs<FQ>&, int);

// Division of stored numerator columns by their divisors (the unfused form of MODE 1 above; C-ABI stage entry point).
template <class F> __global__ __launch_bounds__(256) void fib_divide_kernel(FibDivideArgs<F> a) {
    typedef typename F::T T;
    const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.ce_n) return;
    const uint64_t x = gl::mul(a.offset, tw2(a.tw_lo, a.tw_hi, (uint32_t)s, a.tw_h));
    const uint64_t d0 = gl::sub(x, 1), d1 = gl::sub(x, a.w_last);
    const uint64_t inv01 = gl::inv(gl::mul(d0, d1));
    const uint64_t i0 = gl::mul(inv01, d1), i1 = gl::mul(inv01, d0);
    const uint64_t tdiv = gl::mul(d1, a.zn_inv[s & (a.C - 1)]);
    T num[3];
    for (int c = 0; c < 3; c++)
        num[c] = F::make(a.cols[(size_t)(c * F::DEG) * a.ce_n + s], F::DEG > 1 ? a.cols[(size_t)(c * F::DEG + 1) * a.ce_n + s] : 0);
    T h = F::mulb(num[0], tdiv);
    h = F::add(h, F::mulb(num[1], i0));
    h = F::add(h, F::mulb(num[2], i1));
    for (int d = 0; d < F::DEG; d++) a.out_h[d][s] = F::comp(h, d);
}
template <class F> void launch_fib_divide(Context* ctx, const FibDivideArgs<F>& a) {
    AERO_LAUNCH(ctx, "fib_divide_kernel", a.ce_n * 8 * 4 * F::DEG, (fib_divide_kernel<F>), dim3((unsigned)((a.ce_n + 255) / 256)), dim3(256), 0, a);
    ctx->check_launch("fib_divide");
}
template void launch_fib_divide<FB>(Context*, const FibDivideArgs<FB>&);
template void launch_fib_divide<FQ>(Context*, const FibDivideArgs<FQ>&);

// ------------------------------------------------------------------------------------------------
// Evaluate polynomials stored as bit-reversed coefficient vectors at up to 2 points:
//   out[col][pt] = sum_p coeff[col][p] * y_pt^(rev_L(p)).
// Position p = blk * 2^r + k  ->  rev_L(p) = rev_r(k) * 2^(L-r) + rev_(L-r)(blk), so
//   y^rev(p) = ktab_pt[k] * y^rev(blk),  ktab_pt[k] = (y^(2^(L-r)))^rev_r(k)   (table built by eval_ktab_kernel).
// grid = (blocks, columns); each workgroup reduces one block; a second tiny kernel adds the block partials.
template <class F> __global__ void eval_ktab_kernel(typename F::T* tab, int r, typename F::T y0, typename F::T y1, int npts) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (1u << r)) return;
    uint32_t e = gl::bitrev(k, r);
    tab[k] = gl::fpow<F>(y0, e);
    if (npts > 1) tab[(1u << r) + k] = gl::fpow<F>(y1, e);
}
template <class F> __global__ __launch_bounds__(256) void eval_bitrev_kernel(EvalArgs<F> a) {
    typedef typename F::T T;
    __shared__ T red[2][256];
    const uint32_t blk = blockIdx.x, col = blockIdx.y;
    const int R = 1 << a.r;
    // E-valued polynomial: second component lives comp_stride elements after the first
    const uint64_t* c0 = a.coeffs + (size_t)col * a.col_stride + ((size_t)blk << a.r);
    const uint64_t* c1 = a.comps > 1 ? c0 + a.comp_stride : c0;
    T acc0 = F::zero(), acc1 = F::zero();
    for (int k = threadIdx.x; k < R; k += 256) {
        if (a.comps == 1) {
            uint64_t c = c0[k];
            acc0 = F::add(acc0, F::mulb(a.ktab[k], c));
            if (a.npts > 1) acc1 = F::add(acc1, F::mulb(a.ktab[R + k], c));
        } else {
            T c = F::make(c0[k], c1[k]);
            acc0 = F::add(acc0, F::mul(a.ktab[k], c));
            if (a.npts > 1) acc1 = F::add(acc1, F::mul(a.ktab[R + k], c));
        }
    }
    red[0][threadIdx.x] = acc0; red[1][threadIdx.x] = acc1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] = F::add(red[0][threadIdx.x], red[0][threadIdx.x + s]);
            red[1][threadIdx.x] = F::add(red[1][threadIdx.x], red[1][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x < (unsigned)a.npts) {
        uint32_t rb = gl::bitrev(blk, a.L - a.r);
        T y = threadIdx.x == 0 ? a.y0 : a.y1;
        T v = F::mul(red[threadIdx.x][0], gl::fpow<F>(y, rb));
        a.partials[((size_t)col * a.npts + threadIdx.x) * gridDim.x + blk] = v;
    }
}
template <class F> __global__ __launch_bounds__(256) void eval_reduce_kernel(const typename F::T* partials, uint32_t nblk, typename F::T* out) {
    typedef typename F::T T;
    __shared__ T red[256];
    const typename F::T* p = partials + (size_t)blockIdx.x * nblk;
    T acc = F::zero();
    for (uint32_t i = threadIdx.x; i < nblk; i += 256) acc = F::add(acc, p[i]);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = F::add(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
// ---- the same for several jobs of one length at once (stark_kernels.hpp: EvalMultiArgs) ----
template <class F> __global__ void eval_ktab_multi_kernel(EvalMultiArgs<F> a) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x, job = blockIdx.y;
    if (k >= (1u << a.r)) return;
    const uint32_t e = gl::bitrev(k, a.r);
    typename F::T* tab = a.ktab + ((size_t)job * 2 << a.r);
    tab[k] = gl::fpow<F>(a.b0[job], e);
    if (a.jobs[job].npts > 1) tab[(1u << a.r) + k] = gl::fpow<F>(a.b1[job], e);
}
template <class F> __global__ __launch_bounds__(256) void eval_bitrev_multi_kernel(EvalMultiArgs<F> a) {
    typedef typename F::T T;
    __shared__ T red[2][256];
    const uint32_t blk = blockIdx.x;
    uint32_t col = blockIdx.y;
    int job = 0;
    while (job + 1 < a.n_jobs && col >= (uint32_t)a.jobs[job].ncols) { col -= (uint32_t)a.jobs[job].ncols; job++; }
    const EvalJob<F>& J = a.jobs[job];
    const int R = 1 << a.r;
    const T* ktab = a.ktab + ((size_t)job * 2 << a.r);
    const uint64_t* c0 = J.coeffs + (size_t)col * J.col_stride + ((size_t)blk << a.r);
    const uint64_t* c1 = J.comps > 1 ? c0 + J.comp_stride : c0;
    T acc0 = F::zero(), acc1 = F::zero();
    for (int k = threadIdx.x; k < R; k += 256) {
        if (J.comps == 1) {
            const uint64_t c = c0[k];
            acc0 = F::add(acc0, F::mulb(ktab[k], c));
            if (J.npts > 1) acc1 = F::add(acc1, F::mulb(ktab[R + k], c));
        } else {
            const T c = F::make(c0[k], c1[k]);
            acc0 = F::add(acc0, F::mul(ktab[k], c));
            if (J.npts > 1) acc1 = F::add(acc1, F::mul(ktab[R + k], c));
        }
    }
    red[0][threadIdx.x] = acc0; red[1][threadIdx.x] = acc1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] = F::add(red[0][threadIdx.x], red[0][threadIdx.x + s]);
            red[1][threadIdx.x] = F::add(red[1][threadIdx.x], red[1][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x < (unsigned)J.npts) {
        const uint32_t rb = gl::bitrev(blk, a.L - a.r);
        const T y = threadIdx.x == 0 ? J.y0 : J.y1;
        const T v = F::mul(red[threadIdx.x][0], gl::fpow<F>(y, rb));
        a.partials[((size_t)J.out_off + (size_t)col * J.npts + threadIdx.x) * gridDim.x + blk] = v;
    }
}
template <class F> void launch_eval_multi(Context* ctx, const EvalJob<F>* jobs, int n_jobs, int L, typename F::T* out) {
    typedef typename F::T T;
    if (n_jobs < 1 || n_jobs > EVAL_MAX_JOBS) fail("eval_multi: bad job count", ST_INTERNAL);
    EvalMultiArgs<F> a;
    a.n_jobs = n_jobs; a.L = L; a.r = L < 10 ? L : 10;
    const uint32_t nblk = 1u << (L - a.r);
    uint32_t cols = 0, outs = 0;
    size_t abytes = 0;
    for (int j = 0; j < n_jobs; j++) {
        a.jobs[j] = jobs[j];
        a.b0[j] = gl::fpow<F>(jobs[j].y0, 1ull << (L - a.r));
        a.b1[j] = jobs[j].npts > 1 ? gl::fpow<F>(jobs[j].y1, 1ull << (L - a.r)) : F::zero();
        cols += (uint32_t)jobs[j].ncols; outs += (uint32_t)(jobs[j].ncols * jobs[j].npts);
        abytes += ((size_t)jobs[j].ncols * jobs[j].comps * 8) << L;
    }
    for (int j = n_jobs; j < EVAL_MAX_JOBS; j++) { a.jobs[j] = EvalJob<F>{}; a.jobs[j].y0 = a.jobs[j].y1 = F::zero(); a.b0[j] = a.b1[j] = F::zero(); }
    a.ktab = (T*)ctx->scratch_alloc(sizeof(T) * 2 * (size_t)n_jobs * ((size_t)1 << a.r));
    a.partials = (T*)ctx->scratch_alloc(sizeof(T) * (size_t)outs * nblk);
    AERO_LAUNCH(ctx, "eval_ktab_kernel", 0, (eval_ktab_multi_kernel<F>), dim3(((1u << a.r) + 255) / 256, n_jobs), dim3(256), 0, a);
    AERO_LAUNCH(ctx, "eval_bitrev_kernel", abytes, (eval_bitrev_multi_kernel<F>), dim3(nblk, cols), dim3(256), 0, a);
    AERO_LAUNCH(ctx, "eval_reduce_kernel", 0, (eval_reduce_kernel<F>), dim3(outs), dim3(256), 0, a.partials, nblk, out);
    ctx->check_launch("eval_multi");
}
template void launch_eval_multi<FB>(Context*, const EvalJob<FB>*, int, int, FB::T*);
template void launch_eval_multi<FQ>(Context*, const EvalJob<FQ>*, int, int, FQ::T*);

// out (device, ncols*npts values, index col*npts + pt)
template <class F>
void launch_eval_bitrev(Context* ctx, const uint64_t* coeffs, size_t col_stride, size_t comp_stride, int ncols, int comps, int L,
                        typename F::T y0, typename F::T y1, int npts, typename F::T* out) {
    typedef typename F::T T;
    int r = L < 10 ? L : 10;
    uint32_t nblk = 1u << (L - r);
    T* ktab = (T*)ctx->scratch_alloc(sizeof(T) * 2 * ((size_t)1 << r));
    T* partials = (T*)ctx->scratch_alloc(sizeof(T) * (size_t)ncols * npts * nblk);
    T b0 = gl::fpow<F>(y0, 1ull << (L - r)), b1 = gl::fpow<F>(y1, 1ull << (L - r));
    AERO_LAUNCH(ctx, "eval_ktab_kernel", 0, (eval_ktab_kernel<F>), dim3(((1u << r) + 255) / 256), dim3(256), 0, ktab, r, b0, b1, npts);
    EvalArgs<F> a;
    a.coeffs = coeffs; a.col_stride = col_stride; a.comp_stride = comp_stride; a.comps = comps; a.L = L; a.r = r; a.npts = npts; a.y0 = y0; a.y1 = y1;
    a.ktab = ktab; a.partials = partials;
    AERO_LAUNCH(ctx, "eval_bitrev_kernel", ((size_t)ncols * comps * 8) << L, (eval_bitrev_kernel<F>), dim3(nblk, ncols), dim3(256), 0, a);
    AERO_LAUNCH(ctx, "eval_reduce_kernel", 0, (eval_reduce_kernel<F>), dim3(ncols * npts), dim3(256), 0, partials, nblk, out);
    ctx->check_launch("eval_bitrev");
}
template void launch_eval_bitrev<FB>(Context*, const uint64_t*, size_t, size_t, int, int, int, uint64_t, uint64_t, int, uint64_t*);
template void launch_eval_bitrev<FQ>(Context*, const uint64_t*, size_t, size_t, int, int, int, gl::E2, gl::E2, int, gl::E2*);

// ------------------------------------------------------------------------------------------------
// DEEP composition (composer.cairo:48-316 is the verifier-side mirror), evaluated on every row_step-th LDE row.
// The DEEP polynomial has degree < n, so the prover evaluates it on the n-point coset 7<w_n> only (row_step = blowup: 1/8
// of the field inversions) and extends it to the LDE domain with one more column NTT:
//   deep(x) = [ sum_i a_i (T_i(x) - T_i(z)) / (x - z) + sum_i b_i (T_i(x) - T_i(z g)) / (x - z g)
//             (+ sum_i c_i (T_i(x) - conj T_i(z)) / (x - conj z)  when E = F_p^2)
//             + sum_c d_c (H_c(x) - H_c(z^C)) / (x - z^C) ] * (lambda + mu x)
template <class F, int K> __global__ __launch_bounds__(256) void deep_kernel(DeepArgs<F> a) {
    typedef typename F::T T;
    constexpr int ND = F::DEG > 1 ? 4 : 3;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t nthreads = a.count / K;
    if (t >= nthreads) return;
    T den[ND * K], pre[ND * K];
    uint64_t xs[K];
#pragma unroll
    for (int q = 0; q < K; q++) {
        const size_t r = (t + (size_t)q * nthreads) * a.row_step;
        const uint64_t x = gl::mul(a.offset, tw2(a.tw_lo, a.tw_hi, (uint32_t)r, a.tw_h));
        xs[q] = x;
        const T xe = F::from(x);
        den[ND * q] = F::sub(xe, a.z); den[ND * q + 1] = F::sub(xe, a.z_next); den[ND * q + 2] = F::sub(xe, a.z_c);
        if (ND == 4) den[ND * q + 3] = F::sub(xe, a.z_conj);
    }
    T run = F::one();
#pragma unroll
    for (int i = 0; i < ND * K; i++) { pre[i] = run; run = F::mul(run, den[i]); }
    T ia = F::inv(run);
#pragma unroll
    for (int i = ND * K - 1; i >= 0; i--) { T v = F::mul(ia, pre[i]); ia = F::mul(ia, den[i]); den[i] = v; }
#pragma unroll
    for (int q = 0; q < K; q++) {
        const size_t m = t + (size_t)q * nthreads;
        const size_t rt = m * a.t_step, rc = m * a.c_step, ra = m * a.a_step;
        T s1 = F::zero(), s2 = F::zero(), s3 = F::zero();
        for (uint32_t c = 0; c < a.W; c++) {
            const T v = F::from(a.tlde[(size_t)c * a.t_stride + rt]);
            s1 = F::add(s1, F::mul(F::sub(v, a.ood_cur[c]), a.da[c]));
            s2 = F::add(s2, F::mul(F::sub(v, a.ood_next[c]), a.db[c]));
            if (F::DEG > 1) s3 = F::add(s3, F::mul(F::sub(v, F::conj(a.ood_cur[c])), a.dg[c]));
        }
        for (uint32_t c = 0; c < a.A; c++) {   // aux columns are E-valued: no conjugate term
            const size_t o = (size_t)(c * F::DEG) * a.a_stride;
            const T v = F::make(a.alde[o + ra], F::DEG > 1 ? a.alde[o + a.a_stride + ra] : 0);
            s1 = F::add(s1, F::mul(F::sub(v, a.ood_cur[a.W + c]), a.da[a.W + c]));
            s2 = F::add(s2, F::mul(F::sub(v, a.ood_next[a.W + c]), a.db[a.W + c]));
        }
        T acc = F::add(F::mul(s1, den[ND * q]), F::mul(s2, den[ND * q + 1]));
        if (F::DEG > 1) acc = F::add(acc, F::mul(s3, den[ND * q + 3]));
        T sc = F::zero();
        for (uint32_t c = 0; c < a.C; c++) {
            const T v = F::make(a.clde[(size_t)(c * F::DEG) * a.c_stride + rc], F::DEG > 1 ? a.clde[(size_t)(c * F::DEG + F::DEG - 1) * a.c_stride + rc] : 0);
            sc = F::add(sc, F::mul(F::sub(v, a.ood_h[c]), a.dc[c]));
        }
        acc = F::add(acc, F::mul(sc, den[ND * q + 2]));
        acc = F::mul(acc, F::add(a.lambda, F::mulb(a.mu, xs[q])));
        for (int d = 0; d < F::DEG; d++) a.out[d][m] = F::comp(acc, d);
    }
}
template <class F> void launch_deep(Context* ctx, const DeepArgs<F>& a) {
    const size_t bytes = a.count * 8 * ((size_t)a.W + ((size_t)a.C + a.A) * F::DEG + F::DEG);
    // rows per thread share one batched inversion; over F_p^2 four rows need 335 VGPRs + scratch (1 wave per SIMD), two fit
    // (measured, 2^20 rows: F_p 78 / 65 / 90 us for 4 / 2 / 1 rows per thread, F_p^2 262 / 176 / 152 us)
    const int K = F::DEG > 1 ? 1 : 2;
    if (K == 4 && a.count % 4 == 0 && a.count >= 4096)
        AERO_LAUNCH(ctx, "deep_kernel", bytes, (deep_kernel<F, 4>), dim3((unsigned)((a.count / 4 + 255) / 256)), dim3(256), 0, a);
    else if (K == 2 && a.count % 2 == 0 && a.count >= 4096)
        AERO_LAUNCH(ctx, "deep_kernel", bytes, (deep_kernel<F, 2>), dim3((unsigned)((a.count / 2 + 255) / 256)), dim3(256), 0, a);
    else
        AERO_LAUNCH(ctx, "deep_kernel", bytes, (deep_kernel<F, 1>), dim3((unsigned)((a.count + 255) / 256)), dim3(256), 0, a);
    ctx->check_launch("deep");
}
template void launch_deep<FB>(Context*, const DeepArgs<FB>&);
template void launch_deep<FQ>(Context*, const DeepArgs<FQ>&);

// ------------------------------------------------------------------------------------------------
// DEEP composition in coefficient form (DeepCoeffArgs). Synthetic division q_(m-1) = c_m + y q_m is a suffix scan over the coefficient index.
// Coefficient k = a B + b of a bit-reversed vector sits at position rev(b) (n / B) + rev(a): a thread that walks the B coefficients of block a
// serially reads, together with its neighbours (consecutive rev(a)), contiguous segments - and the vector of block values, indexed by rev(a), is
// again a bit-reversed vector, so the scheme nests. Every serial walk fetches its values eight steps ahead (a lone dependent chain per thread would
// otherwise pay one memory latency per step: 133 us instead of 25 for the first kernel). Launches for 2^20 coefficients (blocks of 8, then 8,
// then one workgroup per quotient over 2^14 values):
//   deep_coeff_combine_kernel   the three combined polynomials P_j (stored: the way down reads them again) and their block values E1;
//   deep_coeff_up_kernel        block values E2 of E1 at the points y^8;
//   deep_coeff_mid_kernel       E2 -> carries: serial per thread, log-step scan over the threads through LDS, serial again; one workgroup per
//                               quotient (all three on one CU took 71 us, on three 31);
//   deep_coeff_down_kernel      carries of the level below from the carries of this one (in place over the block values);
//   deep_coeff_down_kernel<FINAL> the quotients from the carries down, their sum times (lam + mu y) on the way out.
constexpr int DC_CHUNK = 8;
struct DeepLevel {            // one level of the scan over three chains stored as [3][1 << log_len] (bit-reversed order)
    uint64_t* vals;           // values of this level (up: read; down: read, then overwritten by the carries unless FINAL)
    uint64_t* blocks;         // [3][len >> log_b]: block values (up: written), carries (down: read)
    int log_len, log_b;
    uint64_t y[3];            // the level's point per chain
};
__device__ __forceinline__ size_t dc_pos(int b, int log_b, int log_len, size_t ra) { return ((size_t)gl::bitrev((uint32_t)b, log_b) << (log_len - log_b)) + ra; }

__global__ __launch_bounds__(256) void deep_coeff_combine_kernel(DeepCoeffArgs a, DeepLevel lv) {
    const size_t nb = (size_t)1 << (lv.log_len - lv.log_b), ra = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)1 << lv.log_len;
    if (ra >= nb) return;
    uint64_t acc[3] = {0, 0, 0};
    const int B = 1 << lv.log_b, step = B < DC_CHUNK ? B : DC_CHUNK;
    for (int b0 = B - step; b0 >= 0; b0 -= step) {
        uint64_t p[DC_CHUNK][3];
        size_t pos[DC_CHUNK];
#pragma unroll
        for (int i = 0; i < DC_CHUNK; i++) { pos[i] = i < step ? dc_pos(b0 + i, lv.log_b, lv.log_len, ra) : 0; p[i][0] = p[i][1] = p[i][2] = 0; }
        for (uint32_t c = 0; c < a.W + a.A; c++) {
            const uint64_t ka = a.da[c], kb = a.db[c];
            const uint64_t* col = c < a.W ? a.tpolys + (size_t)c * a.t_stride : a.apolys + (size_t)(c - a.W) * a.a_stride;
#pragma unroll
            for (int i = 0; i < DC_CHUNK; i++) if (i < step) { const uint64_t v = col[pos[i]]; p[i][0] = gl::add(p[i][0], gl::mul(ka, v)); p[i][1] = gl::add(p[i][1], gl::mul(kb, v)); }
        }
        for (uint32_t c = 0; c < a.C; c++) {
            const uint64_t kc = a.dc[c];
#pragma unroll
            for (int i = 0; i < DC_CHUNK; i++) if (i < step) p[i][2] = gl::add(p[i][2], gl::mul(kc, a.hpolys[(size_t)c * a.h_stride + pos[i]]));
        }
#pragma unroll
        for (int i = DC_CHUNK - 1; i >= 0; i--) if (i < step) {
#pragma unroll
            for (int j = 0; j < 3; j++) { lv.vals[(size_t)j * n + pos[i]] = p[i][j]; acc[j] = gl::add(p[i][j], gl::mul(lv.y[j], acc[j])); }
        }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) lv.blocks[(size_t)j * nb + ra] = acc[j];
}
__global__ __launch_bounds__(256) void deep_coeff_up_kernel(DeepLevel lv) {
    const size_t nb = (size_t)1 << (lv.log_len - lv.log_b), ra = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)1 << lv.log_len;
    if (ra >= nb) return;
    uint64_t acc[3] = {0, 0, 0};
    const int B = 1 << lv.log_b, step = B < DC_CHUNK ? B : DC_CHUNK;
    for (int b0 = B - step; b0 >= 0; b0 -= step) {
        uint64_t p[DC_CHUNK][3];
#pragma unroll
        for (int i = 0; i < DC_CHUNK; i++) if (i < step) {
            const size_t pos = dc_pos(b0 + i, lv.log_b, lv.log_len, ra);
#pragma unroll
            for (int j = 0; j < 3; j++) p[i][j] = lv.vals[(size_t)j * n + pos];
        }
#pragma unroll
        for (int i = DC_CHUNK - 1; i >= 0; i--) if (i < step) {
#pragma unroll
            for (int j = 0; j < 3; j++) acc[j] = gl::add(p[i][j], gl::mul(lv.y[j], acc[j]));
        }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) lv.blocks[(size_t)j * nb + ra] = acc[j];
}
// carries of this level's values from the carries of its blocks. FINAL (level of the polynomials themselves): instead of the carries, the DEEP
// coefficients lam S_m + mu S_(m-1), S = sum of the three quotients, go to `out`
template <bool FINAL> __global__ __launch_bounds__(256) void deep_coeff_down_kernel(DeepLevel lv, uint64_t lam, uint64_t mu, uint64_t* __restrict__ out) {
    const size_t nb = (size_t)1 << (lv.log_len - lv.log_b), ra = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)1 << lv.log_len;
    if (ra >= nb) return;
    uint64_t acc[3];
#pragma unroll
    for (int j = 0; j < 3; j++) acc[j] = lv.blocks[(size_t)j * nb + ra];
    uint64_t s_hi = FINAL ? gl::add(gl::add(acc[0], acc[1]), acc[2]) : 0;      // S_m at the top coefficient m of the block
    const int B = 1 << lv.log_b, step = B < DC_CHUNK ? B : DC_CHUNK;
    for (int b0 = B - step; b0 >= 0; b0 -= step) {
        uint64_t p[DC_CHUNK][3];
        size_t pos[DC_CHUNK];
#pragma unroll
        for (int i = 0; i < DC_CHUNK; i++) if (i < step) {
            pos[i] = dc_pos(b0 + i, lv.log_b, lv.log_len, ra);
#pragma unroll
            for (int j = 0; j < 3; j++) p[i][j] = lv.vals[(size_t)j * n + pos[i]];
        }
#pragma unroll
        for (int i = DC_CHUNK - 1; i >= 0; i--) if (i < step) {
            if (!FINAL) {
#pragma unroll
                for (int j = 0; j < 3; j++) { lv.vals[(size_t)j * n + pos[i]] = acc[j]; acc[j] = gl::add(p[i][j], gl::mul(lv.y[j], acc[j])); }
            } else {
#pragma unroll
                for (int j = 0; j < 3; j++) acc[j] = gl::add(p[i][j], gl::mul(lv.y[j], acc[j]));
                uint64_t s_lo = gl::add(gl::add(acc[0], acc[1]), acc[2]);   // S_(m-1); below coefficient 0 there is the remainder, not a quotient term
                if (b0 + i == 0 && ra == 0) s_lo = 0;
                out[pos[i]] = gl::add(gl::mul(lam, s_hi), gl::mul(mu, s_lo));
                s_hi = s_lo;
            }
        }
    }
}
// vals[j][rev(a)] = E_a  ->  carry_a = sum_(a' > a) E_a' y^(a' - a - 1), in place; workgroup j (one CU) scans chain j. Thread t owns the
// 2^log_per consecutive a of block rev(t) - their positions are rev(i) * threads + t, contiguous over the threads like everywhere else - and
// the scan over the blocks runs through LDS in block order with the host's step multipliers w[j][s] = (y_j^per)^(2^s).
struct DeepMidW { uint64_t w[3][10]; };
__global__ __launch_bounds__(1024) void deep_coeff_mid_kernel(DeepLevel lv, int log_threads, DeepMidW mw) {
    __shared__ uint64_t sh[1024];
    const int j = blockIdx.x;
    const int log_per = lv.log_len - log_threads;
    uint64_t* vals = lv.vals + ((size_t)j << lv.log_len);
    const int per = 1 << log_per, tid = threadIdx.x, threads = 1 << log_threads, step = per < DC_CHUNK ? per : DC_CHUNK;
    const int blk = (int)gl::bitrev((uint32_t)tid, log_threads);
    const uint64_t y = lv.y[j];
    uint64_t acc = 0;
    for (int i0 = per - step; i0 >= 0; i0 -= step) {
        uint64_t p[DC_CHUNK];
#pragma unroll
        for (int i = 0; i < DC_CHUNK; i++) if (i < step) p[i] = vals[dc_pos(i0 + i, log_per, lv.log_len, (size_t)tid)];
#pragma unroll
        for (int i = DC_CHUNK - 1; i >= 0; i--) if (i < step) acc = gl::add(p[i], gl::mul(y, acc));
    }
    sh[blk] = acc;
    __syncthreads();
    // inclusive suffix scan over the blocks: S_t = sum_(t' >= t) v_t' w^(t' - t)
    for (int d = 1, st = 0; d < threads; d <<= 1, st++) {
        const uint64_t add = blk + d < threads ? gl::mul(mw.w[j][st], sh[blk + d]) : 0;
        __syncthreads();
        sh[blk] = gl::add(sh[blk], add);
        __syncthreads();
    }
    acc = blk + 1 < threads ? sh[blk + 1] : 0;       // carry into this block's top value
    for (int i0 = per - step; i0 >= 0; i0 -= step) {
        uint64_t p[DC_CHUNK];
        size_t pos[DC_CHUNK];
#pragma unroll
        for (int i = 0; i < DC_CHUNK; i++) if (i < step) { pos[i] = dc_pos(i0 + i, log_per, lv.log_len, (size_t)tid); p[i] = vals[pos[i]]; }
#pragma unroll
        for (int i = DC_CHUNK - 1; i >= 0; i--) if (i < step) { vals[pos[i]] = acc; acc = gl::add(p[i], gl::mul(y, acc)); }
    }
}
void launch_deep_coeff(Context* ctx, const DeepCoeffArgs& a) {
    // levels: the polynomials (2^log_n, blocks of 2^b1), then - when more than 2^14 blocks remain - their block values (blocks of 2^b2), then one
    // workgroup per quotient (2^14 values: 2^12 measured the same, 2^10 slower, profiles/r5_deep_coeff.md)
    const int L = a.log_n;
    const int b1 = L >= 6 ? 3 : (L + 1) / 2;
    const int L1 = L - b1;
    const int b2 = L1 > 14 ? L1 - 14 : 0;
    const int L2 = L1 - b2;
    const size_t n = (size_t)1 << L;
    // scratch layout (a.blocks): [3][n] combined polynomials | [3][2^L1] | [3][2^L2]
    uint64_t* comb = a.blocks;
    uint64_t* e1 = comb + 3 * n;
    uint64_t* e2 = e1 + ((size_t)3 << L1);
    DeepLevel lv1{comb, e1, L, b1, {a.y[0], a.y[1], a.y[2]}};
    DeepLevel lv2{e1, e2, L1, b2, {0, 0, 0}}, lvm{b2 ? e2 : e1, nullptr, L2, 0, {0, 0, 0}};
    for (int j = 0; j < 3; j++) { lv2.y[j] = gl::pow(a.y[j], 1ull << b1); lvm.y[j] = gl::pow(lv2.y[j], 1ull << b2); }
    const size_t bytes = ((size_t)a.W + a.A + a.C + 3) * 8 << L;
    const dim3 g1((unsigned)((((size_t)1 << L1) + 255) / 256)), g2((unsigned)((((size_t)1 << L2) + 255) / 256));
    AERO_LAUNCH(ctx, "deep_coeff_combine_kernel", bytes, deep_coeff_combine_kernel, g1, dim3(256), 0, a, lv1);
    if (b2) AERO_LAUNCH(ctx, "deep_coeff_up_kernel", 0, deep_coeff_up_kernel, g2, dim3(256), 0, lv2);
    const int log_threads = L2 < 10 ? L2 : 10;
    DeepMidW mw;
    for (int j = 0; j < 3; j++) {
        mw.w[j][0] = gl::pow(lvm.y[j], 1ull << (L2 - log_threads));
        for (int st = 1; st < 10; st++) mw.w[j][st] = gl::mul(mw.w[j][st - 1], mw.w[j][st - 1]);
    }
    AERO_LAUNCH(ctx, "deep_coeff_mid_kernel", 0, deep_coeff_mid_kernel, dim3(3), dim3(1u << log_threads), 0, lvm, log_threads, mw);
    if (b2) AERO_LAUNCH(ctx, "deep_coeff_down_kernel", 0, deep_coeff_down_kernel<false>, g2, dim3(256), 0, lv2, (uint64_t)0, (uint64_t)0, (uint64_t*)nullptr);
    AERO_LAUNCH(ctx, "deep_coeff_quotient_kernel", (size_t)32 << L, deep_coeff_down_kernel<true>, g1, dim3(256), 0, lv1, a.lam, a.mu, a.out);
    ctx->check_launch("deep_coeff");
}

// ------------------------------------------------------------------------------------------------
// FRI fold by `fold` (fri_verifier.cairo:305-315 mirror; winter-fri apply_drp): row i = (v[i + j*rows])_j lies on
// x_i * <w_F>, x_i = 7 * w_dom^i (the offset stays 7 at every layer: fri_verifier.cairo:23,308). With c_k = inverse
// DFT of the row, the interpolant evaluated at alpha is (1/F) * sum_k (alpha / x_i)^k * sum_j v_j w_F^(-jk).
template <class F> __global__ __launch_bounds__(256) void fri_fold_kernel(FoldArgs<F> a) {
    typedef typename F::T T;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.rows) return;
    T v[16];
    for (int j = 0; j < a.fold; j++) v[j] = F::make(a.in[0][i + (size_t)j * a.rows], F::DEG > 1 ? a.in[1][i + (size_t)j * a.rows] : 0);
    const uint64_t xinv = gl::mul(a.gen_inv, tw2(a.twi_lo, a.twi_hi, (uint32_t)i, a.tw_h));
    const T r = F::mulb(a.alpha_dev ? *a.alpha_dev : a.alpha, xinv);
    T rp = F::one(), acc = F::zero();
    for (int k = 0; k < a.fold; k++) {
        T ck = F::zero();
        for (int j = 0; j < a.fold; j++) ck = F::add(ck, F::mulb(v[j], a.dft[(j * k) & (a.fold - 1)]));
        acc = F::add(acc, F::mul(ck, rp));
        rp = F::mul(rp, r);
    }
    acc = F::mulb(acc, a.fold_inv);
    for (int d = 0; d < F::DEG; d++) a.out[d][i] = F::comp(acc, d);
}
// Same fold with the inverse DFT done as radix-2 butterflies in registers (fold = 2, 4, 8): the inverse DFT leaves
// F * c_k at position bitrev(k); constant twiddles are powers of two (dft_small.hpp).
template <class F, int LOGF> __global__ __launch_bounds__(256) void fri_fold_fft_kernel(FoldArgs<F> a) {
    typedef typename F::T T;
    constexpr int FD = 1 << LOGF;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.rows) return;
    uint64_t y0[FD], y1[FD];
#pragma unroll
    for (int j = 0; j < FD; j++) {
        y0[j] = a.in[0][i + (size_t)j * a.rows];
        if (F::DEG > 1) y1[j] = a.in[1][i + (size_t)j * a.rows];
    }
    dft_dif_inv<LOGF>(y0);
    if (F::DEG > 1) dft_dif_inv<LOGF>(y1);
    const uint64_t xinv = gl::mul(a.gen_inv, tw2(a.twi_lo, a.twi_hi, (uint32_t)i, a.tw_h));
    const T r = F::mulb(a.alpha_dev ? *a.alpha_dev : a.alpha, xinv);
    T acc = F::zero();
#pragma unroll
    for (int k = FD - 1; k >= 0; k--) {
        const int p = (int)gl::bitrev((uint32_t)k, LOGF);
        acc = F::add(F::mul(acc, r), F::make(y0[p], F::DEG > 1 ? y1[p] : 0));
    }
    acc = F::mulb(acc, a.fold_inv);
    for (int d = 0; d < F::DEG; d++) a.out[d][i] = F::comp(acc, d);
}
template <class F> void launch_fri_fold(Context* ctx, const FoldArgs<F>& a) {
    const dim3 grid((unsigned)((a.rows + 255) / 256)), block(256);
    const size_t bytes = a.rows * 8 * F::DEG * ((size_t)a.fold + 1);
    if (a.fold == 8) AERO_LAUNCH(ctx, "fri_fold_kernel", bytes, (fri_fold_fft_kernel<F, 3>), grid, block, 0, a);
    else if (a.fold == 4) AERO_LAUNCH(ctx, "fri_fold_kernel", bytes, (fri_fold_fft_kernel<F, 2>), grid, block, 0, a);
    else if (a.fold == 2) AERO_LAUNCH(ctx, "fri_fold_kernel", bytes, (fri_fold_fft_kernel<F, 1>), grid, block, 0, a);
    else AERO_LAUNCH(ctx, "fri_fold_kernel", bytes, (fri_fold_kernel<F>), grid, block, 0, a);
    ctx->check_launch("fri_fold");
}
template <class F> __global__ void fri_coin_step_kernel(Digest* seed_io, const Digest* root, typename F::T* alpha_out) {
    if (threadIdx.x || blockIdx.x) return;
    const Digest seed = b2s::merge(*seed_io, *root);
    *seed_io = seed;
    for (uint64_t ctr = 1; ctr < 1000; ctr++) {
        const Digest d = b2s::merge_with_int(seed, ctr);
        const uint64_t v0 = (uint64_t)d.w[0] | ((uint64_t)d.w[1] << 32), v1 = (uint64_t)d.w[2] | ((uint64_t)d.w[3] << 32);
        if (v0 < gl::P && (F::DEG == 1 || v1 < gl::P)) { *alpha_out = F::make(v0, v1); return; }
    }
    *alpha_out = F::zero();   // unreachable in practice (the host replay of the transcript would fail the same way)
}
template <class F> void launch_fri_coin_step(Context* ctx, Digest* seed_io, const Digest* root, typename F::T* alpha_out) {
    AERO_LAUNCH(ctx, "fri_coin_step_kernel", 0, (fri_coin_step_kernel<F>), dim3(1), dim3(64), 0, seed_io, root, alpha_out);
    ctx->check_launch("fri_coin_step");
}
template void launch_fri_coin_step<FB>(Context*, Digest*, const Digest*, uint64_t*);
template void launch_fri_coin_step<FQ>(Context*, Digest*, const Digest*, gl::E2*);
template void launch_fri_fold<FB>(Context*, const FoldArgs<FB>&);
template void launch_fri_fold<FQ>(Context*, const FoldArgs<FQ>&);

// ------------------------------------------------------------------------------------------------
// Auxiliary segment: prefix products over the rows. Three launches: per-block totals, exclusive scan of the block totals,
// per-row values. A thread owns AUX_K consecutive rows, a workgroup AUX_K * 256; the factor of row i is
// rands[c mod R] + trace[c mod W][i] (rows >= n contribute 1).
constexpr int AUX_K = 8;
template <class F> __device__ __forceinline__ typename F::T aux_factor(const uint64_t* __restrict__ col, typename F::T r, size_t i, size_t n, uint32_t D) {
    if (i >= n) return F::one();
    const typename F::T f1 = F::add(r, F::from(col[i]));
    typename F::T f = f1;
    for (uint32_t e = 2; e < D; e++) f = F::mul(f, f1);
    return f;
}
// inclusive scan of one value per thread over the workgroup (Hillis-Steele through LDS); returns this thread's inclusive value
template <class F> __device__ __forceinline__ typename F::T block_scan_mul(typename F::T v, typename F::T* sh) {
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
template <class F> __global__ __launch_bounds__(256) void aux_block_totals_kernel(const uint64_t* trace, size_t n, uint32_t W, uint32_t R, uint32_t D,
                                                                               const typename F::T* rands, typename F::T* totals) {
    typedef typename F::T T;
    __shared__ T sh[256];
    const uint32_t c = blockIdx.y;
    const uint64_t* col = trace + (size_t)(c % W) * n;
    const T r = rands[c % R];
    const size_t first = ((size_t)blockIdx.x * 256 + threadIdx.x) * AUX_K;
    T p = F::one();
#pragma unroll
    for (int k = 0; k < AUX_K; k++) p = F::mul(p, aux_factor<F>(col, r, first + k, n, D));
    const T inc = block_scan_mul<F>(p, sh);
    if (threadIdx.x == 255) totals[(size_t)c * gridDim.x + blockIdx.x] = inc;
}
// in place: totals[c][b] <- product of totals[c][0..b) (one workgroup per column)
template <class F> __global__ __launch_bounds__(256) void aux_scan_totals_kernel(typename F::T* totals, uint32_t nblk) {
    typedef typename F::T T;
    __shared__ T sh[256];
    T* row = totals + (size_t)blockIdx.x * nblk;
    const uint32_t per = (nblk + 255) / 256, lo = threadIdx.x * per;
    T p = F::one();
    for (uint32_t i = lo; i < lo + per && i < nblk; i++) p = F::mul(p, row[i]);
    const T inc = block_scan_mul<F>(p, sh);
    T run = threadIdx.x == 0 ? F::one() : sh[threadIdx.x - 1];
    (void)inc;
    for (uint32_t i = lo; i < lo + per && i < nblk; i++) { const T v = row[i]; row[i] = run; run = F::mul(run, v); }
}
template <class F> __global__ __launch_bounds__(256) void aux_apply_kernel(const uint64_t* trace, size_t n, uint32_t W, uint32_t R, uint32_t D,
                                                                        const typename F::T* rands, const typename F::T* totals, uint64_t* out) {
    typedef typename F::T T;
    __shared__ T sh[256];
    const uint32_t c = blockIdx.y;
    const uint64_t* col = trace + (size_t)(c % W) * n;
    const T r = rands[c % R];
    const size_t first = ((size_t)blockIdx.x * 256 + threadIdx.x) * AUX_K;
    T f[AUX_K];
    T p = F::one();
#pragma unroll
    for (int k = 0; k < AUX_K; k++) { f[k] = aux_factor<F>(col, r, first + k, n, D); p = F::mul(p, f[k]); }
    block_scan_mul<F>(p, sh);
    T run = totals[(size_t)c * gridDim.x + blockIdx.x];
    if (threadIdx.x) run = F::mul(run, sh[threadIdx.x - 1]);
#pragma unroll
    for (int k = 0; k < AUX_K; k++) {
        if (first + k < n)
            for (int d = 0; d < F::DEG; d++) out[((size_t)c * F::DEG + d) * n + first + k] = F::comp(run, d);
        run = F::mul(run, f[k]);
    }
}
template <class F> void launch_aux_columns(Context* ctx, const uint64_t* trace, size_t n, uint32_t W, uint32_t A, uint32_t R, uint32_t D,
                                           const typename F::T* rands_dev, uint64_t* out) {
    typedef typename F::T T;
    const uint32_t nblk = (uint32_t)((n + (size_t)AUX_K * 256 - 1) / ((size_t)AUX_K * 256));
    T* totals = (T*)ctx->scratch_alloc(sizeof(T) * (size_t)A * nblk);
    AERO_LAUNCH(ctx, "aux_columns_kernel", (size_t)A * n * 8, (aux_block_totals_kernel<F>), dim3(nblk, A), dim3(256), 0, trace, n, W, R, D, rands_dev, totals);
    AERO_LAUNCH(ctx, "aux_columns_kernel", 0, (aux_scan_totals_kernel<F>), dim3(A), dim3(256), 0, totals, nblk);
    AERO_LAUNCH(ctx, "aux_columns_kernel", (size_t)A * n * 8 * (1 + F::DEG), (aux_apply_kernel<F>), dim3(nblk, A), dim3(256), 0, trace, n, W, R, D, rands_dev, totals, out);
    ctx->check_launch("aux_columns");
}
template void launch_aux_columns<FB>(Context*, const uint64_t*, size_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint64_t*, uint64_t*);
template void launch_aux_columns<FQ>(Context*, const uint64_t*, size_t, uint32_t, uint32_t, uint32_t, uint32_t, const gl::E2*, uint64_t*);

// ------------------------------------------------------------------------------------------------
// Grinding (random.cairo:282-316 mirror): smallest nonce >= 1 whose BLAKE2s(seed || LE64(nonce)) has at least
// `bits` leading zero bits (MSB-first from digest byte 0). One launch tests a contiguous batch of nonces, every
// hit does an atomicMin; the host reads the result and launches the next batch only if the batch had no hit, so the
// minimum is exact and no workgroup ever polls another workgroup's result (per-XCD L2s are not coherent: a polled
// word can stay stale for the whole kernel).
__device__ __forceinline__ uint32_t leading_zeros_be(const Digest& d) {
    uint32_t w0 = __builtin_bswap32(d.w[0]), w1 = __builtin_bswap32(d.w[1]);
    if (w0) return __clz(w0);
    if (w1) return 32 + __clz(w1);
    return 64;
}
__global__ __launch_bounds__(256) void grind_kernel(Digest seed, uint32_t bits, uint64_t first, unsigned long long* best, unsigned long long* reset) {
    const uint64_t nonce = first + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) *reset = ~0ull;         // the slot the NEXT launch takes its minimum in
    Digest d = b2s::merge_with_int(seed, nonce);
    if (leading_zeros_be(d) >= bits) atomicMin(best, (unsigned long long)nonce);
}
// Returns the smallest nonce >= 1 with the required leading zeros (synchronises the stream once per batch). The minimum is taken in one of
// two device words that the launches use in turn: every launch also resets the OTHER word, so no launch needs a fill in front of it (one
// small launch less on the critical path of every proof).
uint64_t run_grind(Context* ctx, const Digest& seed, uint32_t bits) {
    if (bits == 0) return 1;
    if (!ctx->grind_slots) {
        ctx->grind_slots = (unsigned long long*)ctx->dev_alloc(16);
        AERO_HIP(hipMemsetAsync(ctx->grind_slots, 0xff, 16, ctx->stream));
        ctx->grind_parity = 0;
    }
    // batch = 4x the expected number of trials (a batch without a hit has probability e^-4), at least 2^16
    uint64_t batch = 1ull << (bits + 2 < 16 ? 16 : bits + 2);
    if (batch > (1ull << 30)) batch = 1ull << 30;
    for (uint64_t first = 1;; first += batch) {
        unsigned long long* slot = ctx->grind_slots + ctx->grind_parity;
        unsigned long long* other = ctx->grind_slots + (ctx->grind_parity ^ 1);
        ctx->grind_parity ^= 1;
        AERO_LAUNCH(ctx, "grind_kernel", 0, grind_kernel, dim3((unsigned)(batch / 256)), dim3(256), 0, seed, bits, first, slot, other);
        ctx->check_launch("grind");
        unsigned long long best = 0;
        ctx->fetch(&best, slot, 8);
        if (best != ~0ull) return best;
        if (first > (1ull << 40)) fail("grind: no nonce found", ST_INTERNAL);
    }
}

// ------------------------------------------------------------------------------------------------
// Input validation: field elements handed over by the host must be canonical (winter's BaseElement::new reduces on
// construction; this boundary takes raw u64 and refuses anything >= p instead of silently computing with it).
__global__ __launch_bounds__(256) void canonical_check_kernel(const uint64_t* __restrict__ v, size_t count, unsigned int* bad) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    unsigned int any = 0;
    for (; i < count; i += stride) any |= v[i] >= gl::P ? 1u : 0u;
    if (any) atomicOr(bad, 1u);
}
bool all_canonical(Context* ctx, const uint64_t* vals, size_t count) {
    if (count == 0) return true;
    DevBuf<unsigned int> d_bad(ctx, 1);
    AERO_HIP(hipMemsetAsync(d_bad.get(), 0, 4, ctx->stream));
    size_t blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    AERO_LAUNCH(ctx, "canonical_check_kernel", count * 8, canonical_check_kernel, dim3((unsigned)blocks), dim3(256), 0, vals, count, d_bad.get());
    ctx->check_launch("canonical_check");
    unsigned int bad = 0;
    AERO_HIP(hipMemcpyAsync(&bad, d_bad.get(), 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    return bad == 0;
}

// v <- v mod p in place (what Felt::new does to a raw u64): for data that arrives in the reference's message formats
__global__ __launch_bounds__(256) void reduce_canonical_kernel(uint64_t* __restrict__ v, size_t count) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < count; i += stride) { const uint64_t x = v[i]; if (x >= gl::P) v[i] = x - gl::P; }
}
void reduce_canonical(Context* ctx, uint64_t* vals, size_t count) {
    if (!count) return;
    size_t blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    AERO_LAUNCH(ctx, "reduce_canonical_kernel", count * 8, reduce_canonical_kernel, dim3((unsigned)blocks), dim3(256), 0, vals, count);
    ctx->check_launch("reduce_canonical");
}
// kernel only: ORs a 1 into *d_bad (device) when a value is not canonical
void canonical_check_accumulate(Context* ctx, const uint64_t* vals, size_t count, unsigned int* d_bad) {
    if (!count) return;
    size_t blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    AERO_LAUNCH(ctx, "canonical_check_kernel", count * 8, canonical_check_kernel, dim3((unsigned)blocks), dim3(256), 0, vals, count, d_bad);
    ctx->check_launch("canonical_check");
}
// Same check without waiting for it: the verdict lands in *h_bad_pinned (pinned host memory, 0 = all canonical) once the stream
// reaches this point; the caller reads it after its next stream synchronisation.
void canonical_check_enqueue(Context* ctx, const uint64_t* vals, size_t count, unsigned int* h_bad_pinned) {
    unsigned int* d_bad = (unsigned int*)ctx->scratch_alloc(4);
    AERO_HIP(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    size_t blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (count) AERO_LAUNCH(ctx, "canonical_check_kernel", count * 8, canonical_check_kernel, dim3((unsigned)blocks), dim3(256), 0, vals, count, d_bad);
    ctx->check_launch("canonical_check");
    AERO_HIP(hipMemcpyAsync(h_bad_pinned, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
}

// ------------------------------------------------------------------------------------------------
// Field-arithmetic self test. The device formulations of add / mul (32-bit carry chains, gl_field.hpp), the shift multiplications of
// the NTT butterflies (dft_small.hpp) and the F_p^2 operations are compared, on the device they run on and inside a kernel
// that mixes them the way the real kernels do, with results the host computed with plain 128-bit arithmetic. A carry-chain
// `sub` combined with the carry-chain `mul` once miscompiled inside deep_kernel (ROCm 7.2; each alone was exact): this test
// exists so that a compiler or driver change that breaks a formulation is reported by name instead of as a wrong proof.
template <int K> __device__ __forceinline__ uint64_t selftest_pow2(uint64_t x) { return mul_pow2<K>(x); }
__global__ __launch_bounds__(256) void field_selftest_kernel(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, size_t n, uint64_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t x = a[i], y = b[i];
    uint64_t* o = out + i * 17;
    o[0] = gl::add(x, y);
    o[1] = gl::sub(x, y);
    o[2] = gl::mul(x, y);
    o[3] = gl::sqr(x);
    o[4] = gl::neg(x);
    // the mix of deep_kernel / the butterflies: differences and sums feeding multiplications
    o[5] = gl::mul(gl::sub(x, y), gl::add(x, y));
    o[6] = gl::sub(gl::mul(x, y), gl::mul(y, y));
    o[7] = mul_2_24(x);
    o[8] = mul_w4(x);
    o[9] = mul_2_72(x);
    o[10] = gl::add(selftest_pow2<39>(x), selftest_pow2<78>(y));       // w_64, w_32
    o[11] = gl::sub(selftest_pow2<60>(x), selftest_pow2<7>(y));
    o[12] = selftest_pow2<95>(x);
    const gl::E2 e = gl::mul(gl::E2{x, y}, gl::E2{y, gl::add(x, 1)});
    o[13] = e.a0; o[14] = e.a1;
    o[15] = (i & 63) == 0 ? gl::mul(gl::inv(x), x) : 1;              // inversion on a sample (x != 0 by construction)
    // 160-bit sum of products (gl::Wide): 40 terms that make the limbs carry, reduced once
    gl::Wide w = gl::wzero();
    uint64_t u = x, v = y;
    for (int k = 0; k < 40; k++) { gl::wmac(w, u, v); u = gl::sub(gl::P - 1, mul_2_24(u)); v = gl::add(v, x); }
    o[16] = gl::wreduce(w);
}
// every shift multiplication of the register transforms: out[i * 96 + K] = x_i * 2^K, K < 96
template <int K> __device__ __forceinline__ void pow2_all(uint64_t x, uint64_t* o) {
    o[K] = mul_pow2<K>(x);
    if constexpr (K + 1 < 96) pow2_all<K + 1>(x, o);
}
__global__ __launch_bounds__(256) void pow2_selftest_kernel(const uint64_t* __restrict__ a, size_t n, uint64_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    pow2_all<0>(a[i], out + i * 96);
}
static uint64_t ref_mulmod(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % gl::P); }
static uint64_t ref_addmod(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + b) % gl::P); }
static uint64_t ref_submod(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + gl::P - b) % gl::P); }
static uint64_t ref_pow2(uint64_t a, int k) { uint64_t r = a; for (int i = 0; i < k; i++) r = ref_addmod(r, r); return r; }
void field_selftest(Context* ctx, size_t n, uint64_t seed) {
    std::vector<uint64_t> a(n), b(n);
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    auto next = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t i = 0; i < n; i++) {
        a[i] = next() % gl::P; b[i] = next() % gl::P;
        if (a[i] == 0) a[i] = 1;
    }
    // edge values: 0 / 1 / p - 1 / 2^32 - 1 / 2^32 / 2^63 around the carry boundaries of the limb arithmetic
    const uint64_t edge[] = {1, gl::P - 1, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFF00000000ull, 1ull << 63, gl::P - 0xFFFFFFFFull, 2};
    for (size_t i = 0; i < 64 && i < n; i++) { a[i] = edge[i & 7]; b[i] = i < 8 ? 0 : edge[(i >> 3) & 7]; }
    DevBuf<uint64_t> da(ctx, n), db(ctx, n), dout(ctx, n * 17);
    AERO_HIP(hipMemcpyAsync(da.get(), a.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    AERO_HIP(hipMemcpyAsync(db.get(), b.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    AERO_LAUNCH(ctx, "field_selftest_kernel", 0, field_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, da.get(), db.get(), n, dout.get());
    ctx->check_launch("field_selftest");
    std::vector<uint64_t> out(n * 17);
    AERO_HIP(hipMemcpyAsync(out.data(), dout.get(), n * 17 * 8, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    static const char* names[17] = {"add", "sub", "mul", "sqr", "neg", "mul(sub, add)", "sub(mul, mul)", "mul_2_24", "mul_w4 (2^48)", "mul_2_72",
                                    "add(pow2<39>, pow2<78>)", "sub(pow2<60>, pow2<7>)", "pow2<95>", "E2 mul (component 0)", "E2 mul (component 1)", "inv",
                                    "160-bit sum of 40 products (Wide)"};
    {
        // all 96 shift multiplications on the values whose shifted limbs straddle p: x = 2^(64 - r) - 1 - d puts (x << r) mod 2^64 at or just
        // below 2^64 - 2^r (>= p for r < 32), plus the generic edge values and the first random samples
        std::vector<uint64_t> xs;
        for (int r = 1; r < 32; r++) for (uint64_t d = 0; d < 3; d++) xs.push_back((((uint64_t)1 << (64 - r)) - 1 - d) % gl::P);
        for (uint64_t e : edge) xs.push_back(e % gl::P);
        xs.push_back(0); xs.push_back(gl::P - 2);
        for (size_t i = 0; i < n && i < 1024; i++) xs.push_back(a[i]);
        const size_t m = xs.size();
        DevBuf<uint64_t> dx(ctx, m), dp(ctx, m * 96);
        AERO_HIP(hipMemcpyAsync(dx.get(), xs.data(), m * 8, hipMemcpyHostToDevice, ctx->stream));
        AERO_LAUNCH(ctx, "pow2_selftest_kernel", 0, pow2_selftest_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, dx.get(), m, dp.get());
        ctx->check_launch("pow2_selftest");
        std::vector<uint64_t> po(m * 96);
        AERO_HIP(hipMemcpyAsync(po.data(), dp.get(), m * 96 * 8, hipMemcpyDeviceToHost, ctx->stream));
        ctx->sync();
        for (size_t i = 0; i < m; i++) {
            uint64_t want = xs[i];
            for (int k = 0; k < 96; k++) {
                if (po[i * 96 + k] != want) {
                    char msg[256];
                    snprintf(msg, sizeof msg, "field self-test: device mul_pow2<%d> is wrong for x = %llu: got %llu, expected %llu", k, (unsigned long long)xs[i],
                             (unsigned long long)po[i * 96 + k], (unsigned long long)want);
                    fail(msg, ST_INTERNAL);
                }
                want = ref_addmod(want, want);
            }
        }
    }
    for (size_t i = 0; i < n; i++) {
        const uint64_t x = a[i], y = b[i];
        uint64_t want[17];
        want[0] = ref_addmod(x, y); want[1] = ref_submod(x, y); want[2] = ref_mulmod(x, y); want[3] = ref_mulmod(x, x); want[4] = ref_submod(0, x);
        want[5] = ref_mulmod(ref_submod(x, y), ref_addmod(x, y));
        want[6] = ref_submod(ref_mulmod(x, y), ref_mulmod(y, y));
        want[7] = ref_pow2(x, 24); want[8] = ref_pow2(x, 48); want[9] = ref_pow2(x, 72);
        want[10] = ref_addmod(ref_pow2(x, 39), ref_pow2(y, 78));
        want[11] = ref_submod(ref_pow2(x, 60), ref_pow2(y, 7));
        want[12] = ref_pow2(x, 95);
        {   // (x + y phi)(y + (x + 1) phi), phi^2 = phi - 2
            const uint64_t b1 = ref_addmod(x, 1), a0b0 = ref_mulmod(x, y), a1b1 = ref_mulmod(y, b1);
            want[13] = ref_submod(a0b0, ref_addmod(a1b1, a1b1));
            want[14] = ref_submod(ref_mulmod(ref_addmod(x, y), ref_addmod(y, b1)), a0b0);
        }
        want[15] = 1;
        {
            uint64_t acc = 0, u = x, v = y;
            for (int k = 0; k < 40; k++) { acc = ref_addmod(acc, ref_mulmod(u, v)); u = ref_submod(gl::P - 1, ref_pow2(u, 24)); v = ref_addmod(v, x); }
            want[16] = acc;
        }
        for (int k = 0; k < 17; k++)
            if (out[i * 17 + k] != want[k]) {
                char msg[256];
                snprintf(msg, sizeof msg, "field self-test: device %s is wrong for x = %llu, y = %llu: got %llu, expected %llu", names[k],
                         (unsigned long long)x, (unsigned long long)y, (unsigned long long)out[i * 17 + k], (unsigned long long)want[k]);
                fail(msg, ST_INTERNAL);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Query gathers: rows of a column-major matrix at given positions; digests at given node indices.
__global__ void gather_rows_kernel(const uint64_t* cols, size_t col_stride, int ncols, const uint64_t* pos, int npos, uint64_t* out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npos * ncols) return;
    int q = t / ncols, c = t % ncols;
    const uint64_t p = pos[q];
    out[t] = p == GATHER_SKIP ? 0 : cols[(size_t)c * col_stride + p];
}
// FRI rows: out[q][j][d] = comp[d][pos[q] + j*rows]
__global__ void gather_fri_rows_kernel(const uint64_t* c0, const uint64_t* c1, int deg, size_t rows, int fold, const uint64_t* pos, int npos, uint64_t* out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npos * fold * deg) return;
    int d = t % deg, j = (t / deg) % fold, q = t / (deg * fold);
    const uint64_t* c = d ? c1 : c0;
    const uint64_t p = pos[q];
    out[t] = p == GATHER_SKIP ? 0 : c[p + (size_t)j * rows];
}
__global__ void gather_digests_kernel(const Digest* nodes, const uint64_t* idx, int n, Digest* out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint64_t i = idx[t];
    Digest z{};
    out[t] = i == GATHER_SKIP ? z : nodes[i];
}
// All openings of a proof in ONE launch: item t < n_u64 copies the u64 at device address addr[t]; the following n_dig items copy
// the 32-byte digest at addr[n_u64 + i] to out + n_u64 + 4 i. Address 0 yields zeros (item owned by another shard, or a low tree
// node that merkle_recompute fills in afterwards).
__global__ __launch_bounds__(256) void gather_addr_kernel(const uint64_t* __restrict__ addr, uint32_t n_u64, uint32_t n_dig, uint64_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < n_u64) {
        const uint64_t* p = reinterpret_cast<const uint64_t*>(addr[t]);
        out[t] = p ? *p : 0;
    } else if (t < n_u64 + n_dig) {
        const uint32_t i = t - n_u64;
        const ulonglong2* p = reinterpret_cast<const ulonglong2*>(addr[t]);
        ulonglong2 a = make_ulonglong2(0, 0), b = a;
        if (p) { a = p[0]; b = p[1]; }
        uint64_t* o = out + n_u64 + 4 * (size_t)i;
        o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
    }
}
void launch_gather_addr(Context* ctx, const uint64_t* addr, uint32_t n_u64, uint32_t n_dig, uint64_t* out) {
    const uint32_t n = n_u64 + n_dig;
    if (!n) return;
    AERO_LAUNCH(ctx, "gather_addr_kernel", 0, gather_addr_kernel, dim3((n + 255) / 256), dim3(256), 0, addr, n_u64, n_dig, out);
    ctx->check_launch("gather_addr");
}
void launch_gather_rows(Context* ctx, const uint64_t* cols, size_t col_stride, int ncols, const uint64_t* pos, int npos, uint64_t* out) {
    int n = npos * ncols;
    AERO_LAUNCH(ctx, "gather_rows_kernel", 0, gather_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, cols, col_stride, ncols, pos, npos, out);
    ctx->check_launch("gather_rows");
}
void launch_gather_fri_rows(Context* ctx, const uint64_t* c0, const uint64_t* c1, int deg, size_t rows, int fold, const uint64_t* pos, int npos, uint64_t* out) {
    int n = npos * fold * deg;
    AERO_LAUNCH(ctx, "gather_fri_rows_kernel", 0, gather_fri_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, c0, c1, deg, rows, fold, pos, npos, out);
    ctx->check_launch("gather_fri_rows");
}
void launch_gather_digests(Context* ctx, const Digest* nodes, const uint64_t* idx, int n, Digest* out) {
    AERO_LAUNCH(ctx, "gather_digests_kernel", 0, gather_digests_kernel, dim3((n + 255) / 256), dim3(256), 0, nodes, idx, n, out);
    ctx->check_launch("gather_digests");
}


// ------------------------------------------------------------------------------------------------
// Merge `parts` equally long pieces into interleaved order (coset shards -> natural order): out[u*parts + k] = in[k*src_stride + u].
// One thread per u: `parts` coalesced read streams, one contiguous write of parts*sizeof(T) bytes per thread.
template <class T> __global__ __launch_bounds__(256) void interleave_kernel(const T* __restrict__ in, size_t src_stride, T* __restrict__ out, int parts, size_t len) {
    const size_t u = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= len) return;
    for (int k = 0; k < parts; k++) out[u * parts + k] = in[(size_t)k * src_stride + u];
}
void launch_interleave_digests(Context* ctx, const Digest* in, size_t src_stride, Digest* out, int parts, size_t len) {
    AERO_LAUNCH(ctx, "interleave_kernel", 2 * len * parts * sizeof(Digest), (interleave_kernel<Digest>), dim3((unsigned)((len + 255) / 256)), dim3(256), 0, in, src_stride, out, parts, len);
    ctx->check_launch("interleave_digests");
}
// out[t] = the value at global LDE row J = (first + t * step) mod rows_total, taken from the all-gathered block in[owner][local]
// (owner = J mod parts, local = J / parts): the points of one coset of the constraint domain out of the shards of `parts` ranks.
__global__ void select_coset_kernel(const uint64_t* __restrict__ in, size_t src_stride, uint64_t* __restrict__ out, size_t count, uint32_t first,
                                    size_t step, size_t rows_total, uint32_t parts) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const size_t J = (first + t * step) & (rows_total - 1);
    out[t] = in[(J % parts) * src_stride + J / parts];
}
void launch_select_coset_u64(Context* ctx, const uint64_t* in, size_t src_stride, uint64_t* out, size_t count, uint32_t first, size_t step,
                             size_t rows_total, uint32_t parts) {
    AERO_LAUNCH(ctx, "select_coset_kernel", 16 * count, select_coset_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, in, src_stride, out, count,
                first, step, rows_total, parts);
    ctx->check_launch("select_coset");
}
// data[c][p] *= h^rev(p) for every column c: coefficients in bit-reversed order take the coset pre-scaling h^i after the fact
// (the sharded host hand-over all-gathers PLAIN coefficients; every rank then applies its own coset offset). h^i from a two-level
// table of powers of h; one thread walks all columns of its position, so the factor is formed once per position.
__global__ __launch_bounds__(256) void scale_pow_bitrev_kernel(uint64_t* __restrict__ data, size_t n, int ncols, int log_n, const uint64_t* __restrict__ lo,
                                                             const uint64_t* __restrict__ hi, int lo_bits) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t i = gl::bitrev((uint32_t)p, log_n);
    const uint64_t f = gl::mul(lo[i & ((1u << lo_bits) - 1)], hi[i >> lo_bits]);
    for (int c = 0; c < ncols; c++) data[(size_t)c * n + p] = gl::mul(data[(size_t)c * n + p], f);
}
void launch_scale_pow_bitrev(Context* ctx, uint64_t* data, size_t n, int ncols, int log_n, const uint64_t* lo, const uint64_t* hi, int lo_bits) {
    AERO_LAUNCH(ctx, "scale_pow_bitrev_kernel", (size_t)ncols * n * 16, scale_pow_bitrev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, data, n, ncols, log_n, lo, hi, lo_bits);
    ctx->check_launch("scale_pow_bitrev");
}
void launch_interleave_u64(Context* ctx, const uint64_t* in, size_t src_stride, uint64_t* out, int parts, size_t len) {
    AERO_LAUNCH(ctx, "interleave_kernel", 2 * len * parts * 8, (interleave_kernel<uint64_t>), dim3((unsigned)((len + 255) / 256)), dim3(256), 0, in, src_stride, out, parts, len);
    ctx->check_launch("interleave_u64");
}

}  // namespace aero
