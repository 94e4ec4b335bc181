// Argument blocks and launchers of the per-row STARK kernels (stark.hip).
#pragma once
#include "aero_internal.hpp"

namespace aero {

template <class F> struct FibConsArgs {
    typedef typename F::T T;
    const uint64_t* lde;       // trace LDE, column-major W x N
    size_t N;
    uint32_t W, C, blowup, ce_step;
    uint32_t split_log;        // rows of `lde` / `aux` are de-interleaved into 2^split_log parts (CompactOut); 0 = plain row order
    uint32_t xmask;            // x^n = h^n w^(s & xmask): C - 1 on the constraint domain; (rows / n) - 1 when the rows are a coset shard
    size_t first, count;       // ce rows [first, first + count)
    const T *ta, *tb, *ba, *bb;   // composition coefficient pairs (device): transition[W + A], boundary[W + W/2 + A]
    const uint64_t* results;      // W/2 public results (device)
    const uint64_t *tw_lo, *tw_hi, *twi_lo, *twi_hi;   // two-level tables of w_ce and its inverse
    int tw_h;
    uint64_t offset;              // domain offset h (7, or 7 w_N^rank for one coset shard of the LDE domain)
    uint64_t gen_inv, k7;         // h^-1, h^ce_n
    const uint64_t* xn_inv;       // C entries: (h^n w_C^k)^-1
    const uint64_t* zn_inv;       // C entries: (h^n w_C^k - 1)^-1
    uint64_t w_last;              // w_n^(n-1)
    const uint64_t* inv_tab;      // MODE 1, optional: the per-shape table of fib_inverse_table_kernel, five arrays of inv_tab_n words - (x - 1)^-1,
    size_t inv_tab_n;             //   (x - w_last)^-1, the transition divisor's inverse, the two x^adj; nullptr = the kernel computes them per row
    // auxiliary segment (A = 0: none): A columns over E stored as A*DEG base columns (index c*DEG + d) with the same row
    // count / stride as `lde`; p_c' = p_c * (rands[c mod R] + main[c mod W]); p_c(0) = 1
    const uint64_t* aux;
    uint32_t A, R, D;             // D = degree of the aux transition constraint: p' = p * (rand + main)^(D-1)
    const T* rands;               // R elements (device)
    const uint64_t* xn;           // C entries: (h^n w_C^k)^(C+1-D); times x^(D-2) = the degree adjustment x^((C+1-D)n + D-2) of the aux group
    uint64_t* out_cols;           // MODE 0: (3*DEG) x count, column-major
    uint64_t* out_h[2];           // MODE 1: DEG component arrays of ce_n values
};
template <class F> void launch_fib_constraints(Context* ctx, const FibConsArgs<F>& a, int mode);
// what the constraint kernel needs of every point x_s = offset * w_rows^s, s < c.count, that depends on the domain only: out[5 * c.count] (c: the
// launch's arguments with first = 0 and the device tables xn_inv / zn_inv in place)
template <class F> void launch_fib_inverse_table(Context* ctx, uint64_t* out, const FibConsArgs<F>& c);

// The division step of `ConstraintEvaluationTable::into_poly` for FibAir: H(x) = sum over the three divisor columns of
// numerator / divisor on the constraint-evaluation domain (row s <-> x = offset * w_ce^s).
template <class F> struct FibDivideArgs {
    const uint64_t* cols;       // (3*DEG) x ce_n numerators: [transition, boundary(step 0), boundary(step n-1)] x components
    size_t ce_n;
    uint32_t C;
    const uint64_t *tw_lo, *tw_hi;   // two-level table of w_ce
    int tw_h;
    uint64_t offset, w_last;
    const uint64_t* zn_inv;     // C entries: (offset^n w_C^k - 1)^-1
    uint64_t* out_h[2];
};
template <class F> void launch_fib_divide(Context* ctx, const FibDivideArgs<F>& a);

template <class F> struct EvalArgs {
    typedef typename F::T T;
    const uint64_t* coeffs;
    size_t col_stride, comp_stride;
    int comps, L, r, npts;
    T y0, y1;
    const T* ktab;
    T* partials;
};
template <class F>
void launch_eval_bitrev(Context* ctx, const uint64_t* coeffs, size_t col_stride, size_t comp_stride, int ncols, int comps, int L,
                        typename F::T y0, typename F::T y1, int npts, typename F::T* out);

// Several evaluations of the same length in THREE launches (tables, block sums, reduction) instead of three each: the OOD frame of a proof is
// the trace polynomials at (z, z g), the auxiliary ones at the same points and the composition columns at z^C - nine small launches in a row on
// the critical path of one proof. out index of (job, column, point) = job.out_off + column * npts + point.
constexpr int EVAL_MAX_JOBS = 3;
template <class F> struct EvalJob {
    typedef typename F::T T;
    const uint64_t* coeffs = nullptr;
    size_t col_stride = 0, comp_stride = 0;
    int ncols = 0, comps = 1, npts = 1;
    T y0, y1;                      // evaluation points (y1 unused when npts = 1)
    uint32_t out_off = 0;          // first output index of this job
};
template <class F> struct EvalMultiArgs {
    typedef typename F::T T;
    EvalJob<F> jobs[EVAL_MAX_JOBS];
    T b0[EVAL_MAX_JOBS], b1[EVAL_MAX_JOBS];     // y^(2^(L - r)) per job: bases of the in-block tables
    int n_jobs, L, r;
    T* ktab;                       // [job][2][2^r]
    T* partials;                   // [output index][block]
};
template <class F> void launch_eval_multi(Context* ctx, const EvalJob<F>* jobs, int n_jobs, int L, typename F::T* out);

template <class F> struct DeepArgs {
    typedef typename F::T T;
    const uint64_t* tlde;   // W x N
    const uint64_t* clde;   // (C*DEG) x N
    const uint64_t* alde;   // (A*DEG) x N auxiliary segment columns (E-valued, component columns), or nullptr
    uint32_t A;
    size_t N;                        // LDE rows: point m sits at LDE row m * row_step, x = h w_N^(m row_step)
    size_t count;                    // points evaluated (output index m < count)
    uint32_t row_step;
    // where point m lives in each matrix: column stride and row step (the full LDE: N / row_step; a compact every-k-th-row
    // copy written by the LDE's last pass: N / k and row_step / k)
    size_t t_stride, c_stride, a_stride;
    uint32_t t_step, c_step, a_step;
    uint32_t W, C;
    const uint64_t *tw_lo, *tw_hi;   // two-level table of w_N
    int tw_h;
    uint64_t offset;                 // domain offset h: row r <-> x = h w_N^r
    T z, z_next, z_c, z_conj, lambda, mu;
    const T *ood_cur, *ood_next, *ood_h, *da, *db, *dg, *dc;   // device; ood_cur/ood_next/da/db/dg cover main then aux columns
    uint64_t* out[2];
};
template <class F> void launch_deep(Context* ctx, const DeepArgs<F>& a);

// DEEP composition in COEFFICIENT form (base field, one GPU): the quotients (P(x) - P(z)) / (x - z) are synthetic divisions of the combined
// column polynomials - no field inversion and no interpolation of the result. All arrays are bit-reversed coefficient vectors pre-scaled by
// h^k (what the interpolation stage leaves and the LDE takes), so the divisions run at the points z / h:
//   out_k = lam * S_k + mu * S_(k-1),  S = Q_1 + Q_2 + Q_3,  Q_j = synthetic quotient of P_j at y_j,
//   P_1 = sum da_i T_i, P_2 = sum db_i T_i (main then aux columns), P_3 = sum dc_c H_(chunk c)        (lam = lambda / h, mu as drawn)
struct DeepCoeffArgs {
    const uint64_t* tpolys; size_t t_stride; uint32_t W;      // main segment polynomials
    const uint64_t* apolys; size_t a_stride; uint32_t A;      // auxiliary segment polynomials (base field: one component), or A = 0
    const uint64_t* hpolys; size_t h_stride; uint32_t C;      // composition column c = chunk c of this buffer (dc is given in chunk order)
    const uint64_t *da, *db, *dc;                             // device: W + A, W + A, C coefficients
    uint64_t y[3];                                            // division points z / h, z g / h, z^C / h
    uint64_t lam, mu;
    int log_n;                                                // n = 2^log_n coefficients
    uint64_t* blocks;                                         // scratch: deep_coeff_scratch_words(log_n) words
    uint64_t* out;                                            // n coefficients of the DEEP polynomial
};
inline size_t deep_coeff_scratch_words(int log_n) { return (size_t)5 << log_n; }   // [3][n] combined polynomials + two levels of block values (at most n / 4 each)
void launch_deep_coeff(Context* ctx, const DeepCoeffArgs& a);

template <class F> struct FoldArgs {
    typedef typename F::T T;
    const uint64_t* in[2];
    uint64_t* out[2];
    size_t rows;
    int fold;
    T alpha;
    const T* alpha_dev;                // non-null: the folding challenge is read from device memory (drawn by fri_coin_step)
    const uint64_t *twi_lo, *twi_hi;   // two-level table of w_dom^-1
    int tw_h;
    uint64_t gen_inv, fold_inv;        // inverse of the domain offset (7, or 7 w_dom^rank for a coset shard), 1/fold
    uint64_t dft[16];                  // w_F^-m, m < fold
};
template <class F> void launch_fri_fold(Context* ctx, const FoldArgs<F>& a);

// Auxiliary segment columns (a synthetic stand-in for Miden's multiset-check columns): for c < A
//   p_c(0) = 1,  p_c(i+1) = p_c(i) * (rands[c mod R] + trace[c mod W][i])^(D-1)      (a prefix product over the rows, in E)
// out = (A*DEG) x n column-major component columns (index c*DEG + d).
template <class F> void launch_aux_columns(Context* ctx, const uint64_t* trace, size_t n, uint32_t W, uint32_t A, uint32_t R, uint32_t D,
                                           const typename F::T* rands_dev, uint64_t* out);

// true iff every one of the `count` device values is a canonical field element (< p); synchronises the stream
bool all_canonical(Context* ctx, const uint64_t* vals, size_t count);
void reduce_canonical(Context* ctx, uint64_t* vals, size_t count);   // v <- v mod p in place (Felt::new)
void canonical_check_accumulate(Context* ctx, const uint64_t* vals, size_t count, unsigned int* d_bad);
void canonical_check_enqueue(Context* ctx, const uint64_t* vals, size_t count, unsigned int* h_bad_pinned);

// One transcript step of the FRI commit phase on the device (random.cairo:108-166 mirror): seed <- BLAKE2s(seed || root),
// then alpha = the first draw (counter 1, 2, ... until the 8-byte words are canonical). Lets the host enqueue every layer of
// the commit phase without waiting for a root.
template <class F> void launch_fri_coin_step(Context* ctx, Digest* seed_io, const Digest* root, typename F::T* alpha_out);

// device field arithmetic against a host reference on n random + edge-case operand pairs; throws ST_INTERNAL naming the operation
void field_selftest(Context* ctx, size_t n, uint64_t seed);

uint64_t run_grind(Context* ctx, const Digest& seed, uint32_t bits);
void launch_gather_rows(Context* ctx, const uint64_t* cols, size_t col_stride, int ncols, const uint64_t* pos, int npos, uint64_t* out);
void launch_gather_fri_rows(Context* ctx, const uint64_t* c0, const uint64_t* c1, int deg, size_t rows, int fold, const uint64_t* pos, int npos, uint64_t* out);
void launch_gather_digests(Context* ctx, const Digest* nodes, const uint64_t* idx, int n, Digest* out);
// In all three gathers an index of GATHER_SKIP yields zeros (the item belongs to another shard).
constexpr uint64_t GATHER_SKIP = ~0ull;
void launch_gather_addr(Context* ctx, const uint64_t* addr, uint32_t n_u64, uint32_t n_dig, uint64_t* out);
// out[u * parts + k] = in[k * src_stride + u], u < len: merges `parts` equally long pieces into their interleaved order
void launch_interleave_digests(Context* ctx, const Digest* in, size_t src_stride, Digest* out, int parts, size_t len);
void launch_interleave_u64(Context* ctx, const uint64_t* in, size_t src_stride, uint64_t* out, int parts, size_t len);
// data[c][p] *= h^rev_(log_n)(p), h^i = lo[i & (2^lo_bits - 1)] * hi[i >> lo_bits]
void launch_scale_pow_bitrev(Context* ctx, uint64_t* data, size_t n, int ncols, int log_n, const uint64_t* lo, const uint64_t* hi, int lo_bits);
void launch_select_coset_u64(Context* ctx, const uint64_t* in, size_t src_stride, uint64_t* out, size_t count, uint32_t first, size_t step,
                             size_t rows_total, uint32_t parts);

}  // namespace aero
