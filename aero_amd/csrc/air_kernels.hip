// Device interpreter of compiled AEROAIR programs (air_program.hpp) for gfx950.
//
// Reference seam replaced: `ConstraintEvaluator::evaluate_fragment(&trace_lde, &domain, &mut fragment)` for ANY `Air`
// (/root/reference/aero-sdk/miden-wasm/src/constraints_worker.rs:32-59; divisor list and fragment stitching
// proving_worker.rs:374-437) and the `build_aux_segment` step of `commit_to_trace_and_validate` (proving_worker.rs:323-332).
//
// Execution model. One lane = R rows of the constraint-evaluation domain, interpreted in lockstep; one wavefront = one
// workgroup, so control flow is the PROGRAM's control flow and uniform by construction: the opcode dispatch is a scalar branch
// and is paid once per R x 64 rows. Nothing an instruction needs is reached through a dependent scalar load: instructions are
// 32-byte words fetched two ahead, and every scalar they can use (constants, public inputs, random elements, folded nodes, the
// (alpha, beta') pair of the constraint or assertion they feed) lies in one per-proof pool whose block for instruction i + 1 is
// requested while instruction i executes. Per-lane state: a register file in LDS - slot s of row q of lane l at word
// (s R + q) 64 + l, bank-conflict-free 8-byte accesses - sized by the host's register allocation; LOAD instructions bring eight
// frame values per row at a time from the column-major LDE (lane <-> row, coalesced, all loads of a batch in flight together);
// the transition accumulators (sum alpha_k t_k, sum beta'_k t_k per degree group) and the boundary accumulators of up to four
// divisor groups live in VGPRs. A degree adjustment x^adj is ONE lookup in the two-level table of the domain generator
// (x^adj = offset^adj w^(s adj mod rows), offset^adj folded into beta' on the host). The boundary divisors x^a - b of a lane's R
// rows are inverted with one field inversion (Montgomery's trick in registers), so the fused division costs 1/R inversion a row.
#include "air_kernels.hpp"

namespace aero {

using gl::FB;
using gl::FQ;
using namespace air;

constexpr int AIR_WG = 64;

__device__ __forceinline__ uint64_t air_tw(const uint64_t* __restrict__ lo, const uint64_t* __restrict__ hi, uint64_t e, int h) {
    return gl::mul(lo[e & ((1ull << h) - 1)], hi[e >> h]);
}

// ---- the interpreter core: R rows per lane ------------------------------------------------------------------------------------
template <class F, int R> struct AirLane {
    typedef typename F::T T;
    const uint64_t* main;
    const uint64_t* aux;
    const uint64_t* ptab;
    const uint64_t* pool;
    size_t stride;
    uint32_t r[R], rn[R], s[R];      // matrix row of the frame's current / next row, evaluation row (periodic index)
    uint64_t *ldsB, *ldsE;           // register file, already offset by the lane
};
struct PoolBlock { uint64_t w[4]; };
template <class F, int R> __device__ __forceinline__ uint64_t air_get_b(const AirLane<F, R>& c, const PoolBlock& P, uint32_t kind, uint32_t idx, int q) {
    return kind == D_SLOT_B ? c.ldsB[(idx * R + q) * AIR_WG] : P.w[0];
}
template <class F, int R> __device__ __forceinline__ typename F::T air_get_e(const AirLane<F, R>& c, const PoolBlock& P, uint32_t kind, uint32_t idx, int q) {
    if (kind == D_SLOT_E) return F::make(c.ldsE[((idx * F::DEG) * R + q) * AIR_WG], F::DEG > 1 ? c.ldsE[((idx * F::DEG + 1) * R + q) * AIR_WG] : 0);
    if (kind == D_SCAL_E) return F::make(P.w[0], P.w[1]);
    return F::from(air_get_b<F, R>(c, P, kind, idx, q));
}
template <class F, int R> __device__ __forceinline__ void air_put_e(const AirLane<F, R>& c, uint32_t slot, int q, typename F::T v) {
    c.ldsE[((slot * F::DEG) * R + q) * AIR_WG] = F::comp(v, 0);
    if (F::DEG > 1) c.ldsE[((slot * F::DEG + 1) * R + q) * AIR_WG] = F::comp(v, 1);
}
// OP_ADD_B .. OP_MULB_E on all R rows: operands first, then the arithmetic (R independent chains), then the stores
template <class F, int R> __device__ __forceinline__ void air_arith(const AirLane<F, R>& c, const PoolBlock& P, uint32_t op, uint32_t ka, uint32_t kb, uint32_t dst, uint32_t ia, uint32_t ib) {
    typedef typename F::T T;
    if (op <= OP_MUL_B) {
        uint64_t x[R], y[R];
#pragma unroll
        for (int q = 0; q < R; q++) { x[q] = air_get_b<F, R>(c, P, ka, ia, q); y[q] = air_get_b<F, R>(c, P, kb, ib, q); }
#pragma unroll
        for (int q = 0; q < R; q++) x[q] = op == OP_ADD_B ? gl::add(x[q], y[q]) : op == OP_SUB_B ? gl::sub(x[q], y[q]) : gl::mul(x[q], y[q]);
#pragma unroll
        for (int q = 0; q < R; q++) c.ldsB[(dst * R + q) * AIR_WG] = x[q];
    } else if (op == OP_MULB_E) {
        T x[R];
        uint64_t y[R];
#pragma unroll
        for (int q = 0; q < R; q++) { x[q] = air_get_e<F, R>(c, P, ka, ia, q); y[q] = air_get_b<F, R>(c, P, kb, ib, q); }
#pragma unroll
        for (int q = 0; q < R; q++) air_put_e<F, R>(c, dst, q, F::mulb(x[q], y[q]));
    } else {
        T x[R], y[R];
#pragma unroll
        for (int q = 0; q < R; q++) { x[q] = air_get_e<F, R>(c, P, ka, ia, q); y[q] = air_get_e<F, R>(c, P, kb, ib, q); }
#pragma unroll
        for (int q = 0; q < R; q++) air_put_e<F, R>(c, dst, q, op == OP_ADD_E ? F::add(x[q], y[q]) : op == OP_SUB_E ? F::sub(x[q], y[q]) : F::mul(x[q], y[q]));
    }
}
// LOAD: the descriptors are the one thing an instruction fetches itself; all R x 8 loads are issued before the first store
template <class F, int R> __device__ __forceinline__ void air_load_main(const AirLane<F, R>& c, uint32_t poff) {
    uint64_t v[LOAD_WIDTH][R];
    uint64_t d[LOAD_WIDTH];
#pragma unroll
    for (uint32_t k = 0; k < LOAD_WIDTH; k++) {
        d[k] = c.pool[poff + k];
        const uint32_t lo = (uint32_t)d[k];
        if (d[k] >> 62 & 1) {
            const uint32_t mask = (1u << (uint32_t)((d[k] >> 44) & 63)) - 1;
#pragma unroll
            for (int q = 0; q < R; q++) v[k][q] = c.ptab[lo + (c.s[q] & mask)];
        } else {
            const uint64_t* col = c.main + (size_t)lo * c.stride;
            const bool nx = d[k] >> 63;
#pragma unroll
            for (int q = 0; q < R; q++) v[k][q] = col[nx ? c.rn[q] : c.r[q]];
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < LOAD_WIDTH; k++) {
        const uint32_t slot = (uint32_t)(d[k] >> 32) & 0xfff;
#pragma unroll
        for (int q = 0; q < R; q++) c.ldsB[(slot * R + q) * AIR_WG] = v[k][q];
    }
}
template <class F, int R> __device__ __forceinline__ void air_load_aux(const AirLane<F, R>& c, uint32_t poff) {
    uint64_t v[LOAD_WIDTH / 2][R][2];
    uint64_t d[LOAD_WIDTH / 2];
#pragma unroll
    for (uint32_t k = 0; k < LOAD_WIDTH / 2; k++) {
        d[k] = c.pool[poff + k];
        const uint64_t* col = c.aux + (size_t)((uint32_t)d[k] * F::DEG) * c.stride;
        const bool nx = d[k] >> 63;
#pragma unroll
        for (int q = 0; q < R; q++) {
            const uint32_t row = nx ? c.rn[q] : c.r[q];
            v[k][q][0] = col[row];
            v[k][q][1] = F::DEG > 1 ? col[c.stride + row] : 0;
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < LOAD_WIDTH / 2; k++) {
        const uint32_t slot = (uint32_t)(d[k] >> 32) & 0xfff;
#pragma unroll
        for (int q = 0; q < R; q++) air_put_e<F, R>(c, slot, q, F::make(v[k][q][0], v[k][q][1]));
    }
}
__device__ __forceinline__ uint4 air_word(const uint4* code, uint32_t pc, uint32_t half) { return code[2 * pc + half]; }

// MODE 0: numerator columns (the reference's ConstraintEvaluationTable seam); MODE 1: divided by the divisors and summed (H).
// NB = boundary divisor groups whose accumulators and inverse divisors are held in registers (n_bgroups <= NB); NB = 0: any number
// of groups, accumulators in LDS behind the register file, one inversion per row (R = 1).
template <class F, int MODE, int R, int NB> __global__ __launch_bounds__(AIR_WG) void air_constraints_kernel(AirConsArgs<F> a) {
    typedef typename F::T T;
    static_assert(NB > 0 || R == 1, "the LDS-accumulator form interprets one row per lane");
    constexpr int NBR = NB > 0 ? NB : 1;
    extern __shared__ uint64_t air_lds[];
    const uint32_t lane = threadIdx.x;
    const size_t nthreads = a.count / R;
    size_t t = (size_t)blockIdx.x * AIR_WG + lane;
    const bool active = t < nthreads;
    if (!active) t = nthreads - 1;        // the wavefront stays convergent: idle lanes shadow the last rows and store nothing
    const uint64_t rmask = a.rows - 1;
    const uint32_t nb = a.n_bgroups;
    AirLane<F, R> c;
    c.main = a.lde; c.aux = a.aux; c.ptab = a.ptab; c.pool = a.pool; c.stride = a.N;
    c.ldsB = air_lds + lane;
    c.ldsE = air_lds + (size_t)a.slotsB * R * AIR_WG + lane;
    uint64_t* const ldsG = air_lds + ((size_t)a.slotsB + (size_t)a.slotsE * F::DEG) * R * AIR_WG + lane;   // NB == 0: [group][sa, sb][component]
    uint64_t srow[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
        const uint64_t s = a.first + t + (uint64_t)q * nthreads;
        srow[q] = s;
        size_t r = (size_t)s * a.ce_step;
        size_t rn = (r + a.blowup) & (a.N - 1);
        if (a.split_log) {
            const size_t part_len = a.N >> a.split_log, pm = ((size_t)1 << a.split_log) - 1;
            r = (r & pm) * part_len + (r >> a.split_log);
            rn = (rn & pm) * part_len + (rn >> a.split_log);
        }
        c.r[q] = (uint32_t)r; c.rn[q] = (uint32_t)rn; c.s[q] = (uint32_t)s;
    }
    // ---- boundary divisors x^a - b of the lane's rows, inverted with ONE field inversion
    uint64_t dinv[R][NBR];
    if (MODE == 1 && NB > 0) {
        uint64_t pre[R][NBR];
        uint64_t run = 1;
#pragma unroll
        for (int q = 0; q < R; q++)
#pragma unroll
            for (int j = 0; j < NBR; j++) {
                dinv[q][j] = 1; pre[q][j] = run;
                if ((uint32_t)j < nb) {
                    const AirBGroupDev g = a.bgroups[j];
                    dinv[q][j] = gl::sub(gl::mul(g.ha, air_tw(a.tw_lo, a.tw_hi, (srow[q] * g.a_exp) & rmask, a.tw_h)), g.b);
                    run = gl::mul(run, dinv[q][j]);
                }
            }
        uint64_t ia = gl::inv(run);
#pragma unroll
        for (int q = R - 1; q >= 0; q--)
#pragma unroll
            for (int j = NBR - 1; j >= 0; j--)
                if ((uint32_t)j < nb) { const uint64_t d = dinv[q][j]; dinv[q][j] = gl::mul(ia, pre[q][j]); ia = gl::mul(ia, d); }
    }
    if (NB == 0) for (uint32_t i = 0; i < 2 * nb * F::DEG; i++) ldsG[i * AIR_WG] = 0;
    // ---- the program
    T acc_a[R], acc_b[R], total[R], gsa[R][NBR], gsb[R][NBR];
#pragma unroll
    for (int q = 0; q < R; q++) {
        acc_a[q] = acc_b[q] = total[q] = F::zero();
#pragma unroll
        for (int j = 0; j < NBR; j++) gsa[q][j] = gsb[q][j] = F::zero();
    }
    const uint4* const code = reinterpret_cast<const uint4*>(a.code);
    uint4 I = air_word(code, 0, 0);
    uint32_t Ipoff = air_word(code, 0, 1).x;
    uint4 In = air_word(code, 1, 0);
    uint32_t Inpoff = air_word(code, 1, 1).x;
    PoolBlock P;
#pragma unroll
    for (int k = 0; k < 4; k++) P.w[k] = a.pool[Ipoff + k];
#pragma unroll 1
    for (uint32_t pc = 2;; pc++) {
        // two instructions and one pool block ahead (the stream ends with spare END words, the pool with a spare block)
        const uint4 Inn = air_word(code, pc, 0);
        const uint32_t Innpoff = air_word(code, pc, 1).x;
        PoolBlock Pn;
#pragma unroll
        for (int k = 0; k < 4; k++) Pn.w[k] = a.pool[Inpoff + k];
        const uint32_t op = I.x & 0xff, ka = (I.x >> 8) & 0xf, kb = (I.x >> 12) & 0xf;
        if (op == OP_END) break;
        if (op <= OP_MULB_E) {
            air_arith<F, R>(c, P, op, ka, kb, I.y, I.z, I.w);
        } else if (op == OP_LOAD_MAIN) {
            air_load_main<F, R>(c, Ipoff);
        } else if (op == OP_EMIT_B) {
            const T ca = F::make(P.w[0], P.w[1]), cb = F::DEG > 1 ? F::make(P.w[2], P.w[3]) : F::make(P.w[1], 0);
#pragma unroll
            for (int q = 0; q < R; q++) {
                const uint64_t v = c.ldsB[(I.z * R + q) * AIR_WG];
                acc_a[q] = F::add(acc_a[q], F::mulb(ca, v));
                acc_b[q] = F::add(acc_b[q], F::mulb(cb, v));
            }
        } else if (op == OP_EMIT3_B) {            // the last two nodes of the constraint folded into the EMIT: a o1 (b o2 c) / (b o2 c) o1 a
            const T ca = F::make(P.w[0], P.w[1]), cb = F::DEG > 1 ? F::make(P.w[2], P.w[3]) : F::make(P.w[1], 0);
            const uint32_t o1 = (I.x >> 16) & 3, o2 = (I.x >> 18) & 3, left = (I.x >> 20) & 1;
#pragma unroll
            for (int q = 0; q < R; q++) {
                const uint64_t x = c.ldsB[(I.z * R + q) * AIR_WG], y = c.ldsB[(I.w * R + q) * AIR_WG], z = c.ldsB[(I.y * R + q) * AIR_WG];
                const uint64_t in = o2 == 1 ? gl::add(y, z) : o2 == 2 ? gl::sub(y, z) : gl::mul(y, z);
                const uint64_t l = left ? in : x, r = left ? x : in;
                const uint64_t v = o1 == 1 ? gl::add(l, r) : o1 == 2 ? gl::sub(l, r) : gl::mul(l, r);
                acc_a[q] = F::add(acc_a[q], F::mulb(ca, v));
                acc_b[q] = F::add(acc_b[q], F::mulb(cb, v));
            }
        } else if (op >= OP_EMIT_ADD_B) {         // OP_EMIT_ADD_B / SUB / MUL: the root operation folded into the EMIT
            const T ca = F::make(P.w[0], P.w[1]), cb = F::DEG > 1 ? F::make(P.w[2], P.w[3]) : F::make(P.w[1], 0);
#pragma unroll
            for (int q = 0; q < R; q++) {
                const uint64_t x = c.ldsB[(I.z * R + q) * AIR_WG], y = c.ldsB[(I.w * R + q) * AIR_WG];
                const uint64_t v = op == OP_EMIT_ADD_B ? gl::add(x, y) : op == OP_EMIT_SUB_B ? gl::sub(x, y) : gl::mul(x, y);
                acc_a[q] = F::add(acc_a[q], F::mulb(ca, v));
                acc_b[q] = F::add(acc_b[q], F::mulb(cb, v));
            }
        } else if (op == OP_BOUND_B || op == OP_BOUND_E) {
            const T ca = F::make(P.w[0], P.w[1]), cb = F::DEG > 1 ? F::make(P.w[2], P.w[3]) : F::make(P.w[1], 0);
            const uint32_t g = I.y;
            T pa[R], pb[R];
#pragma unroll
            for (int q = 0; q < R; q++) {
                if (op == OP_BOUND_B) { const uint64_t v = c.ldsB[(I.z * R + q) * AIR_WG]; pa[q] = F::mulb(ca, v); pb[q] = F::mulb(cb, v); }
                else { const T v = air_get_e<F, R>(c, P, D_SLOT_E, I.z, q); pa[q] = F::mul(ca, v); pb[q] = F::mul(cb, v); }
            }
            if (NB > 0) {
                // the group is uniform: a scalar branch picks the accumulator registers (no per-lane selects)
#pragma unroll
                for (int j = 0; j < NBR; j++)
                    if (__builtin_amdgcn_readfirstlane(g) == (uint32_t)j) {
#pragma unroll
                        for (int q = 0; q < R; q++) { gsa[q][j] = F::add(gsa[q][j], pa[q]); gsb[q][j] = F::add(gsb[q][j], pb[q]); }
                    }
            }
#pragma unroll
            for (int q = 0; q < R; q++) {
                if (NB > 0) {
                } else {
                    uint64_t* gp = ldsG + (size_t)g * 2 * F::DEG * AIR_WG;
                    const T sa = F::add(F::make(gp[0], F::DEG > 1 ? gp[AIR_WG] : 0), pa[q]);
                    const T sb = F::add(F::make(gp[F::DEG * AIR_WG], F::DEG > 1 ? gp[(F::DEG + 1) * AIR_WG] : 0), pb[q]);
                    for (int d = 0; d < F::DEG; d++) { gp[d * AIR_WG] = F::comp(sa, d); gp[(F::DEG + d) * AIR_WG] = F::comp(sb, d); }
                }
            }
        } else if (op == OP_EMIT_E) {
            const T ca = F::make(P.w[0], P.w[1]), cb = F::DEG > 1 ? F::make(P.w[2], P.w[3]) : F::make(P.w[1], 0);
#pragma unroll
            for (int q = 0; q < R; q++) {
                const T v = air_get_e<F, R>(c, P, D_SLOT_E, I.z, q);
                acc_a[q] = F::add(acc_a[q], F::mul(ca, v));
                acc_b[q] = F::add(acc_b[q], F::mul(cb, v));
            }
        } else if (op == OP_LOAD_AUX) {
            air_load_aux<F, R>(c, Ipoff);
        } else {                                  // OP_GROUP_END
            const uint64_t e = a.dg_exp[I.y];
#pragma unroll
            for (int q = 0; q < R; q++) {
                total[q] = F::add(total[q], F::mulb(acc_b[q], air_tw(a.tw_lo, a.tw_hi, (srow[q] * e) & rmask, a.tw_h)));
                acc_b[q] = F::zero();
            }
        }
        I = In; Ipoff = Inpoff; P = Pn; In = Inn; Inpoff = Innpoff;
    }
    // ---- per row: transition numerator (/ divisor), boundary groups (sa - A) + x^adj (sb - B) (/ divisor)
#pragma unroll
    for (int q = 0; q < R; q++) {
        const uint64_t s = srow[q];
        const size_t o = (size_t)(s - a.first);
        const T tot = F::add(total[q], acc_a[q]);
        T h = F::zero();
        if (MODE == 0) {
            if (active) for (int d = 0; d < F::DEG; d++) a.out_cols[(size_t)d * a.count + o] = F::comp(tot, d);
        } else {
            // 1 / ((x^n - 1) / prod (x - w^(n-i))) = prod (x - w^(n-i)) * (x^n - 1)^-1
            const uint64_t x = gl::mul(a.offset, air_tw(a.tw_lo, a.tw_hi, s & rmask, a.tw_h));
            uint64_t tdiv = a.zn_inv[s & a.xmask];
            for (uint32_t i = 0; i < a.n_exempt; i++) tdiv = gl::mul(tdiv, gl::sub(x, a.exempt[i]));
            h = F::mulb(tot, tdiv);
        }
        if (NB > 0) {
#pragma unroll
            for (int j = 0; j < NBR; j++) {
                if ((uint32_t)j >= nb) continue;
                const T sa = F::sub(gsa[q][j], a.gA[j]), sb = F::sub(gsb[q][j], a.gB[j]);
                T gnum = F::add(sa, F::mulb(sb, air_tw(a.tw_lo, a.tw_hi, (s * a.bgroups[j].adj_exp) & rmask, a.tw_h)));
                if (a.bgroups[j].seq && active) {      // sequence assertions: the row's share of the values (air_host.hip: seq_tables)
                    const uint64_t* tp = a.seq_tab + (size_t)(a.bgroups[j].seq - 1) * F::DEG * a.rows + s;
                    gnum = F::sub(gnum, F::make(tp[0], F::DEG > 1 ? tp[a.rows] : 0));
                }
                if (MODE == 0) { if (active) for (int d = 0; d < F::DEG; d++) a.out_cols[(size_t)((1 + j) * F::DEG + d) * a.count + o] = F::comp(gnum, d); }
                else h = F::add(h, F::mulb(gnum, dinv[q][j]));
            }
        } else {
            // any number of groups: Montgomery over this row's divisors through LDS (prefix products behind the accumulators)
            uint64_t* const ldsD = ldsG + (size_t)2 * nb * F::DEG * AIR_WG;
            uint64_t* const ldsP = ldsD + (size_t)nb * AIR_WG;
            uint64_t ia = 0;
            if (MODE == 1) {
                uint64_t run = 1;
                for (uint32_t j = 0; j < nb; j++) {
                    const AirBGroupDev g = a.bgroups[j];
                    const uint64_t d = gl::sub(gl::mul(g.ha, air_tw(a.tw_lo, a.tw_hi, (s * g.a_exp) & rmask, a.tw_h)), g.b);
                    ldsD[j * AIR_WG] = d; ldsP[j * AIR_WG] = run;
                    run = gl::mul(run, d);
                }
                ia = gl::inv(run);
            }
            for (uint32_t j = nb; j-- > 0;) {
                const uint64_t* gp = ldsG + (size_t)j * 2 * F::DEG * AIR_WG;
                const T sa = F::sub(F::make(gp[0], F::DEG > 1 ? gp[AIR_WG] : 0), a.gA[j]);
                const T sb = F::sub(F::make(gp[F::DEG * AIR_WG], F::DEG > 1 ? gp[(F::DEG + 1) * AIR_WG] : 0), a.gB[j]);
                T gnum = F::add(sa, F::mulb(sb, air_tw(a.tw_lo, a.tw_hi, (s * a.bgroups[j].adj_exp) & rmask, a.tw_h)));
                if (a.bgroups[j].seq && active) {      // sequence assertions: the row's share of the values (air_host.hip: seq_tables)
                    const uint64_t* tp = a.seq_tab + (size_t)(a.bgroups[j].seq - 1) * F::DEG * a.rows + s;
                    gnum = F::sub(gnum, F::make(tp[0], F::DEG > 1 ? tp[a.rows] : 0));
                }
                if (MODE == 0) { if (active) for (int d = 0; d < F::DEG; d++) a.out_cols[(size_t)((1 + j) * F::DEG + d) * a.count + o] = F::comp(gnum, d); }
                else { h = F::add(h, F::mulb(gnum, gl::mul(ia, ldsP[j * AIR_WG]))); ia = gl::mul(ia, ldsD[j * AIR_WG]); }
            }
        }
        if (MODE == 1 && active) for (int d = 0; d < F::DEG; d++) a.out_h[d][s] = F::comp(h, d);
    }
}

static void air_set_lds(const void* kern, size_t bytes) {
    if (bytes > 160 * 1024) fail("air program: the interpreter's register file does not fit the 160 KiB of LDS", ST_UNSUPPORTED);
    if (bytes > 48 * 1024) AERO_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}
template <class F, int MODE, int R, int NB> static bool air_launch_variant(Context* ctx, const AirConsArgs<F>& a, size_t abytes) {
    const size_t slots = (size_t)a.slotsB + (size_t)a.slotsE * F::DEG;
    size_t words = slots * R;
    if (NB == 0) words += (size_t)a.n_bgroups * (2 * F::DEG + 2);
    const size_t lds = (words ? words : 1) * AIR_WG * 8;
    if (lds > 160 * 1024) return false;
    air_set_lds((const void*)air_constraints_kernel<F, MODE, R, NB>, lds);
    const size_t nthreads = a.count / R;
    AERO_LAUNCH(ctx, "air_constraints_kernel", abytes, (air_constraints_kernel<F, MODE, R, NB>), dim3((unsigned)((nthreads + AIR_WG - 1) / AIR_WG)), dim3(AIR_WG), lds, a);
    ctx->check_launch("air_constraints");
    return true;
}
template <class F, int MODE> static bool air_launch_mode(Context* ctx, const AirConsArgs<F>& a, size_t abytes) {
    // Rows per lane. Measured on fib_2^20 x 72 (MI355X, tools/air_bench.py): 2 rows per lane 1.48 ms, 4 rows 1.71 ms (170 VGPRs and
    // 16 KiB of LDS per wave leave 2.5 waves per SIMD), 1 row with LDS accumulators 1.94 ms.
    constexpr int RW = 2;
    const bool wide = a.count % 4 == 0 && a.count >= (size_t)4 * AIR_WG * 64;       // enough rows to fill the chip several at a time
    if (wide && a.n_bgroups <= 2 && air_launch_variant<F, MODE, RW, 2>(ctx, a, abytes)) return true;
    if (wide && a.n_bgroups <= 4 && air_launch_variant<F, MODE, RW, 4>(ctx, a, abytes)) return true;
    // a VM's assertions (first / last / interior steps, a few periodic strides) make 5 - 8 divisor groups: their accumulators still fit
    // the registers (2 rows x 8 groups x 2 sums), which keeps the read-modify-write of every BOUND out of the LDS
    if (wide && a.n_bgroups <= 8 && air_launch_variant<F, MODE, RW, 8>(ctx, a, abytes)) return true;
    return air_launch_variant<F, MODE, 1, 0>(ctx, a, abytes);
}
template <class F> bool launch_air_constraints(Context* ctx, const AirConsArgs<F>& a, int mode) {
    const size_t in_cols = (size_t)a.W + (size_t)a.A * F::DEG;
    if (mode == 0) {
        if (!air_launch_mode<F, 0>(ctx, a, a.count * 8 * (in_cols + (1 + a.n_bgroups) * F::DEG)))
            fail("air program: the interpreter's register file does not fit the 160 KiB of LDS", ST_UNSUPPORTED);
        return true;
    }
    return air_launch_mode<F, 1>(ctx, a, a.count * 8 * (in_cols + F::DEG));
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
__global__ __launch_bounds__(256) void air_scatter_kernel(uint64_t* __restrict__ dst, const uint64_t* __restrict__ pos, const uint64_t* __restrict__ val, size_t count) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) dst[pos[i]] = val[i];
}
void launch_air_scatter(Context* ctx, uint64_t* dst, const uint64_t* pos, const uint64_t* val, size_t count) {
    if (!count) return;
    AERO_LAUNCH(ctx, "air_scatter_kernel", count * 24, air_scatter_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, dst, pos, val, count);
    ctx->check_launch("air_scatter");
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
    AirLane<F, 1> c;
    c.main = a.trace; c.aux = nullptr; c.ptab = a.ptab; c.pool = a.pool; c.stride = a.n;
    c.r[0] = (uint32_t)i; c.rn[0] = (uint32_t)((i + 1) & (a.n - 1)); c.s[0] = (uint32_t)i;
    c.ldsB = air_lds + lane;
    c.ldsE = air_lds + (size_t)a.slotsB * AIR_WG + lane;
    const uint4* const code = reinterpret_cast<const uint4*>(a.code);
#pragma unroll 1
    for (uint32_t pc = 0;; pc++) {
        const uint4 I = air_word(code, pc, 0);
        const uint32_t poff = air_word(code, pc, 1).x;
        const uint32_t op = I.x & 0xff, ka = (I.x >> 8) & 0xf, kb = (I.x >> 12) & 0xf;
        if (op == OP_END) break;
        PoolBlock P;
        P.w[0] = a.pool[poff]; P.w[1] = a.pool[poff + 1]; P.w[2] = P.w[3] = 0;
        if (op <= OP_MULB_E) {
            air_arith<F, 1>(c, P, op, ka, kb, I.y, I.z, I.w);
        } else if (op == OP_LOAD_MAIN) {
            air_load_main<F, 1>(c, poff);
        } else {                                  // OP_OUT_B / OP_OUT_E
            const T v = op == OP_OUT_B ? F::from(air_get_b<F, 1>(c, P, ka, I.z, 0)) : air_get_e<F, 1>(c, P, ka, I.z, 0);
            if (active) for (int d = 0; d < F::DEG; d++) fac[((size_t)I.y * F::DEG + d) * a.n + i] = F::comp(v, d);
        }
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
template <class F> __global__ __launch_bounds__(256) void air_fac_totals_kernel(const uint64_t* fac, size_t n, const uint8_t* has_den, const uint8_t* has_add, typename F::T* totals) {
    typedef typename F::T T;
    __shared__ T sh[256];
    const uint32_t j = blockIdx.y;
    if (((j & 1) && !has_den[j >> 1]) || has_add[j >> 1]) return;        // affine columns take their own path (air_aff_*)
    const size_t first = ((size_t)blockIdx.x * 256 + threadIdx.x) * SCAN_K;
    T p = F::one();
#pragma unroll
    for (int k = 0; k < SCAN_K; k++) p = F::mul(p, air_fac_at<F>(fac, n, j, first + k));
    const T inc = air_wg_scan_mul<F>(p, sh);
    if (threadIdx.x == 255) totals[(size_t)j * gridDim.x + blockIdx.x] = inc;
}
// in place: totals[j][b] <- product of totals[j][0..b) (one workgroup per factor column)
template <class F> __global__ __launch_bounds__(256) void air_scan_totals_kernel(typename F::T* totals, uint32_t nblk, const uint8_t* has_den, const uint8_t* has_add) {
    typedef typename F::T T;
    __shared__ T sh[256];
    const uint32_t j = blockIdx.x;
    if (((j & 1) && !has_den[j >> 1]) || has_add[j >> 1]) return;
    T* row = totals + (size_t)j * nblk;
    const uint32_t per = (nblk + 255) / 256, lo = threadIdx.x * per;
    T p = F::one();
    for (uint32_t i = lo; i < lo + per && i < nblk; i++) p = F::mul(p, row[i]);
    air_wg_scan_mul<F>(p, sh);
    T run = threadIdx.x == 0 ? F::one() : sh[threadIdx.x - 1];
    for (uint32_t i = lo; i < lo + per && i < nblk; i++) { const T v = row[i]; row[i] = run; run = F::mul(run, v); }
}
// grid (blocks, A): column c = init * prefix(num) / prefix(den)
template <class F> __global__ __launch_bounds__(256) void air_fac_apply_kernel(const uint64_t* fac, size_t n, const uint8_t* has_den, const uint8_t* has_add,
                                                                            const typename F::T* init, const typename F::T* totals, uint64_t* out) {
    typedef typename F::T T;
    __shared__ T sh[256];
    const uint32_t c = blockIdx.y;
    if (has_add[c]) return;
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
        const T tot = F::mul(rd, pd);
        if (F::comp(tot, 0) == 0 && (F::DEG == 1 || F::comp(tot, 1) == 0)) {
            // a zero denominator among this lane's rows (or before them): element by element, 1 / 0 = 0 as winter-math's inversion has it - the rows
            // BEFORE the zero keep their values (one shared inversion of the lane's whole product would zero them too), the rows behind it are zero
            T pre = rd;
#pragma unroll
            for (int k = 0; k < SCAN_K; k++) { val[k] = F::mul(val[k], F::inv(pre)); pre = F::mul(pre, fd[k]); }
        } else {
            T inv = F::inv(tot);
#pragma unroll
            for (int k = SCAN_K - 1; k >= 0; k--) { inv = F::mul(inv, fd[k]); val[k] = F::mul(val[k], inv); }
        }
    }
#pragma unroll
    for (int k = 0; k < SCAN_K; k++)
        if (first + k < n)
            for (int d = 0; d < F::DEG; d++) out[((size_t)c * F::DEG + d) * n + first + k] = F::comp(val[k], d);
}
// ---- affine builders (AEROAIR version 2): column(i + 1) = column(i) * m_i + t_i with m_i = num_i / den_i, t_i = add_i / add_den_i.
// Row maps x -> m x + t compose associatively, (m2, t2) o (m1, t1) = (m1 m2, t1 m2 + t2), so the column is a prefix scan over pairs -
// the running product above is the special case t = 0. Step 0 turns the factor columns of an affine column into m and t (in
// place: one inversion per lane for the up to 2 SCAN_K denominators of its rows); then block totals, scan of the totals, apply.
template <class F> struct AffPair { typename F::T m, t; };
template <class F> __device__ __forceinline__ AffPair<F> aff_then(const AffPair<F>& first, const AffPair<F>& second) {      // second o first
    return AffPair<F>{F::mul(first.m, second.m), F::add(F::mul(first.t, second.m), second.t)};
}
template <class F> __device__ __forceinline__ AffPair<F> air_wg_scan_aff(AffPair<F> v, typename F::T* shm, typename F::T* sht) {   // inclusive, 256 lanes
    const int t = threadIdx.x;
    shm[t] = v.m; sht[t] = v.t;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        AffPair<F> u{shm[t], sht[t]};
        if (t >= off) u = aff_then<F>(AffPair<F>{shm[t - off], sht[t - off]}, u);
        __syncthreads();
        shm[t] = u.m; sht[t] = u.t;
        __syncthreads();
    }
    return AffPair<F>{shm[t], sht[t]};
}
template <class F> __device__ __forceinline__ void air_fac_put(uint64_t* fac, size_t n, uint32_t j, size_t i, typename F::T v) {
    const size_t o = (size_t)(j * F::DEG) * n + i;
    for (int d = 0; d < F::DEG; d++) fac[o + (size_t)d * n] = F::comp(v, d);
}
// grid (blocks, A): factor columns of an affine column -> m (in column 2c) and t (in column 2A + 2c)
template <class F> __global__ __launch_bounds__(256) void air_aff_ratio_kernel(uint64_t* fac, size_t n, uint32_t A, const uint8_t* has_den, const uint8_t* has_add) {
    typedef typename F::T T;
    const uint32_t c = blockIdx.y;
    if (!has_add[c] || has_add[c] == 4) return;
    const bool hd = has_den[c], ad = has_add[c] & 2;
    if (!hd && !ad) return;
    const size_t first = ((size_t)blockIdx.x * 256 + threadIdx.x) * SCAN_K;
    T den[2 * SCAN_K], pre[2 * SCAN_K];
    T run = F::one();
#pragma unroll
    for (int k = 0; k < SCAN_K; k++) {
        den[k] = hd ? air_fac_at<F>(fac, n, 2 * c + 1, first + k) : F::one();
        den[SCAN_K + k] = ad ? air_fac_at<F>(fac, n, 2 * A + 2 * c + 1, first + k) : F::one();
    }
    // one inversion for the lane's 2 SCAN_K denominators; a ZERO among them is skipped in the product and inverted to zero (winter-math's
    // batch_inversion does the same): only the offending row is affected, not every row of the lane
    bool zero[2 * SCAN_K];
#pragma unroll
    for (int k = 0; k < 2 * SCAN_K; k++) {
        zero[k] = F::comp(den[k], 0) == 0 && (F::DEG == 1 || F::comp(den[k], 1) == 0);
        pre[k] = run;
        if (!zero[k]) run = F::mul(run, den[k]);
    }
    T inv = F::inv(run);
#pragma unroll
    for (int k = 2 * SCAN_K - 1; k >= 0; k--) {                                                     // den[k] <- 1 / den[k]
        if (zero[k]) { den[k] = F::zero(); continue; }
        const T di = F::mul(inv, pre[k]); inv = F::mul(inv, den[k]); den[k] = di;
    }
#pragma unroll
    for (int k = 0; k < SCAN_K; k++) {
        if (first + k >= n) continue;
        if (hd) air_fac_put<F>(fac, n, 2 * c, first + k, F::mul(air_fac_at<F>(fac, n, 2 * c, first + k), den[k]));
        if (ad) air_fac_put<F>(fac, n, 2 * A + 2 * c, first + k, F::mul(air_fac_at<F>(fac, n, 2 * A + 2 * c, first + k), den[SCAN_K + k]));
    }
}
template <class F> __device__ __forceinline__ AffPair<F> air_aff_at(const uint64_t* fac, size_t n, uint32_t A, uint32_t c, size_t i) {
    if (i >= n) return AffPair<F>{F::one(), F::zero()};
    const size_t om = (size_t)(2 * c * F::DEG) * n + i, ot = (size_t)((2 * A + 2 * c) * F::DEG) * n + i;
    return AffPair<F>{F::make(fac[om], F::DEG > 1 ? fac[om + n] : 0), F::make(fac[ot], F::DEG > 1 ? fac[ot + n] : 0)};
}
// grid (blocks, A): composition of each block's rows; totals[c][b] = (m, t)
template <class F> __global__ __launch_bounds__(256) void air_aff_totals_kernel(const uint64_t* fac, size_t n, uint32_t A, const uint8_t* has_add, AffPair<F>* totals) {
    typedef typename F::T T;
    __shared__ T shm[256], sht[256];
    const uint32_t c = blockIdx.y;
    if (!has_add[c] || has_add[c] == 4) return;      // 4: general recurrence, built on the host
    const size_t first = ((size_t)blockIdx.x * 256 + threadIdx.x) * SCAN_K;
    AffPair<F> p{F::one(), F::zero()};
#pragma unroll
    for (int k = 0; k < SCAN_K; k++) p = aff_then<F>(p, air_aff_at<F>(fac, n, A, c, first + k));
    const AffPair<F> inc = air_wg_scan_aff<F>(p, shm, sht);
    if (threadIdx.x == 255) totals[(size_t)c * gridDim.x + blockIdx.x] = inc;
}
// in place: totals[c][b] <- composition of totals[c][0..b) (one workgroup per column)
template <class F> __global__ __launch_bounds__(256) void air_aff_scan_totals_kernel(AffPair<F>* totals, uint32_t nblk, const uint8_t* has_add) {
    typedef typename F::T T;
    __shared__ T shm[256], sht[256];
    const uint32_t c = blockIdx.x;
    if (!has_add[c] || has_add[c] == 4) return;
    AffPair<F>* row = totals + (size_t)c * nblk;
    const uint32_t per = (nblk + 255) / 256, lo = threadIdx.x * per;
    AffPair<F> p{F::one(), F::zero()};
    for (uint32_t i = lo; i < lo + per && i < nblk; i++) p = aff_then<F>(p, row[i]);
    air_wg_scan_aff<F>(p, shm, sht);
    AffPair<F> run = threadIdx.x == 0 ? AffPair<F>{F::one(), F::zero()} : AffPair<F>{shm[threadIdx.x - 1], sht[threadIdx.x - 1]};
    for (uint32_t i = lo; i < lo + per && i < nblk; i++) { const AffPair<F> v = row[i]; row[i] = run; run = aff_then<F>(run, v); }
}
// grid (blocks, A): column c(i) = (prefix map up to row i)(init)
template <class F> __global__ __launch_bounds__(256) void air_aff_apply_kernel(const uint64_t* fac, size_t n, uint32_t A, const uint8_t* has_add, const typename F::T* init,
                                                                            const AffPair<F>* totals, uint64_t* out) {
    typedef typename F::T T;
    __shared__ T shm[256], sht[256];
    const uint32_t c = blockIdx.y;
    if (!has_add[c] || has_add[c] == 4) return;      // 4: general recurrence, built on the host
    const size_t first = ((size_t)blockIdx.x * 256 + threadIdx.x) * SCAN_K;
    AffPair<F> f[SCAN_K];
    AffPair<F> p{F::one(), F::zero()};
#pragma unroll
    for (int k = 0; k < SCAN_K; k++) { f[k] = air_aff_at<F>(fac, n, A, c, first + k); p = aff_then<F>(p, f[k]); }
    air_wg_scan_aff<F>(p, shm, sht);
    AffPair<F> before = totals[(size_t)c * gridDim.x + blockIdx.x];
    if (threadIdx.x) before = aff_then<F>(before, AffPair<F>{shm[threadIdx.x - 1], sht[threadIdx.x - 1]});
    T x = F::add(F::mul(init[c], before.m), before.t);
#pragma unroll
    for (int k = 0; k < SCAN_K; k++) {
        if (first + k < n)
            for (int d = 0; d < F::DEG; d++) out[((size_t)c * F::DEG + d) * n + first + k] = F::comp(x, d);
        x = F::add(F::mul(x, f[k].m), f[k].t);
    }
}
template <class F> void launch_air_aux(Context* ctx, const AirAuxArgs<F>& a, const std::vector<uint8_t>& has_den_host, const std::vector<uint8_t>& has_add_host) {
    typedef typename F::T T;
    const size_t n = a.n;
    const uint32_t A = a.A;
    bool any_affine = false, any_product = false;
    for (uint32_t c = 0; c < A; c++) { const uint8_t k = c < has_add_host.size() ? has_add_host[c] : 0; if (k == 4) continue; if (k) any_affine = true; else any_product = true; }
    uint64_t* fac = (uint64_t*)ctx->scratch_alloc((size_t)(any_affine ? 4 : 2) * A * F::DEG * n * 8);
    const size_t slots = (size_t)a.slotsB + (size_t)a.slotsE * F::DEG;
    const size_t lds = (slots ? slots : 1) * AIR_WG * 8;
    air_set_lds((const void*)air_aux_factors_kernel<F>, lds);
    AERO_LAUNCH(ctx, "air_aux_kernel", (size_t)n * 8 * (a.W + 2 * A * F::DEG), (air_aux_factors_kernel<F>), dim3((unsigned)((n + AIR_WG - 1) / AIR_WG)), dim3(AIR_WG), lds, a, fac);
    const uint32_t nblk = (uint32_t)((n + (size_t)SCAN_K * 256 - 1) / ((size_t)SCAN_K * 256));
    T* totals = (T*)ctx->scratch_alloc(sizeof(T) * (size_t)2 * A * nblk);
    (void)has_den_host;
    if (any_product) {
        AERO_LAUNCH(ctx, "air_aux_kernel", (size_t)A * n * 8 * F::DEG, (air_fac_totals_kernel<F>), dim3(nblk, 2 * A), dim3(256), 0, fac, n, a.has_den, a.has_add, totals);
        AERO_LAUNCH(ctx, "air_aux_kernel", 0, (air_scan_totals_kernel<F>), dim3(2 * A), dim3(256), 0, totals, nblk, a.has_den, a.has_add);
        AERO_LAUNCH(ctx, "air_aux_kernel", (size_t)A * n * 8 * 2 * F::DEG, (air_fac_apply_kernel<F>), dim3(nblk, A), dim3(256), 0, fac, n, a.has_den, a.has_add, a.init, totals, a.out);
    }
    if (any_affine) {
        AffPair<F>* atot = (AffPair<F>*)ctx->scratch_alloc(sizeof(AffPair<F>) * (size_t)A * nblk);
        AERO_LAUNCH(ctx, "air_aux_kernel", (size_t)A * n * 8 * 4 * F::DEG, (air_aff_ratio_kernel<F>), dim3(nblk, A), dim3(256), 0, fac, n, A, a.has_den, a.has_add);
        AERO_LAUNCH(ctx, "air_aux_kernel", (size_t)A * n * 8 * 2 * F::DEG, (air_aff_totals_kernel<F>), dim3(nblk, A), dim3(256), 0, (const uint64_t*)fac, n, A, a.has_add, atot);
        AERO_LAUNCH(ctx, "air_aux_kernel", 0, (air_aff_scan_totals_kernel<F>), dim3(A), dim3(256), 0, atot, nblk, a.has_add);
        AERO_LAUNCH(ctx, "air_aux_kernel", (size_t)A * n * 8 * 3 * F::DEG, (air_aff_apply_kernel<F>), dim3(nblk, A), dim3(256), 0, (const uint64_t*)fac, n, A, a.has_add, a.init, (const AffPair<F>*)atot, a.out);
    }
    ctx->check_launch("air_aux");
}
template void launch_air_aux<FB>(Context*, const AirAuxArgs<FB>&, const std::vector<uint8_t>&, const std::vector<uint8_t>&);
template void launch_air_aux<FQ>(Context*, const AirAuxArgs<FQ>&, const std::vector<uint8_t>&, const std::vector<uint8_t>&);

// ---- general auxiliary recurrence: one wavefront per column (air_kernels.hpp: AirGeneralArgs) ------------------------------------------
__device__ __forceinline__ uint64_t gen_readlane(uint64_t v, uint32_t lane) {
    return gl::mk64((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)lane), (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)lane));
}
__device__ __forceinline__ uint64_t gen_bcast(uint64_t v, uint32_t lane) { return gen_readlane(v, lane); }
__device__ __forceinline__ gl::E2 gen_bcast(gl::E2 v, uint32_t lane) { return gl::E2{gen_readlane(v.a0, lane), gen_readlane(v.a1, lane)}; }
template <class F> __global__ __launch_bounds__(64) void air_general_column_kernel(AirGeneralArgs<F> a) {
    typedef typename F::T T;
    extern __shared__ __attribute__((aligned(16))) unsigned char gen_lds_raw[];
    T* slots = reinterpret_cast<T*>(gen_lds_raw);              // [slot][lane]
    const uint32_t lane = threadIdx.x;
    const size_t n = a.n;
    uint64_t* out[2] = {a.aux + (size_t)a.col * F::DEG * n, a.aux + ((size_t)a.col * F::DEG + (F::DEG > 1 ? 1 : 0)) * n};
    auto apply = [](uint32_t op, T x, T y) { return op == 1 ? F::add(x, y) : op == 2 ? F::sub(x, y) : F::mul(x, y); };
    T x = a.init;
    if (lane == 0) for (int d = 0; d < F::DEG; d++) out[d][0] = F::comp(x, d);
    T sv = F::zero();                                           // value of serial node `lane`
    for (size_t base = 0; base + 1 < n; base += 64) {
        const size_t row = base + lane < n ? base + lane : n - 1;
        for (uint32_t k = 0; k < a.n_loads; k++) {
            const GenLoad L = a.loads[k];
            T v;
            if (L.kind == GLD_MAIN_CUR) v = F::from(a.trace[(size_t)L.col * n + row]);
            else if (L.kind == GLD_MAIN_NXT) v = F::from(a.trace[(size_t)L.col * n + (row + 1 < n ? row + 1 : row)]);
            else if (L.kind == GLD_AUX_CUR) v = F::make(a.aux[(size_t)L.col * F::DEG * n + row], F::DEG > 1 ? a.aux[((size_t)L.col * F::DEG + 1) * n + row] : 0);
            else v = F::from(a.ptab[L.col + ((uint32_t)row & L.mask)]);
            slots[L.slot * 64 + lane] = v;
        }
        for (uint32_t k = 0; k < a.n_par; k++) {
            const GenInsn I = a.par[k];
            const T u = I.ka == GOP_SLOT ? slots[I.ia * 64 + lane] : a.consts[I.ia];
            const T w = I.kb == GOP_SLOT ? slots[I.ib * 64 + lane] : a.consts[I.ib];
            slots[I.dst * 64 + lane] = apply(I.op, u, w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        T mine = F::zero();
        const uint32_t rows_here = (uint32_t)(n - 1 - base < 64 ? n - 1 - base : 64);        // rows base + r, r < rows_here, have a successor
        for (uint32_t r = 0; r < rows_here; r++) {
            auto operand = [&](uint32_t kind, uint32_t idx) -> T {
                if (kind == GOP_SLOT) return slots[idx * 64 + r];                                  // broadcast read
                if (kind == GOP_SER) return gen_bcast(sv, idx);
                if (kind == GOP_X) return x;
                return a.consts[idx];
            };
            for (uint32_t k = 0; k < a.n_ser; k++) {
                const GenInsn I = a.ser[k];
                const T v = apply(I.op, operand(I.ka, I.ia), operand(I.kb, I.ib));
                if (lane == I.dst) sv = v;
            }
            x = operand(a.res_kind, a.res_idx);
            if (lane == r) mine = x;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < rows_here) for (int d = 0; d < F::DEG; d++) out[d][base + lane + 1] = F::comp(mine, d);
    }
}
template <class F> void launch_air_general_column(Context* ctx, const AirGeneralArgs<F>& a) {
    const size_t lds = (size_t)GEN_MAX_SLOTS * 64 * sizeof(typename F::T);
    AERO_LAUNCH(ctx, "air_general_column_kernel", (size_t)a.n * 8 * (a.n_loads + F::DEG), air_general_column_kernel<F>, dim3(1), dim3(64), lds, a);
    ctx->check_launch("air_general_column");
}
template void launch_air_general_column<FB>(Context*, const AirGeneralArgs<FB>&);
template void launch_air_general_column<FQ>(Context*, const AirGeneralArgs<FQ>&);

}  // namespace aero
