// Goldilocks NTT for gfx950: batched column transforms staged through LDS, radix-8 butterflies in registers.
//
// Replaces, for the reference's stage 1 (/root/reference/aero-sdk/miden-wasm/src/proving_worker.rs:271-274):
//   main_trace.interpolate_columns()            -> interpolate (evaluations on <w_n>  -> coefficients)
//   trace_polys.evaluate_columns_over(&domain)  -> lde         (coefficients -> evaluations on 7*<w_N>, natural row order)
// whose bodies live in winter-math 0.4 `fft` (submodule absent from the mount).
//
// Layout decisions (MI355X-first, not a translation of winter's in-place serial FFT):
//  * Column-major matrices, one transform per column, all columns of a pass in one launch (grid.y = column).
//  * Coefficients are kept in BIT-REVERSED order and PRE-SCALED by offset^i, so neither direction ever runs a
//    bit-reversal permutation pass and the LDE needs no separate coset-shift pass:
//        interpolate = decimation-in-frequency (natural in -> bit-reversed out),
//        lde         = decimation-in-time      (bit-reversed in -> natural out).
//  * A transform of size 2^L is split "four-step" style into passes: the first pass covers the low 12 index bits
//    (4096 contiguous elements = 32 KiB of LDS per workgroup, perfectly coalesced), the remaining bits are covered by
//    strided passes over tiles of R x TL elements (TL consecutive addresses per row, R*TL = 4096), so every HBM
//    access is a segment of at least TL*8 bytes and each pass reads and writes every element exactly once.
//    Between passes elements are multiplied by w_n^(S * rev(block) * k) (two-level twiddle table, L1-resident).
//  * The LDE is ONE transform of size N = blowup*n on the zero-padded coefficient vector: in bit-reversed order the
//    non-zero inputs sit at positions = 0 mod blowup, so the first log2(blowup) butterfly stages are a broadcast and
//    are skipped; the first pass reads n/... coefficients and writes `blowup` times as many values.
// Algorithmic HBM bytes per element and pass: 16 (8 read + 8 written); interpolate(2^20) = 2 passes, lde(2^20 -> 2^23)
// = 3 passes of which the first reads 1/8 of what it writes.
#include "aero_internal.hpp"
#include "dft_small.hpp"

namespace aero {

using gl::add;
using gl::mul;
using gl::sub;

// twiddle exponent e on w_n -> value, via two-level table: w^e = lo[e & (2^h - 1)] * hi[e >> h]
__device__ __forceinline__ uint64_t tw_lookup(const uint64_t* __restrict__ lo, const uint64_t* __restrict__ hi, uint32_t e, int h) {
    uint64_t a = lo[e & ((1u << h) - 1)];
    uint64_t b = hi[e >> h];
    return mul(a, b);
}

struct PassArgs {
    const uint64_t* in;
    uint64_t* out;
    size_t in_col_stride, out_col_stride;   // elements
    int log_n;        // transform size
    int log_s;        // stride of this pass (0 for the contiguous pass)
    int log_r;        // radix of this pass
    int log_tl;       // tile width (consecutive addresses per row)
    int log_pad;      // forward first pass only: input holds n >> log_pad coefficients (zero-padded transform)
    int first;        // forward: this is the pass with the largest stride? (no twiddle after it) / inverse: no twiddle before
    const uint64_t* tw_r;     // w_4096^j (j < 2048), forward or inverse roots
    const uint64_t* tw_lo;    // two-level table of w_n (or its inverse)
    const uint64_t* tw_hi;
    int tw_h;
    const uint64_t* tw_pass;  // register passes only: pass-boundary twiddles, row b * R + k
    const uint64_t* tw_mt;    // two-phase contiguous pass only: [r * 64 + k] = w_2048^(r k)
    // last pass only (optional): rows = 0 mod 2^compact_log are also written, densely, to compact[row >> compact_log]
    uint64_t* compact;
    size_t compact_col_stride;
    int compact_log, compact_split;
    // inverse contiguous pass only: out[p] *= ktab[k] * blockfac(block)
    const uint64_t* ktab;     // R entries: c0 * (a^(2^(L-r)) * b^(2^(L-r-shift)))^rev_r(k)
    uint64_t sc_a, sc_b;      // per-block factor = a^rev(block) * b^(rev(block) >> shift)
    int sc_shift;
    // inverse, first executed pass only (optional): OR 1 into *bad when an input element is not canonical (>= p) - the validation
    // of a trace handed over by the host rides on the pass that reads it anyway
    unsigned int* bad;
    int chains;       // ntt_fwd_first_pass_8: number of interleaved pass-boundary progressions (1 or 2)
};

// LDS index skew: one extra slot per 32 elements breaks the power-of-two strides of the radix-8 gathers
// (stride-8 element reads would otherwise put 8 lanes on one bank).
__device__ __forceinline__ int skew(int e) { return e + (e >> 5); }
constexpr int LDS_ELEMS = 4096 + 128;

// One register round over bits [s, s+RB) of the pass-local index k (tile element e = (k << log_tl) | tl, data in LDS).
// Cooley-Tukey step: the 2^RB sub-transform values of a group (bit-reversed order) are multiplied by T^rev(i),
// T = w_(2^(s+RB))^(k mod 2^s), then combined by the plain 2^RB-point DFT above. tw = w_4096^e (forward) or w_4096^-e.
template <int RB, bool INV, int NT = 256>
__device__ __forceinline__ void ntt_round(uint64_t* lds, int log_e, int log_tl, int s, const uint64_t* __restrict__ tw) {
    constexpr int G = 1 << RB;
    const int shift = s + log_tl;
    const int ngroups = 1 << (log_e - RB);
    const int tsh = 12 - s - RB;
    for (int g = threadIdx.x; g < ngroups; g += NT) {
        const int base = (g & ((1 << shift) - 1)) | ((g >> shift) << (shift + RB));
        const int klow = (base >> log_tl) & ((1 << s) - 1);
        uint64_t y[G];
#pragma unroll
        for (int i = 0; i < G; i++) y[i] = lds[skew(base + (i << shift))];
        if (!INV) {
            if (klow) {
#pragma unroll
                for (int i = 1; i < G; i++) y[i] = mul(y[i], tw[(klow * (int)gl::bitrev((uint32_t)i, RB)) << tsh]);
            }
            dft_dit<RB>(y);
        } else {
            dft_dif_inv<RB>(y);
            if (klow) {
#pragma unroll
                for (int i = 1; i < G; i++) y[i] = mul(y[i], tw[(klow * (int)gl::bitrev((uint32_t)i, RB)) << tsh]);
            }
        }
#pragma unroll
        for (int i = 0; i < G; i++) lds[skew(base + (i << shift))] = y[i];
    }
    __syncthreads();
}

// exchange between the two halves of a wavefront (V_PERMLANE32_SWAP): x of lanes 32..63 <-> y of lanes 0..31
__device__ __forceinline__ void swap_halves(uint64_t& x, uint64_t& y) {       // x of lanes 32..63 <-> y of lanes 0..31
    const auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)x, (uint32_t)y, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)(x >> 32), (uint32_t)(y >> 32), false, false);
    x = gl::mk64(lo[0], hi[0]); y = gl::mk64(lo[1], hi[1]);
}

// Workgroups are dispatched round-robin over the 8 XCDs (each with its own L2 and TLBs). Handing XCD x the x-th contiguous
// eighth of the tiles keeps the tiles that are in flight on one XCD next to each other in memory: a strided pass touches
// R rows per tile, and neighbouring tiles share those rows' pages and DRAM rows.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t bid, uint32_t nblocks) {
#ifdef AERO_NTT_LINEAR_TILES
    return bid;
#else
    return (nblocks & 7u) ? bid : (bid & 7u) * (nblocks >> 3) + (bid >> 3);
#endif
}

// Forward: decimation in time, bit-reversed input -> natural output (within the pass' index bits).
// NT = threads per 4096-element tile: 256 (16 elements per thread) or 512 (8 per thread: small launches, where a tile per CU leaves one
// wave per SIMD and the pass takes as long as ONE wave needs for its share - half the share, half the time).
template <int NT> __global__ __launch_bounds__(NT) void ntt_fwd_pass(PassArgs a) {
    __shared__ __attribute__((aligned(16))) uint64_t lds[LDS_ELEMS];
    const int TL = 1 << a.log_tl, log_e = a.log_r + a.log_tl, E = 1 << log_e;
    const uint32_t tile = xcd_tile(blockIdx.x, gridDim.x);
    const uint32_t tiles_per_b = 1u << (a.log_s - a.log_tl);       // S / TL
    const uint32_t b = tile >> (a.log_s - a.log_tl);
    const uint32_t lo0 = (tile & (tiles_per_b - 1)) << a.log_tl;
    const uint64_t* in = a.in + (size_t)blockIdx.y * a.in_col_stride;
    uint64_t* out = a.out + (size_t)blockIdx.y * a.out_col_stride;
    const size_t base = (size_t)lo0 + (((size_t)b << a.log_r) << a.log_s);

    if (a.log_pad == 0) {
        for (int e = threadIdx.x; e < E; e += NT) {
            int tl = e & (TL - 1), k = e >> a.log_tl;
            lds[skew(e)] = in[base + tl + ((size_t)k << a.log_s)];
        }
    } else {
        // contiguous pass of a zero-padded transform: position k of the block holds coefficient (b*R + k) >> log_pad
        // when k = 0 mod 2^log_pad, zero otherwise; the first log_pad butterfly stages therefore broadcast it.
        const size_t cbase = ((size_t)b << a.log_r) >> a.log_pad;
        for (int e = threadIdx.x; e < E; e += NT) lds[skew(e)] = in[cbase + (e >> a.log_pad)];
    }
    __syncthreads();
    for (int s = a.log_pad; s < a.log_r;) {
        const int rb = a.log_r - s >= 3 ? 3 : a.log_r - s;
        if (rb == 3) ntt_round<3, false, NT>(lds, log_e, a.log_tl, s, a.tw_r);
        else if (rb == 2) ntt_round<2, false, NT>(lds, log_e, a.log_tl, s, a.tw_r);
        else ntt_round<1, false, NT>(lds, log_e, a.log_tl, s, a.tw_r);
        s += rb;
    }
    const int bbits = a.log_n - a.log_s - a.log_r;
    const uint32_t rb = gl::bitrev(b, bbits);
    for (int e = threadIdx.x; e < E; e += NT) {
        int tl = e & (TL - 1), k = e >> a.log_tl;
        uint64_t v = lds[skew(e)];
        if (!a.first && rb && k) {
            uint32_t ex = (uint32_t)((((uint64_t)rb * (uint32_t)k) << a.log_s) & ((1ull << a.log_n) - 1));
            v = mul(v, tw_lookup(a.tw_lo, a.tw_hi, ex, a.tw_h));
        }
        out[base + tl + ((size_t)k << a.log_s)] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Contiguous first pass of the forward transform with the global traffic moved into the rounds: the first round reads its
// operands straight from global memory (zero-padded transforms: the broadcast coefficient), the last round applies the
// pass-boundary twiddle and stores straight to global memory, so a tile crosses LDS once per round boundary only
// (2 exchanges and 2 barriers for 9 butterfly stages instead of 4 and 5).
template <int RB, bool FIRST, bool LAST, int NT>
__device__ __forceinline__ void first_pass_round(const PassArgs& a, uint64_t* lds, const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                 size_t base, size_t cbase, int s, uint32_t rbk) {
    constexpr int G = 1 << RB;
    const int ngroups = 1 << (a.log_r - RB);
    const int tsh = 12 - s - RB;
    for (int g = threadIdx.x; g < ngroups; g += NT) {
        const int b0 = (g & ((1 << s) - 1)) | ((g >> s) << (s + RB));
        const int klow = b0 & ((1 << s) - 1);
        uint64_t y[G];
#pragma unroll
        for (int i = 0; i < G; i++) {
            const int p = b0 + (i << s);
            y[i] = FIRST ? in[cbase + ((size_t)p >> a.log_pad)] : lds[skew(p)];
        }
        if (klow) {
#pragma unroll
            for (int i = 1; i < G; i++) y[i] = mul(y[i], a.tw_r[(uint32_t)(klow * (int)gl::bitrev((uint32_t)i, RB)) << tsh]);
        }
        dft_dit<RB>(y);
#pragma unroll
        for (int i = 0; i < G; i++) {
            const int p = b0 + (i << s);
            if (LAST) {
                uint64_t v = y[i];
                if (!a.first && rbk && p) {
                    const uint32_t ex = (uint32_t)(((uint64_t)rbk * (uint32_t)p) & ((1ull << a.log_n) - 1));
                    v = mul(v, tw_lookup(a.tw_lo, a.tw_hi, ex, a.tw_h));
                }
                out[base + p] = v;
            } else {
                lds[skew(p)] = y[i];
            }
        }
    }
}
template <bool FIRST, bool LAST, int NT>
__device__ __forceinline__ void first_pass_round_any(int rb, const PassArgs& a, uint64_t* lds, const uint64_t* in, uint64_t* out, size_t base,
                                                     size_t cbase, int s, uint32_t rbk) {
    if (rb == 3) first_pass_round<3, FIRST, LAST, NT>(a, lds, in, out, base, cbase, s, rbk);
    else if (rb == 2) first_pass_round<2, FIRST, LAST, NT>(a, lds, in, out, base, cbase, s, rbk);
    else first_pass_round<1, FIRST, LAST, NT>(a, lds, in, out, base, cbase, s, rbk);
}
template <int NT> __global__ __launch_bounds__(NT) void ntt_fwd_first_pass(PassArgs a) {
    __shared__ __attribute__((aligned(16))) uint64_t lds[LDS_ELEMS];
    const uint32_t b = xcd_tile(blockIdx.x, gridDim.x);
    const uint64_t* in = a.in + (size_t)blockIdx.y * a.in_col_stride;
    uint64_t* out = a.out + (size_t)blockIdx.y * a.out_col_stride;
    const size_t base = (size_t)b << a.log_r, cbase = base >> a.log_pad;
    const uint32_t rbk = gl::bitrev(b, a.log_n - a.log_r);
    const int bits = a.log_r - a.log_pad, nrounds = (bits + 2) / 3;      // >= 1 (checked by the launcher)
    int s = a.log_pad;
    for (int rd = 0; rd < nrounds; rd++) {
        const int rb = a.log_r - s >= 3 ? 3 : a.log_r - s;
        const bool first = rd == 0, last = rd + 1 == nrounds;
        if (first && last) first_pass_round_any<true, true, NT>(rb, a, lds, in, out, base, cbase, s, rbk);
        else if (first) first_pass_round_any<true, false, NT>(rb, a, lds, in, out, base, cbase, s, rbk);
        else if (last) first_pass_round_any<false, true, NT>(rb, a, lds, in, out, base, cbase, s, rbk);
        else first_pass_round_any<false, false, NT>(rb, a, lds, in, out, base, cbase, s, rbk);
        if (!last) __syncthreads();
        s += rb;
    }
}

// ------------------------------------------------------------------------------------------------
// Contiguous first pass of the blowup-8 LDE (log_r = 11, log_pad = 3: 8 butterfly stages on 2048-point tiles) as TWO register
// transforms around ONE exchange, one wavefront per tile, 32 values per lane:
//   A  bits 3..5: the three skipped stages broadcast 8 coefficients over the positions [64h, 64h + 64) of the tile. Lane
//      (h, half) loads them and computes, for the 4 values klow = 4 half + klow' of (position mod 8), the twiddled 8-point
//      transform. Every twiddle is a power of w_64 = 2^39, i.e. a compile-time shift (dft_small.hpp): no multiplication, no
//      table. The two halves run the same code: the factor w_64^(4 rev(i)) that separates klow from klow + 4 is applied to
//      the inputs of the upper half with a select.
//   -- the 32 x 64 transpose through LDS (XOR-swizzled columns: conflict-free writes and reads, no padding) --
//   B  bits 6..10: lane k holds positions k + 64 i. One table multiplication per element (w_2048^(k rev(i)), read from a
//      [rev(i)][k] table so that a wavefront's load is one contiguous 512-byte segment), then the 32-point transform whose
//      internal twiddles are shifts again; the pass-boundary twiddle w_N^(rev(tile) p) runs as a geometric progression in i
//      (one lookup per lane + one per tile instead of one lookup per element); stores are contiguous 512-byte segments.
// Against the LDS-round formulation (ntt_fwd_first_pass) this removes a third of the full multiplications, one of two LDS
// exchanges and both workgroup barriers (the waves of a workgroup never synchronise with each other). A first attempt with
// 4096-point tiles and 64 values per lane left one wavefront per SIMD (285 VGPRs, 33 KiB of LDS per wave) and ran 1.8x SLOWER
// than the LDS rounds: the carry chains of the field arithmetic need a second wave to hide their dependent-issue latency.
template <int E> __device__ __forceinline__ uint64_t mul_w64(uint64_t x) {      // x * w_64^E
    constexpr int K = (39 * E) % 192;
    if constexpr (K == 0) return x;
    else if constexpr (K < 96) return mul_pow2<K>(x);
    else if constexpr (K == 96) return gl::neg(x);
    else return gl::neg(mul_pow2<K - 96>(x));
}
template <int KLOW> __device__ __forceinline__ void first8_group(const uint64_t (&c)[8], uint64_t (&v)[32]) {
    // members in position order i carry the sub-transform values in bit-reversed order: twiddle T^rev3(i), T = w_64^KLOW
    uint64_t y[8] = {c[0], mul_w64<KLOW * 4>(c[1]), mul_w64<KLOW * 2>(c[2]), mul_w64<KLOW * 6>(c[3]),
                     mul_w64<KLOW * 1>(c[4]), mul_w64<KLOW * 5>(c[5]), mul_w64<KLOW * 3>(c[6]), mul_w64<KLOW * 7>(c[7])};
    dft_dit<3>(y);
#pragma unroll
    for (int i = 0; i < 8; i++) v[KLOW + 4 * i] = y[i];
}
constexpr int F8_TILE_LDS = 32 * 64;             // elements per tile: element (row, col) of the 32 x 64 exchange lives at
                                                 // row * 64 + (col ^ (2 row & 63)) - conflict-free for the row-wise writes of
                                                 // phase A and the column-wise reads of phase B without any padding
constexpr int F8_WAVES = 4;                      // tiles per workgroup
// The exchange is made in TWO rounds of 16 rows (8 KiB of LDS per wave): with the whole 16 KiB tile per wave the LDS capped the
// kernel at two waves per SIMD (4 waves x 16 KiB = 64 KiB per workgroup, two workgroups per CU); now the 158 VGPRs allow three,
// which the carry chains of the field arithmetic need to hide their dependent-issue latency (107 -> 96 us per 2 columns 2^20 -> 2^23;
// a fourth wave per SIMD costs spills and gains nothing). The 31 table twiddles of phase B are read where they are used (they
// depend on the lane and the row only). Dropped after measuring: a per-ELEMENT table of all pass-boundary twiddles (64 MiB, one
// multiplication instead of the progression's two: 362 us against 252 us per proof), walking 4 tiles per wave (6 % slower), and
// replacing the progression's serial chain by a per-tile table of the 32 step powers read through scalar loads (2 - 5 % slower).
// One column of one 2048-point tile by one wavefront. TW selects the pass-boundary twiddle (w_N^(rev(tile) (lane + 64 i)) for row i):
//   0 = none (the transform's only pass, or tile 0);
//   1 = the progression cur <- cur * step (two multiplications per element), as one or two interleaved chains;
//   2 = the progression, and the factors of the rows i = `wave` (mod F8_WAVES) are ALSO stored to `tab` (32 x 64 words of LDS);
//   3 = one multiplication per element, the factor read from `tab`.
template <int TW>
__device__ __forceinline__ void first8_column(const PassArgs& a, const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t* lds,
                                              int lane, int wave, uint32_t b, uint32_t rbk, uint64_t* tab, int z = 0) {
    const uint64_t* __restrict__ tw_mt = a.tw_mt;
    const int h = lane & 31;
    const bool upper = lane >= 32;
    const uint64_t nmask = ((uint64_t)1 << a.log_n) - 1;
    const size_t base = (size_t)b << 11;
    uint64_t y[32];
    {
        uint64_t c[8];
        const ulonglong2* cp = reinterpret_cast<const ulonglong2*>(in + ((size_t)b << 8) + 8 * h);
#pragma unroll
        for (int q = 0; q < 4; q++) { const ulonglong2 t = cp[q]; c[2 * q] = t.x; c[2 * q + 1] = t.y; }
        { uint64_t t;
          t = mul_w64<16>(c[1]); c[1] = upper ? t : c[1];   t = mul_w64<8>(c[2]);  c[2] = upper ? t : c[2];
          t = mul_w64<24>(c[3]); c[3] = upper ? t : c[3];   t = mul_w64<4>(c[4]);  c[4] = upper ? t : c[4];
          t = mul_w64<20>(c[5]); c[5] = upper ? t : c[5];   t = mul_w64<12>(c[6]); c[6] = upper ? t : c[6];
          t = mul_w64<28>(c[7]); c[7] = upper ? t : c[7]; }
        uint64_t v[32];
        first8_group<0>(c, v); first8_group<1>(c, v); first8_group<2>(c, v); first8_group<3>(c, v);
        uint64_t* row = lds + (h & 15) * 64;
        const int sw = (2 * h) & 63, c0 = upper ? 4 : 0;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if ((h >> 4) == r) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
#pragma unroll
                    for (int k = 0; k < 4; k++) row[(c0 + k + 8 * i) ^ sw] = v[k + 4 * i];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int i = 0; i < 16; i++) y[16 * r + i] = lds[i * 64 + (lane ^ ((2 * (16 * r + i)) & 63))];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
#pragma unroll
    for (int i = 1; i < 32; i++) y[i] = mul(y[i], tw_mt[(int)gl::bitrev((uint32_t)i, 5) * 64 + lane + z]);
    dft_dit_reg<5>(y);
    if constexpr (TW == 0) {
#pragma unroll
        for (int i = 0; i < 32; i++) out[base + lane + 64 * i] = y[i];
    } else if constexpr (TW == 3) {
#pragma unroll
        for (int i = 0; i < 32; i++) out[base + lane + 64 * i] = mul(y[i], tab[i * 64 + lane + z]);
    } else {
        uint64_t cur = tw_lookup(a.tw_lo, a.tw_hi, (uint32_t)(((uint64_t)rbk * (uint32_t)lane) & nmask), a.tw_h);
        const uint64_t step = tw_lookup(a.tw_lo, a.tw_hi, (uint32_t)(((uint64_t)rbk * 64u) & nmask), a.tw_h);
        if (TW == 1 && a.chains == 2) {
            // two interleaved progressions (even and odd rows): the serial chain cur <- cur * step is half as deep
            uint64_t cur1 = mul(cur, step);
            const uint64_t step2 = mul(step, step);
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                out[base + lane + 64 * i] = mul(y[i], cur);
                out[base + lane + 64 * (i + 1)] = mul(y[i + 1], cur1);
                cur = mul(cur, step2); cur1 = mul(cur1, step2);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 32; i++) {
                if (TW == 2 && (i & (F8_WAVES - 1)) == wave) tab[i * 64 + lane] = cur;       // wave-uniform
                out[base + lane + 64 * i] = mul(y[i], cur);
                cur = mul(cur, step);
            }
        }
    }
}
__global__ __launch_bounds__(64 * F8_WAVES, 3) void ntt_fwd_first_pass_8(PassArgs a) {
    __shared__ __attribute__((aligned(16))) uint64_t f8_lds[F8_WAVES * F8_TILE_LDS / 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t* lds = f8_lds + wave * (F8_TILE_LDS / 2);
    const uint32_t b = xcd_tile(blockIdx.x, gridDim.x) * F8_WAVES + wave;
    const uint64_t* in = a.in + (size_t)blockIdx.y * a.in_col_stride;
    uint64_t* out = a.out + (size_t)blockIdx.y * a.out_col_stride;
    const uint32_t rbk = gl::bitrev(b, a.log_n - 11);
    if (!a.first && rbk) first8_column<1>(a, in, out, lds, lane, wave, b, rbk, nullptr);
    else first8_column<0>(a, in, out, lds, lane, wave, b, rbk, nullptr);
}
// Wide launches (round 6): the pass-boundary factors depend on the tile, the lane and the row - NOT on the column. A workgroup takes ONE tile
// and F8_WAVES * K columns of it (wave w: columns w, w + F8_WAVES, ...): each wave's first column runs the progression and leaves a quarter
// of the 32 x 64 factors in LDS (16 KiB per workgroup, next to the four 8 KiB exchange buffers: three workgroups per CU as before), one
// workgroup barrier, and the remaining K - 1 columns of every wave spend ONE multiplication per element on the boundary instead of two
// (190 -> 157 VALU instructions per element at K = 9 by SQ_INSTS_VALU, profiles/r6_ntt_f8w_counters.txt). a.chains = columns of the launch here; grid = (tiles, columns / (F8_WAVES K)).
__global__ __launch_bounds__(64 * F8_WAVES, 3) void ntt_fwd_first_pass_8w(PassArgs a, int K) {
    __shared__ __attribute__((aligned(16))) uint64_t f8_lds[F8_WAVES * F8_TILE_LDS / 2];
    __shared__ __attribute__((aligned(16))) uint64_t f8_tab[F8_TILE_LDS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // scalar: the column pointers stay in SGPRs
    uint64_t* lds = f8_lds + wave * (F8_TILE_LDS / 2);
    const uint32_t b = xcd_tile(blockIdx.x, gridDim.x);
    const uint32_t rbk = gl::bitrev(b, a.log_n - 11);
    const int col0 = (int)blockIdx.y * F8_WAVES * K + wave;
    const uint64_t* in = a.in + (size_t)col0 * a.in_col_stride;
    uint64_t* out = a.out + (size_t)col0 * a.out_col_stride;
    // the table twiddles and the boundary factors do not change from column to column: left to itself the compiler hoists all 31 + 32 loads
    // out of the column loop and spills a hundred registers. An index offset the compiler cannot see through (zero, re-laundered per
    // iteration) keeps every column's reads where they are used (L1 / LDS hits) WITHOUT touching the pointers: laundering the pointers
    // themselves loses their address space - every access became a flat load behind its own s_waitcnt vmcnt(0), +58 % on 72 columns.
    int z = 0;
    if (a.first || !rbk) {       // uniform over the workgroup
        for (int j = 0; j < K; j++) {
            asm volatile("" : "+s"(z));
            first8_column<0>(a, in + (size_t)j * F8_WAVES * a.in_col_stride, out + (size_t)j * F8_WAVES * a.out_col_stride, lds, lane, wave, b, rbk, nullptr, z);
        }
        return;
    }
    first8_column<2>(a, in, out, lds, lane, wave, b, rbk, f8_tab);
    __syncthreads();
    for (int j = 1; j < K; j++) {
        asm volatile("" : "+s"(z));
        first8_column<3>(a, in + (size_t)j * F8_WAVES * a.in_col_stride, out + (size_t)j * F8_WAVES * a.out_col_stride, lds, lane, wave, b, rbk, f8_tab, z);
    }
}
// Round 5 measured this pass with TWO wavefronts per tile and 16 values per lane (74 VGPRs, five waves per SIMD instead of three, commit
// d4aaeb6): 7.6 % more VALU instructions per tile (the phase-A pre-multiplications are shared by half as many values) and 8.5 % more time
// on 2 and on 72 columns - both kernels retire one VALU instruction per 3.5 cycles and SIMD whatever their occupancy: the pass is bound by
// VALU issue (40 % of its instructions are 8-byte VOP3 encodings), not by latency (profiles/r5_ntt_first_pass.md).
// tab[r * 64 + k] = root^(r * k), r < rows
__global__ void fill_mul_table64(uint64_t* tab, uint32_t rows, uint64_t root) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 64) return;
    tab[i] = gl::pow(root, (uint64_t)((i >> 6) * (i & 63)));
}

// Inverse: exact mirror (decimation in frequency with inverse roots), natural input -> bit-reversed output.
template <int NT> __global__ __launch_bounds__(NT) void ntt_inv_pass(PassArgs a) {
    __shared__ __attribute__((aligned(16))) uint64_t lds[LDS_ELEMS];
    const int TL = 1 << a.log_tl, log_e = a.log_r + a.log_tl, E = 1 << log_e;
    const uint32_t tile = xcd_tile(blockIdx.x, gridDim.x);
    const uint32_t tiles_per_b = 1u << (a.log_s - a.log_tl);
    const uint32_t b = tile >> (a.log_s - a.log_tl);
    const uint32_t lo0 = (tile & (tiles_per_b - 1)) << a.log_tl;
    const uint64_t* in = a.in + (size_t)blockIdx.y * a.in_col_stride;
    uint64_t* out = a.out + (size_t)blockIdx.y * a.out_col_stride;
    const size_t base = (size_t)lo0 + (((size_t)b << a.log_r) << a.log_s);
    const int bbits = a.log_n - a.log_s - a.log_r;
    const uint32_t rb = gl::bitrev(b, bbits);

    for (int e = threadIdx.x; e < E; e += NT) {
        int tl = e & (TL - 1), k = e >> a.log_tl;
        uint64_t v = in[base + tl + ((size_t)k << a.log_s)];
        if (a.bad && v >= gl::P) atomicOr(a.bad, 1u);
        if (!a.first && rb && k) {
            uint32_t ex = (uint32_t)((((uint64_t)rb * (uint32_t)k) << a.log_s) & ((1ull << a.log_n) - 1));
            v = mul(v, tw_lookup(a.tw_lo, a.tw_hi, ex, a.tw_h));
        }
        lds[skew(e)] = v;
    }
    __syncthreads();
    // mirror of the forward rounds: same bit groups, highest first
    const int nfull = a.log_r / 3, rem = a.log_r % 3;     // forward rounds: nfull x 3 bits from bit 0, then `rem` bits
    if (rem == 2) ntt_round<2, true, NT>(lds, log_e, a.log_tl, 3 * nfull, a.tw_r);
    else if (rem == 1) ntt_round<1, true, NT>(lds, log_e, a.log_tl, 3 * nfull, a.tw_r);
    for (int q = nfull - 1; q >= 0; q--) ntt_round<3, true, NT>(lds, log_e, a.log_tl, 3 * q, a.tw_r);

    uint64_t bf = 1;
    if (a.ktab) {
        // every thread derives the (uniform) block factor itself: <= 2*bbits multiplies, cheaper than a broadcast
        bf = mul(gl::pow(a.sc_a, rb), gl::pow(a.sc_b, rb >> a.sc_shift));
    }
    for (int e = threadIdx.x; e < E; e += NT) {
        int tl = e & (TL - 1), k = e >> a.log_tl;
        uint64_t v = lds[skew(e)];
        if (a.ktab) v = mul(v, mul(a.ktab[k], bf));
        out[base + tl + ((size_t)k << a.log_s)] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Contiguous LAST pass of the inverse transform (log_r = 11, 2048-point tiles), decimation in frequency with the inverse roots: TWO
// wavefronts per tile, 16 values per lane. The tile as 16 x 16 x 8, position p = 128 j + 8 s + u, three register transforms around two
// exchanges through LDS and no cross-lane step:
//   1  thread t < 128 holds p = t + 128 j: boundary twiddle w^-(rev(tile) p) as a geometric progression in j (its start carries the tile's
//      share of the output scaling, uniform over the tile), 16-point inverse transform over j (-> f2 = rev4(j')), times
//      w_2048^-(f2 t) = table[f2][t mod 64] * (w_32^-f2, a shift, in the wavefront that holds t >= 64);
//   2  thread (j', u) holds t = u + 8 s: 16-point inverse transform over s (-> g1 = rev4(s')), times w_128^-(g1 u) = table[2 g1][8 u];
//   3  thread (j', m) holds rows s' = 2 m, 2 m + 1 with their eight u each: two 8-point inverse transforms (-> g0 = rev3(u')), then the
//      per-position scale table; frequency f2 + 16 g1 + 256 g0 lands at position 128 j' + 8 s' + u' = 16 thread + 8 (s' & 1) + u'.
// Exchange buffer: row j' at j' * 144 slots (the 16 extra slots put consecutive rows into opposite bank halves); first exchange (j', t) at
// ((t + 8 j') mod 128) of the row, second exchange in output order with the 16-byte pairs of a thread's 128 bytes XOR-permuted.
// Until round 5 this pass mirrored ntt_fwd_first_pass_8 (one wavefront per tile, 32 values per lane, a 32-point and a 64-point transform, the
// latter shared by two lanes through V_PERMLANE32_SWAP rounds: 168 VGPRs with 20 spilled, three waves per SIMD). This form has one full
// multiplication per element more (5 against 4) and measured faster on every launch (profiles/r5_inv_last11.md): 41.2 -> 33.2 us on one
// column of 2^21 points (1024 tiles: the one-wave form left one wavefront per SIMD), 706 -> 596 us on 72 columns of 2^20.
constexpr int I2_ROW = 144;
constexpr int rev4c(int j) { return ((j & 1) << 3) | ((j & 2) << 1) | ((j & 4) >> 1) | ((j & 8) >> 3); }
template <int E> __device__ __forceinline__ uint64_t mul_w64_inv(uint64_t x) {      // x * w_64^-E
    constexpr int K = (192 - (39 * E) % 192) % 192;
    if constexpr (K < 96) return mul_pow2<K>(x);
    else return gl::neg(mul_pow2<K - 96>(x));
}
template <int J> __device__ __forceinline__ void upper_wave_factors(uint64_t (&y)[16]) {   // y[j'] *= w_32^-rev4(j')
    y[J] = mul_w64_inv<2 * rev4c(J)>(y[J]);
    if constexpr (J + 1 < 16) upper_wave_factors<J + 1>(y);
}
__global__ __launch_bounds__(128) void ntt_inv_last_pass_11(PassArgs a, const uint64_t* __restrict__ bftab) {
    __shared__ __attribute__((aligned(16))) uint64_t lds[16 * I2_ROW];
    const int tid = threadIdx.x;
    const uint32_t b = xcd_tile(blockIdx.x, gridDim.x);
    uint64_t* data = a.out + (size_t)blockIdx.y * a.out_col_stride + ((size_t)b << 11);
    const uint64_t* src = a.in + (size_t)blockIdx.y * a.in_col_stride + ((size_t)b << 11);
    const uint64_t nmask = ((uint64_t)1 << a.log_n) - 1;
    const uint32_t rbk = gl::bitrev(b, a.log_n - 11);
    uint64_t y[16];
#pragma unroll
    for (int j = 0; j < 16; j++) y[j] = src[tid + 128 * j];
    if (a.bad) {
        bool any = false;
#pragma unroll
        for (int j = 0; j < 16; j++) any |= y[j] >= gl::P;
        if (any) atomicOr(a.bad, 1u);
    }
    {
        uint64_t cur = bftab[b];
        if (!a.first && rbk) {
            cur = mul(cur, tw_lookup(a.tw_lo, a.tw_hi, (uint32_t)(((uint64_t)rbk * (uint32_t)tid) & nmask), a.tw_h));
            const uint64_t step = tw_lookup(a.tw_lo, a.tw_hi, (uint32_t)(((uint64_t)rbk * 128u) & nmask), a.tw_h);
#pragma unroll
            for (int j = 0; j < 16; j++) { y[j] = mul(y[j], cur); if (j < 15) cur = mul(cur, step); }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) y[j] = mul(y[j], cur);
        }
    }
    dft_dif_inv_reg<4>(y);
    {
        const uint64_t* mt = a.tw_mt + (tid & 63);
#pragma unroll
        for (int j = 1; j < 16; j++) y[j] = mul(y[j], mt[rev4c(j) * 64]);
        if (tid >= 64) upper_wave_factors<1>(y);              // uniform over the wavefront
    }
#pragma unroll
    for (int j = 0; j < 16; j++) lds[j * I2_ROW + ((tid + 8 * j) & 127)] = y[j];
    __syncthreads();
    const int jr = tid >> 3, u = tid & 7;
    {
        const uint64_t* row = lds + jr * I2_ROW;
#pragma unroll
        for (int s = 0; s < 16; s++) y[s] = row[(u + 8 * s + 8 * jr) & 127];
    }
    __syncthreads();                                          // the buffer is written again below
    dft_dif_inv_reg<4>(y);
    {
        const uint64_t* mt = a.tw_mt + 8 * u;
#pragma unroll
        for (int s = 1; s < 16; s++) y[s] = mul(y[s], mt[2 * rev4c(s) * 64]);
    }
    {
        // element (s', u) of row j' is element e = 8 (s' & 1) + u of thread (j', s' >> 1) of phase 3; its 16-byte pair e >> 1 sits at
        // pair (e >> 1) ^ sigma, sigma = (s' >> 1) ^ (bit 1 of j') << 2
        uint64_t* row = lds + jr * I2_ROW + (u & 1);
        const int jb = ((jr >> 1) & 1) << 2;
#pragma unroll
        for (int s = 0; s < 16; s++) row[(s >> 1) * 16 + 2 * ((((s & 1) << 2) + (u >> 1)) ^ (s >> 1) ^ jb)] = y[s];
    }
    __syncthreads();
    uint64_t p0[8], p1[8];
    {
        const int m = tid & 7, sigma = m ^ (((jr >> 1) & 1) << 2);
        const ulonglong2* rp = reinterpret_cast<const ulonglong2*>(lds + jr * I2_ROW + m * 16);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const ulonglong2 t0 = rp[q ^ sigma], t1 = rp[(4 + q) ^ sigma];
            p0[2 * q] = t0.x; p0[2 * q + 1] = t0.y; p1[2 * q] = t1.x; p1[2 * q + 1] = t1.y;
        }
    }
    dft_dif_inv_reg<3>(p0);
    dft_dif_inv_reg<3>(p1);
    const ulonglong2* kt = reinterpret_cast<const ulonglong2*>(a.ktab + 16 * tid);
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(data + 16 * tid);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const ulonglong2 k0 = kt[q], k1 = kt[4 + q];
        ulonglong2 o0, o1;
        o0.x = mul(p0[2 * q], k0.x); o0.y = mul(p0[2 * q + 1], k0.y);
        o1.x = mul(p1[2 * q], k1.x); o1.y = mul(p1[2 * q + 1], k1.y);
        dst[q] = o0; dst[4 + q] = o1;
    }
}
// The 12-bit contiguous last pass (4096-point tiles: every transform of 2^13 points and more whose launch is too small for the 11-bit
// plan) in the same form: 16 x 16 x 16, position p = 256 j + 16 s + u, four wavefronts per tile and 16 values per lane, three 16-point
// register transforms, the twiddles between them from one [16][256] table mt[f][t] = w_4096^-(f t) (w_256^-(g u) = mt[g][16 u]).
// Exchange buffer: row j' at j' * 272 slots; first exchange (j', t) at slot t, second in output order (thread (j', s') owns 16
// consecutive slots) with the 16-byte pairs XOR-permuted by s'. Against the LDS rounds of ntt_inv_pass (four rounds of three bits, a
// table multiplication per element and round): 5 full multiplications per element instead of 6.6 and two barriers less.
constexpr int I3_ROW = 272;
__global__ __launch_bounds__(256) void ntt_inv_last_pass_12(PassArgs a, const uint64_t* __restrict__ bftab, const uint64_t* __restrict__ mt) {
    __shared__ __attribute__((aligned(16))) uint64_t lds[16 * I3_ROW];
    const int tid = threadIdx.x;
    const uint32_t b = xcd_tile(blockIdx.x, gridDim.x);
    uint64_t* data = a.out + (size_t)blockIdx.y * a.out_col_stride + ((size_t)b << 12);
    const uint64_t* src = a.in + (size_t)blockIdx.y * a.in_col_stride + ((size_t)b << 12);
    const uint64_t nmask = ((uint64_t)1 << a.log_n) - 1;
    const uint32_t rbk = gl::bitrev(b, a.log_n - 12);
    uint64_t y[16];
#pragma unroll
    for (int j = 0; j < 16; j++) y[j] = src[tid + 256 * j];
    if (a.bad) {
        bool any = false;
#pragma unroll
        for (int j = 0; j < 16; j++) any |= y[j] >= gl::P;
        if (any) atomicOr(a.bad, 1u);
    }
    {
        uint64_t cur = bftab[b];
        if (!a.first && rbk) {
            cur = mul(cur, tw_lookup(a.tw_lo, a.tw_hi, (uint32_t)(((uint64_t)rbk * (uint32_t)tid) & nmask), a.tw_h));
            const uint64_t step = tw_lookup(a.tw_lo, a.tw_hi, (uint32_t)(((uint64_t)rbk * 256u) & nmask), a.tw_h);
#pragma unroll
            for (int j = 0; j < 16; j++) { y[j] = mul(y[j], cur); if (j < 15) cur = mul(cur, step); }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) y[j] = mul(y[j], cur);
        }
    }
    dft_dif_inv_reg<4>(y);
#pragma unroll
    for (int j = 1; j < 16; j++) y[j] = mul(y[j], mt[rev4c(j) * 256 + tid]);
#pragma unroll
    for (int j = 0; j < 16; j++) lds[j * I3_ROW + tid] = y[j];
    __syncthreads();
    const int jr = tid >> 4, u = tid & 15;
    {
        const uint64_t* row = lds + jr * I3_ROW + u;
#pragma unroll
        for (int s = 0; s < 16; s++) y[s] = row[16 * s];
    }
    __syncthreads();                                          // the buffer is written again below
    dft_dif_inv_reg<4>(y);
#pragma unroll
    for (int s = 1; s < 16; s++) y[s] = mul(y[s], mt[rev4c(s) * 256 + 16 * u]);
    {
        uint64_t* row = lds + jr * I3_ROW + (u & 1);
#pragma unroll
        for (int s = 0; s < 16; s++) row[s * 16 + 2 * ((u >> 1) ^ (s & 7))] = y[s];
    }
    __syncthreads();
    {
        const int sr = tid & 15;
        const ulonglong2* rp = reinterpret_cast<const ulonglong2*>(lds + jr * I3_ROW + sr * 16);
#pragma unroll
        for (int q = 0; q < 8; q++) { const ulonglong2 t = rp[q ^ (sr & 7)]; y[2 * q] = t.x; y[2 * q + 1] = t.y; }
    }
    dft_dif_inv_reg<4>(y);
    const ulonglong2* kt = reinterpret_cast<const ulonglong2*>(a.ktab + 16 * tid);
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(data + 16 * tid);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const ulonglong2 k2 = kt[q];
        ulonglong2 o;
        o.x = mul(y[2 * q], k2.x); o.y = mul(y[2 * q + 1], k2.y);
        dst[q] = o;
    }
}
// tab[r << log_cols | k] = root^(r k), r < rows
__global__ void fill_mul_table(uint64_t* tab, uint32_t rows, int log_cols, uint64_t root) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (rows << log_cols)) return;
    tab[i] = gl::pow(root, (uint64_t)((i >> log_cols) * (i & ((1u << log_cols) - 1))));
}
// tab[b] = a^rev(b) * b2^(rev(b) >> shift), rev over `bits` bits: the tile factors of the inverse transform's output scaling
__global__ void fill_block_factors(uint64_t* tab, int bits, uint64_t abase, uint64_t bbase, int shift) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (1u << bits)) return;
    const uint32_t rb = gl::bitrev(b, bits);
    tab[b] = mul(gl::pow(abase, rb), gl::pow(bbase, rb >> shift));
}

// ------------------------------------------------------------------------------------------------
// Strided passes of radix <= 64 entirely in registers: a thread owns one column position `lo` of a block and the R = 2^LOGR
// values at stride S; the R-point transform needs no LDS because every twiddle of a transform of up to 64 points is a power of
// two in this field (dft_small.hpp). Consecutive threads own consecutive addresses, so each of the R loads / stores of a
// wavefront is one contiguous 512-byte segment. The pass-boundary twiddle depends on (block, row) only: it comes from a small
// per-pass table (2^(log_n - log_s) entries, row b * R + k = w^(S * rev(b) * k)) that every lane of a workgroup reads at the
// same address.
typedef uint32_t nttv2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint64_t buf_ld(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    const nttv2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return gl::mk64(v.x, v.y);
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, uint64_t x) {
    nttv2u v; v.x = (uint32_t)x; v.y = (uint32_t)(x >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}
// descriptor over the 2^LOGR rows of block b of a strided pass (2^(log_s + LOGR + 3) bytes: the launcher checks that it fits 32 bits)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pass_rsrc(const uint64_t* col, uint32_t b, int logr, int log_s) {
    const size_t blk = ((size_t)b << logr) << log_s;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(col + blk), 0, (uint32_t)((((size_t)1 << logr) << log_s) * 8), 0x00020000);
}
template <int LOGR> __global__ __launch_bounds__(256, LOGR == 6 ? 3 : 1) void ntt_fwd_strided_reg(PassArgs a) {
    constexpr int R = 1 << LOGR;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;            // over 2^(log_n - LOGR) threads per column
    const size_t lo = t & (((size_t)1 << a.log_s) - 1);
    const uint32_t b = (uint32_t)(((size_t)blockIdx.x * 256) >> a.log_s);   // S >= 256: uniform over the workgroup (scalar twiddle loads)
    const size_t base = lo + (((size_t)b << LOGR) << a.log_s);
    const uint64_t* in = a.in + (size_t)blockIdx.y * a.in_col_stride;
    uint64_t* out = a.out + (size_t)blockIdx.y * a.out_col_stride;
    uint64_t y[R];
#pragma unroll
    for (int k = 0; k < R; k++) y[k] = in[base + ((size_t)k << a.log_s)];
    if (LOGR == 6) __builtin_amdgcn_sched_barrier(0);   // all 64 loads in flight before the first butterfly (the scheduler otherwise sinks half of them)
    dft_dit_reg<LOGR>(y);
    if (!a.first && b) {
        const uint64_t* tw = a.tw_pass + ((size_t)b << LOGR);
#pragma unroll
        for (int k = 1; k < R; k++) y[k] = mul(y[k], tw[k]);
    }
#pragma unroll
    for (int k = 0; k < R; k++) out[base + ((size_t)k << a.log_s)] = y[k];
    if (a.compact && (lo & (((size_t)1 << a.compact_log) - 1)) == 0) {
        // every 2^compact_log-th row once more, densely: what the per-row kernels that walk the LDE with that stride read
        // (constraint evaluation, DEEP) - a strided walk over the full matrix drags in a whole 64-byte sector per 8 useful bytes
        uint64_t* co = a.compact + (size_t)blockIdx.y * a.compact_col_stride;
        const size_t part_len = (((size_t)1 << a.log_n) >> a.compact_log) >> a.compact_split, pmask = ((size_t)1 << a.compact_split) - 1;
#pragma unroll
        for (int k = 0; k < R; k++) {
            const size_t j = (base + ((size_t)k << a.log_s)) >> a.compact_log;
            co[(j & pmask) * part_len + (j >> a.compact_split)] = y[k];
        }
    }
}
// The radix-64 pass with each 64-point transform shared by TWO lanes (lane l and l + 32 of a wavefront): 32 values per lane, the
// register footprint and occupancy of the radix-32 pass (which streams at the HBM rate where the one-lane radix-64 pass sits at
// 2-3 waves per SIMD and 60-70 % of it). Lane half h holds the rows 32 h + i: the first five stages are the 32-point transform of
// each half (same shift twiddles in both), the last stage pairs row j with row j + 32 across the halves: one V_PERMLANE32_SWAP per
// register pair hands each lane 16 complete (u, v) pairs - rows (i, i + 32) in the lower half, (16 + i, 48 + i) in the upper -
// and the factor w_64^16 = 2^48 that separates the two sets of twiddles is applied to v in the upper half with a select.
template <int I> __device__ __forceinline__ void last_stage_pairs(uint64_t (&y)[32]) {
    bfly_w64<I>(y[I], y[16 + I]);
    if constexpr (I + 1 < 16) last_stage_pairs<I + 1>(y);
}
__global__ __launch_bounds__(256) void ntt_fwd_strided_reg6x2(PassArgs a) {
    const uint32_t lane = threadIdx.x & 63, half = lane >> 5;
    const size_t p = (((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) << 5) + (lane & 31);    // over 2^(log_n - 6) positions per column
    const size_t lo = p & (((size_t)1 << a.log_s) - 1);
    const uint32_t b = (uint32_t)(((size_t)blockIdx.x * 128) >> a.log_s);                  // S >= 128: uniform over the workgroup
    const size_t base = lo + (((size_t)b << 6) << a.log_s);
    const uint64_t* in = a.in + (size_t)blockIdx.y * a.in_col_stride;
    uint64_t* out = a.out + (size_t)blockIdx.y * a.out_col_stride;
    uint64_t y[32];
#pragma unroll
    for (int i = 0; i < 32; i++) y[i] = in[base + ((size_t)(32 * half + i) << a.log_s)];
    dft_dit_reg<5>(y);
#pragma unroll
    for (int i = 0; i < 16; i++) swap_halves(y[i], y[16 + i]);
#pragma unroll
    for (int i = 0; i < 16; i++) { const uint64_t t = mul_w4(y[16 + i]); y[16 + i] = half ? t : y[16 + i]; }
    last_stage_pairs<0>(y);
    // y[i] is row r0 + i, y[16 + i] row r0 + 32 + i, r0 = 16 half
    const uint32_t r0 = 16 * half;
    if (!a.first && b) {
        const uint64_t* tw = a.tw_pass + ((size_t)b << 6) + r0;      // two distinct addresses per wavefront
#pragma unroll
        for (int i = 0; i < 16; i++) { y[i] = mul(y[i], tw[i]); y[16 + i] = mul(y[16 + i], tw[32 + i]); }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        out[base + ((size_t)(r0 + i) << a.log_s)] = y[i];
        out[base + ((size_t)(r0 + 32 + i) << a.log_s)] = y[16 + i];
    }
    if (a.compact && (lo & (((size_t)1 << a.compact_log) - 1)) == 0) {
        uint64_t* co = a.compact + (size_t)blockIdx.y * a.compact_col_stride;
        const size_t part_len = (((size_t)1 << a.log_n) >> a.compact_log) >> a.compact_split, pmask = ((size_t)1 << a.compact_split) - 1;
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const uint32_t row = r0 + (i & 15) + 2 * (i & 16);
            const size_t j = (base + ((size_t)row << a.log_s)) >> a.compact_log;
            co[(j & pmask) * part_len + (j >> a.compact_split)] = y[i];
        }
    }
}
// The same pass with the 64 row accesses of a lane addressed through a buffer descriptor whose base (column + block) lives in SGPRs - the row
// part of every address is a scalar offset (k << log_s, SALU) and the lane part one VGPR computed once, instead of a 64-bit vector shift-add
// per access: 96 VGPRs instead of 135, five waves per SIMD (profiles/r4_ntt_gap.md). Asking for four workgroups per CU by launch bounds alone
// (128 VGPRs, spills) measured slower in both forms and is gone.
__global__ __launch_bounds__(256, 3) void ntt_fwd_strided_reg6x2_buf(PassArgs a) {
    const uint32_t lane = threadIdx.x & 63, half = lane >> 5;
    const size_t p = (((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) << 5) + (lane & 31);    // over 2^(log_n - 6) positions per column
    const size_t lo = p & (((size_t)1 << a.log_s) - 1);
    const uint32_t b = (uint32_t)(((size_t)blockIdx.x * 128) >> a.log_s);                  // S >= 128: uniform over the workgroup
    const size_t base = lo + (((size_t)b << 6) << a.log_s);
    const uint64_t* in = a.in + (size_t)blockIdx.y * a.in_col_stride;
    uint64_t* out = a.out + (size_t)blockIdx.y * a.out_col_stride;
    uint64_t y[32];
    // descriptors over this block's 64 rows (2^(log_s + 9) bytes <= 2^31: the launcher checks)
    const size_t blk = ((size_t)b << 6) << a.log_s;
    const uint32_t span = (uint32_t)(((size_t)64 << a.log_s) * 8);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(in + blk), 0, span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out + blk, 0, span, 0x00020000);
    const uint32_t row_bytes = 8u << a.log_s;
    const uint32_t vi = (uint32_t)lo * 8 + half * 32 * row_bytes;
#pragma unroll
    for (int i = 0; i < 32; i++) y[i] = buf_ld(rin, vi, (uint32_t)i * row_bytes);
    dft_dit_reg<5>(y);
#pragma unroll
    for (int i = 0; i < 16; i++) swap_halves(y[i], y[16 + i]);
#pragma unroll
    for (int i = 0; i < 16; i++) { const uint64_t t = mul_w4(y[16 + i]); y[16 + i] = half ? t : y[16 + i]; }
    last_stage_pairs<0>(y);
    const uint32_t r0 = 16 * half;
    if (!a.first && b) {
        const uint64_t* tw = a.tw_pass + ((size_t)b << 6) + r0;
#pragma unroll
        for (int i = 0; i < 16; i++) { y[i] = mul(y[i], tw[i]); y[16 + i] = mul(y[16 + i], tw[32 + i]); }
    }
    const uint32_t vo = (uint32_t)lo * 8 + half * 16 * row_bytes;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        buf_st(rout, vo, (uint32_t)i * row_bytes, y[i]);
        buf_st(rout, vo, (uint32_t)(32 + i) * row_bytes, y[16 + i]);
    }
    if (a.compact && (lo & (((size_t)1 << a.compact_log) - 1)) == 0) {
        uint64_t* co = a.compact + (size_t)blockIdx.y * a.compact_col_stride;
        const size_t part_len = (((size_t)1 << a.log_n) >> a.compact_log) >> a.compact_split, pmask = ((size_t)1 << a.compact_split) - 1;
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const uint32_t row = r0 + (i & 15) + 2 * (i & 16);
            const size_t j = (base + ((size_t)row << a.log_s)) >> a.compact_log;
            co[(j & pmask) * part_len + (j >> a.compact_split)] = y[i];
        }
    }
}
// Radix-128 pass, each 128-point transform shared by two lanes (l and l + 32), 64 values per lane: lane half h holds rows 64 h + i, the
// first six stages are the 64-point register transform of each half (shift twiddles), the last stage pairs row j with row j + 64
// across the halves (V_PERMLANE32_SWAP: rows (i, 64 + i) end up in the lower half, (32 + i, 96 + i) in the upper) with the
// twiddle w_128^j = w_128^(j mod 32) * (w_4 = 2^48 in the upper half): w_128 is not a power of two (2 has order 192), so this one
// stage multiplies by 31 table constants (w_4096^(32 i), uniform over the wavefront) - half a multiplication per element. Used where
// it saves a whole pass: a 2^25-point LDE (2^22-row traces, BASELINE configs[4]) is 11 + 7 + 7 bits instead of 11 + 5 + 5 + 4.
__global__ __launch_bounds__(256, 3) void ntt_fwd_strided_reg7x2(PassArgs a) {
    const uint32_t lane = threadIdx.x & 63, half = lane >> 5;
    const size_t p = (((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) << 5) + (lane & 31);    // over 2^(log_n - 7) positions per column
    const size_t lo = p & (((size_t)1 << a.log_s) - 1);
    const uint32_t b = (uint32_t)(((size_t)blockIdx.x * 128) >> a.log_s);                  // S >= 128: uniform over the workgroup
    const size_t base = lo + (((size_t)b << 7) << a.log_s);
    const uint64_t* in = a.in + (size_t)blockIdx.y * a.in_col_stride;
    uint64_t* out = a.out + (size_t)blockIdx.y * a.out_col_stride;
    uint64_t y[64];
#pragma unroll
    for (int i = 0; i < 64; i++) y[i] = in[base + ((size_t)(64 * half + i) << a.log_s)];
    __builtin_amdgcn_sched_barrier(0);
    dft_dit_reg<6>(y);
#pragma unroll
    for (int i = 0; i < 32; i++) swap_halves(y[i], y[32 + i]);
    // y[i] = row 32 half + i, y[32 + i] = row 64 + 32 half + i: (u, v) of the last stage, twiddle w_128^(32 half + i). Every pair is
    // finished - butterfly, pass-boundary twiddle, stores - before the next one is touched: its registers are free again at once.
    const uint32_t r0 = 32 * half;
    const bool twiddle = !a.first && b;
    const uint64_t* tw = a.tw_pass + ((size_t)b << 7) + r0;
    const bool compact = a.compact && (lo & (((size_t)1 << a.compact_log) - 1)) == 0;
    uint64_t* co = a.compact + (size_t)blockIdx.y * a.compact_col_stride;
    const size_t part_len = (((size_t)1 << a.log_n) >> a.compact_log) >> a.compact_split, pmask = ((size_t)1 << a.compact_split) - 1;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        uint64_t v = y[32 + i];
        const uint64_t t = mul_w4(v);
        v = half ? t : v;
        if (i) v = mul(v, a.tw_r[32 * i]);
        const uint64_t u = y[i];
        uint64_t lo_row = add(u, v), hi_row = sub(u, v);
        if (twiddle) { lo_row = mul(lo_row, tw[i]); hi_row = mul(hi_row, tw[64 + i]); }
        const size_t p0 = base + ((size_t)(r0 + i) << a.log_s), p1 = base + ((size_t)(r0 + 64 + i) << a.log_s);
        out[p0] = lo_row;
        out[p1] = hi_row;
    
        if (compact) {
            const size_t j0 = p0 >> a.compact_log, j1 = p1 >> a.compact_log;
            co[(j0 & pmask) * part_len + (j0 >> a.compact_split)] = lo_row;
            co[(j1 & pmask) * part_len + (j1 >> a.compact_split)] = hi_row;
        }
    }
}
template <int LOGR> __global__ __launch_bounds__(256, LOGR == 6 ? 3 : 1) void ntt_inv_strided_reg(PassArgs a) {
    constexpr int R = 1 << LOGR;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t lo = t & (((size_t)1 << a.log_s) - 1);
    const uint32_t b = (uint32_t)(((size_t)blockIdx.x * 256) >> a.log_s);   // S >= 256: uniform over the workgroup (scalar twiddle loads)
    const size_t base = lo + (((size_t)b << LOGR) << a.log_s);
    const uint64_t* in = a.in + (size_t)blockIdx.y * a.in_col_stride;
    uint64_t* out = a.out + (size_t)blockIdx.y * a.out_col_stride;
    uint64_t y[R];
#pragma unroll
    for (int k = 0; k < R; k++) y[k] = in[base + ((size_t)k << a.log_s)];
    if (LOGR == 6) __builtin_amdgcn_sched_barrier(0);
    if (a.bad) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < R; k++) any |= y[k] >= gl::P;
        if (any) atomicOr(a.bad, 1u);
    }
    if (!a.first && b) {
        const uint64_t* tw = a.tw_pass + ((size_t)b << LOGR);
#pragma unroll
        for (int k = 1; k < R; k++) y[k] = mul(y[k], tw[k]);
    }
    dft_dif_inv_reg<LOGR>(y);
#pragma unroll
    for (int k = 0; k < R; k++) out[base + ((size_t)k << a.log_s)] = y[k];
}
// tab[(b << log_r) + k] = root^((S * rev(b) * k) mod n), rev over bbits = log_n - log_s - log_r bits
__global__ void fill_pass_twiddles(uint64_t* tab, int log_n, int log_s, int log_r, uint64_t root) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << (log_n - log_s))) return;
    const uint32_t b = i >> log_r, k = i & ((1u << log_r) - 1);
    const uint32_t rb = gl::bitrev(b, log_n - log_s - log_r);
    const uint64_t e = (((uint64_t)rb * k) << log_s) & (((uint64_t)1 << log_n) - 1);
    tab[i] = gl::pow(root, e);
}

// tab[k] = c0 * abase^rev(k) * bbase^(rev(k) >> bshift), rev over `bits` bits, k < 2^bits
__global__ void fill_pow_bitrev(uint64_t* tab, int bits, uint64_t abase, uint64_t bbase, int bshift, uint64_t c0) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (1u << bits)) return;
    uint32_t r = gl::bitrev(k, bits);
    tab[k] = mul(mul(c0, gl::pow(abase, r)), gl::pow(bbase, r >> bshift));
}
// tab[j] = root^(j * step)
__global__ void fill_pow_linear(uint64_t* tab, uint32_t count, uint64_t root, uint64_t step) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    tab[k] = gl::pow(root, (uint64_t)k * step);
}

// ------------------------------------------------------------------------------------------------
// (The buffer-descriptor form was also built for the radix-16 / 32 / 128 register passes and the inverse ones, round 4: their VGPR count
// does not move - 51 to 96, 168 for radix 128 - and nothing measurable changed; only the two-lane radix-64 pass keeps it.)

NttTables* Context::ntt_tables(int log_n) {
    auto it = ntt_tabs.find(log_n);
    if (it != ntt_tabs.end()) return &it->second;
    NttTables t;
    t.log_n = log_n;
    t.h = (log_n + 1) / 2;
    uint32_t nlo = 1u << t.h, nhi = 1u << (log_n - t.h);
    uint64_t w = gl::root_of_unity(log_n), wi = gl::inv(w);
    t.lo_fwd = (uint64_t*)dev_alloc((size_t)nlo * 8);
    t.hi_fwd = (uint64_t*)dev_alloc((size_t)nhi * 8);
    t.lo_inv = (uint64_t*)dev_alloc((size_t)nlo * 8);
    t.hi_inv = (uint64_t*)dev_alloc((size_t)nhi * 8);
    AERO_LAUNCH(this, "fill_pow_linear", 0, fill_pow_linear, dim3((nlo + 255) / 256), dim3(256), 0, t.lo_fwd, nlo, w, 1ull);
    AERO_LAUNCH(this, "fill_pow_linear", 0, fill_pow_linear, dim3((nhi + 255) / 256), dim3(256), 0, t.hi_fwd, nhi, w, (uint64_t)nlo);
    AERO_LAUNCH(this, "fill_pow_linear", 0, fill_pow_linear, dim3((nlo + 255) / 256), dim3(256), 0, t.lo_inv, nlo, wi, 1ull);
    AERO_LAUNCH(this, "fill_pow_linear", 0, fill_pow_linear, dim3((nhi + 255) / 256), dim3(256), 0, t.hi_inv, nhi, wi, (uint64_t)nlo);
    check_launch("ntt tables");
    ntt_tabs[log_n] = t;
    return &ntt_tabs[log_n];
}

const uint64_t* Context::pass_twiddles(int log_n, int log_s, int log_r, bool inverse) {
    const uint64_t key = ((uint64_t)log_n << 24) | ((uint64_t)log_s << 16) | ((uint64_t)log_r << 8) | (inverse ? 1 : 0);
    auto it = pass_tabs.find(key);
    if (it != pass_tabs.end()) return it->second;
    const uint32_t count = 1u << (log_n - log_s);
    uint64_t* tab = (uint64_t*)dev_alloc((size_t)count * 8);
    uint64_t w = gl::root_of_unity(log_n);
    if (inverse) w = gl::inv(w);
    AERO_LAUNCH(this, "fill_pass_twiddles", 0, fill_pass_twiddles, dim3((count + 255) / 256), dim3(256), 0, tab, log_n, log_s, log_r, w);
    check_launch("pass twiddles");
    pass_tabs[key] = tab;
    return tab;
}

void Context::ensure_small_twiddles() {
    if (tw4096_fwd) return;
    tw4096_fwd = (uint64_t*)dev_alloc(4096 * 8);
    tw4096_inv = (uint64_t*)dev_alloc(4096 * 8);
    uint64_t w = gl::root_of_unity(12);
    AERO_LAUNCH(this, "fill_pow_linear", 0, fill_pow_linear, dim3(16), dim3(256), 0, tw4096_fwd, 4096u, w, 1ull);
    AERO_LAUNCH(this, "fill_pow_linear", 0, fill_pow_linear, dim3(16), dim3(256), 0, tw4096_inv, 4096u, gl::inv(w), 1ull);
    // two-phase contiguous pass: [r * 64 + k] = w_2048^(r k), r < 32
    twmt_fwd = (uint64_t*)dev_alloc(2048 * 8);
    twmt_inv = (uint64_t*)dev_alloc(2048 * 8);
    const uint64_t w2048 = gl::mul(w, w);
    AERO_LAUNCH(this, "fill_mul_table64", 0, fill_mul_table64, dim3(8), dim3(256), 0, twmt_fwd, 32u, w2048);
    AERO_LAUNCH(this, "fill_mul_table64", 0, fill_mul_table64, dim3(8), dim3(256), 0, twmt_inv, 32u, gl::inv(w2048));
    // 12-bit contiguous inverse pass: [f * 256 + t] = w_4096^-(f t), f < 16
    twmt12_inv = (uint64_t*)dev_alloc(4096 * 8);
    AERO_LAUNCH(this, "fill_mul_table", 0, fill_mul_table, dim3(16), dim3(256), 0, twmt12_inv, 16u, 8, gl::inv(w));
    check_launch("small twiddles");
}

// LDS-round passes (ntt_fwd_pass, ntt_fwd_first_pass, ntt_inv_pass) with 512 threads per 4096-element tile instead of 256 on launches of up
// to 2^21 elements (columns x points). A small launch is one tile per CU or less - one wave per SIMD, and the pass lasts as long as that wave
// needs for its 16 elements per lane; with 8 per lane and two waves per SIMD the same work takes less: measured (profiles/r4_lds512_ab.txt)
// inverse of 1 column 2^20 (the DEEP stage's) 61.1 -> 51.0 us, of 2 columns 2^19 59.7 -> 49.5 us, interpolation + LDE of 2 columns 2^16
// 95.2 -> 79.0 us. Larger launches fill the SIMDs either way and keep 256.
static bool lds_wide(int ncols, int log_n) { return ((size_t)ncols << log_n) <= ((size_t)1 << 21); }

// Pass plan for a transform of 2^L points: first the contiguous pass (<= 12 bits), then strided passes. With register
// passes every strided pass has radix <= 64 (one more pass over HBM beats a radix-128/256 pass through LDS: measured 2^25
// points, 4 columns: 1162 us for the LDS radix-128 pass against 550 us for a register radix-64 pass); without them the
// strided passes take up to 8 bits each through LDS tiles of R x TL = 4096 elements.
std::vector<NttPass> plan_passes(int L, bool reg, int first_bits, bool radix128, int lds_max_r) {
    std::vector<NttPass> p;
    int r1 = L < first_bits ? L : first_bits;
    p.push_back(NttPass{0, r1, 0});
    int rem = L - r1, s = r1;
    if (rem == 0) return p;
    int max_r = reg ? 6 : lds_max_r;
    // forward register passes: radix 128 (two lanes per transform) where it saves a whole pass over radix <= 64
    if (reg && radix128 && (rem + 6) / 7 < (rem + 5) / 6) max_r = 7;
    int npass = (rem + max_r - 1) / max_r;
    for (int i = 0; i < npass; i++) {
        int r = (rem + (npass - 1 - i)) / (npass - i);   // split as evenly as possible, larger radices first
        p.push_back(NttPass{s, r, 12 - r});               // LDS passes: tile = R x TL = 4096 elements, TL = 4096 / R <= S
        s += r;
        rem -= r;
    }
    return p;
}

// coefficients (bit-reversed, pre-scaled by offset^i), n_in = 2^(log_out - log_pad) per column
//   -> evaluations in natural order over offset * <w_(2^log_out)>
// The two-phase contiguous pass covers 11 bits instead of 12: it is used where that does not cost an extra strided pass.
bool Context::fwd_two_phase(int log_out, int log_pad) const {
    if (!reg_passes || !two_phase || log_pad != 3 || log_out < 13) return false;
    // strided passes behind an 11-bit and behind a 12-bit first pass (radix 128 where it saves a pass, as plan_passes decides)
    auto strided = [&](int rem) { const int r6 = (rem + 5) / 6, r7 = (rem + 6) / 7; return radix128 && r7 < r6 ? r7 : r6; };
    return strided(log_out - 11) == strided(log_out - 12) || log_out - 12 <= 0;
}
bool Context::ntt_forward(const uint64_t* in, size_t in_stride, uint64_t* out, size_t out_stride, int ncols, int log_out, int log_pad,
                          const CompactOut* compact) {
    ensure_small_twiddles();
    bool compact_written = false;
    NttTables* t = ntt_tables(log_out);
    std::vector<NttPass> plan = plan_passes(log_out, reg_passes, fwd_two_phase(log_out, log_pad) ? 11 : 12, radix128, 8);
    if (plan[0].log_r < log_pad) fail("ntt_forward: transform too small for the requested padding");
    for (size_t q = 0; q < plan.size(); q++) {
        PassArgs a{};
        a.in = q == 0 ? in : out;
        a.in_col_stride = q == 0 ? in_stride : out_stride;
        a.out = out;
        a.out_col_stride = out_stride;
        a.log_n = log_out; a.log_s = plan[q].log_s; a.log_r = plan[q].log_r; a.log_tl = plan[q].log_tl;
        a.log_pad = q == 0 ? log_pad : 0;
        a.first = (q + 1 == plan.size());
        a.tw_r = tw4096_fwd; a.tw_lo = t->lo_fwd; a.tw_hi = t->hi_fwd; a.tw_h = t->h;
        const size_t abytes = (size_t)ncols * 8 * ((((size_t)1 << log_out) >> a.log_pad) + ((size_t)1 << log_out));
        if (q > 0 && reg_passes) {
            if (!a.first) a.tw_pass = pass_twiddles(log_out, a.log_s, a.log_r, false);
            if (a.first && compact && compact->ptr && compact->log_step >= 1 && compact->log_step <= a.log_s) {
                a.compact = compact->ptr; a.compact_col_stride = compact->col_stride; a.compact_log = compact->log_step; a.compact_split = compact->log_split;
                compact_written = true;
            }
            dim3 rgrid((unsigned)((((size_t)1 << log_out) >> a.log_r) / 256), ncols);
            const char* nm = pass_names ? (a.log_r == 6 ? (a.first ? "ntt_fwd_reg6_last" : "ntt_fwd_reg6_mid") : a.log_r == 5 ? "ntt_fwd_reg5" : a.log_r == 4 ? "ntt_fwd_reg4" : "ntt_fwd_reg123") : "ntt_fwd_pass";
            switch (a.log_r) {
                case 7:
                    AERO_LAUNCH(this, pass_names ? "ntt_fwd_reg7" : nm, abytes, ntt_fwd_strided_reg7x2, dim3((unsigned)((((size_t)1 << log_out) >> 6) / 256), ncols), dim3(256), 0, a);
                    break;
                case 6:
                    if (a.log_s >= 7) {  // two lanes per transform: 32 values per lane (the block index stays uniform over a workgroup)
                        // Rows addressed through a buffer descriptor (96 VGPRs instead of 135: the 32 row addresses no longer live in VGPR pairs;
                        // measured on MI355X, profiles/r4_ntt_ab.txt: middle pass 65.5 -> 61.0 us on 2 columns 2^23, 1837 -> 1741 us on 72); the
                        // LAST pass gains on narrow launches (49.2 -> 46.0 us) and loses on wide ones (1824 -> 1938 us, five waves per SIMD
                        // streaming 64 rows each plus the compact copy), so it keeps the pointer form from 16 columns on.
                        const dim3 g6((unsigned)((((size_t)1 << log_out) >> 5) / 256), ncols);
                        const bool buf_ok = a.log_s + 9 <= 31;
                        const bool want_buf = !a.first || ncols < 16;
                        if (want_buf && buf_ok) AERO_LAUNCH(this, pass_names ? (a.first ? "ntt_fwd_reg6_last_buf" : "ntt_fwd_reg6_mid_buf") : nm, abytes, ntt_fwd_strided_reg6x2_buf, g6, dim3(256), 0, a);
                        else AERO_LAUNCH(this, nm, abytes, ntt_fwd_strided_reg6x2, g6, dim3(256), 0, a);
                    }
                    else AERO_LAUNCH(this, pass_names ? "ntt_fwd_reg6_1lane" : nm, abytes, ntt_fwd_strided_reg<6>, rgrid, dim3(256), 0, a);
                    break;
                case 5: AERO_LAUNCH(this, nm, abytes, ntt_fwd_strided_reg<5>, rgrid, dim3(256), 0, a); break;
                case 4: AERO_LAUNCH(this, nm, abytes, ntt_fwd_strided_reg<4>, rgrid, dim3(256), 0, a); break;
                case 3: AERO_LAUNCH(this, nm, abytes, ntt_fwd_strided_reg<3>, rgrid, dim3(256), 0, a); break;
                case 2: AERO_LAUNCH(this, nm, abytes, ntt_fwd_strided_reg<2>, rgrid, dim3(256), 0, a); break;
                default: AERO_LAUNCH(this, nm, abytes, ntt_fwd_strided_reg<1>, rgrid, dim3(256), 0, a); break;
            }
            continue;
        }
        size_t E = (size_t)1 << (a.log_r + a.log_tl);
        dim3 grid((unsigned)(((size_t)1 << log_out) / E), ncols);
        if (q == 0 && a.log_r == 11 && a.log_pad == 3 && fwd_two_phase(log_out, log_pad)) {
            a.tw_mt = twmt_fwd;
            // pass-boundary progression as two interleaved chains on narrow launches (2 columns 2^20 -> 2^23: 94.3 -> 91.4 us; 72 columns: 2950 ->
            // 2977 us, so wide launches keep the single chain; four chains gained nothing: profiles/r4_ntt_ab.txt)
            a.chains = ncols < 16 ? 2 : 1;
            const size_t tiles = ((size_t)1 << log_out) >> 11;
            const char* nm = pass_names ? "ntt_fwd_first8" : "ntt_fwd_pass";
            // from 16 columns on, and when the columns split evenly: one tile and F8_WAVES * K columns per workgroup, the boundary factors
            // generated once per tile (ntt_fwd_first_pass_8w); K = the largest divisor of columns / F8_WAVES up to 9. AERO_NTT_F8W=0: never.
            static const bool f8w = !(getenv("AERO_NTT_F8W") && getenv("AERO_NTT_F8W")[0] == '0');
            int K = 0;
            if (f8w && ncols >= 16 && ncols % F8_WAVES == 0)
                for (int k = 9; k >= 2; k--) if ((ncols / F8_WAVES) % k == 0) { K = k; break; }
            if (K) AERO_LAUNCH(this, pass_names ? "ntt_fwd_first8w" : nm, abytes, ntt_fwd_first_pass_8w, dim3((unsigned)tiles, (unsigned)(ncols / (F8_WAVES * K))), dim3(64 * F8_WAVES), 0, a, K);
            else AERO_LAUNCH(this, nm, abytes, ntt_fwd_first_pass_8, dim3((unsigned)(tiles / F8_WAVES), ncols), dim3(64 * F8_WAVES), 0, a);
            continue;
        }
        if (q == 0 && reg_passes && a.log_r > a.log_pad) {
            if (lds_wide(ncols, log_out)) AERO_LAUNCH(this, pass_names ? "ntt_fwd_first_512" : "ntt_fwd_pass", abytes, ntt_fwd_first_pass<512>, grid, dim3(512), 0, a);
            else AERO_LAUNCH(this, pass_names ? "ntt_fwd_first_256" : "ntt_fwd_pass", abytes, ntt_fwd_first_pass<256>, grid, dim3(256), 0, a);
            continue;
        }
        if (lds_wide(ncols, log_out)) AERO_LAUNCH(this, pass_names ? "ntt_fwd_lds_512" : "ntt_fwd_pass", abytes, ntt_fwd_pass<512>, grid, dim3(512), 0, a);
        else AERO_LAUNCH(this, pass_names ? "ntt_fwd_lds_256" : "ntt_fwd_pass", abytes, ntt_fwd_pass<256>, grid, dim3(256), 0, a);
    }
    check_launch("ntt_forward");
    return compact_written;
}

// evaluations in natural order over <w_n> (n = 2^log_n) -> coefficients in bit-reversed order, where coefficient
// with natural index I is multiplied by c0 * sa^I * sb^(I >> shift). In place, or from `src` into `data` (every pass reads a.in and writes
// a.out: the first executed one takes its values from `src`, the others find them in `data`).
void Context::ntt_inverse(uint64_t* data, size_t stride, int ncols, int log_n, uint64_t c0, uint64_t sa, uint64_t sb, int shift, unsigned int* bad,
                          const uint64_t* src, size_t src_stride) {
    ensure_small_twiddles();
    NttTables* t = ntt_tables(log_n);
    // Small launches (lds_wide: up to 2^21 elements) of 2^18 .. 2^21 points: the 6 - 9 bits behind the 12-bit contiguous pass as ONE strided pass
    // through LDS tiles (512 threads, R x TL = 4096 elements) instead of two register passes - two launches instead of three where every launch
    // is latency. Measured (profiles/r4_inv_lds_plan_ab.txt, round 4): 2 columns 2^20 76.0 -> 63.4 us, 1 column 2^20 52.4 -> 46.7, 2 columns 2^18
    // 49.6 -> 43.7, 2^19 50.9 -> 48.5; 2 columns 2^16 (4 bits behind the first pass: one radix-16 register pass) 30.5 -> 33.5, so shorter
    // transforms keep the register pass; eight proofs in flight: unchanged. Round 5 (profiles/r5_inv_last11.md): one column of 2^21 points (the
    // composition polynomial of a 2^20-row proof) the same way with 9 bits in the strided pass (512 x 8 tiles): 66.5 -> 51.6 us.
    const bool lds_plan = this->reg_passes && log_n >= 18 && log_n <= 21 && lds_wide(ncols, log_n);
    const bool reg_passes = this->reg_passes && !lds_plan;          // shadows the member for the rest of this transform
    // 2048-point tiles on launches of at least 2^21 elements - except where 4096-point tiles save a whole strided pass (radix <= 64: the bits
    // behind the contiguous pass are 6 k + 1 with 2048-point tiles, i.e. 2^18- and 2^24-point transforms): 72 columns 2^18 287 -> 260 us,
    // 2 columns 2^24 594 -> 556 us, the other sizes within 2 % either way (profiles/r5_inv_last11.md)
    const bool inv2p = reg_passes && log_n >= 13 && ((size_t)ncols << log_n) >= ((size_t)1 << 21) && (log_n - 11) % 6 != 1;
    std::vector<NttPass> plan = plan_passes(log_n, reg_passes, inv2p ? 11 : 12, false, lds_plan ? 9 : 8);
    const int r1 = plan[0].log_r;
    // per-k table for the final (contiguous) pass
    uint64_t ninv = gl::inv((uint64_t)1 << log_n);
    // natural index I = rev_r1(k) * 2^q + rev(block), q = L - r1 block bits: the k-dependent part of sa^I * sb^(I >> shift) is
    //   q >= shift: (sa^(2^q) * sb^(2^(q-shift)))^rev(k)          (the block contributes sa^rb * sb^(rb >> shift))
    //   q <  shift: (sa^(2^q))^rev(k) * sb^(rev(k) >> (shift-q))  (the block's bits are shifted out: it contributes sa^rb only)
    const int q = log_n - r1;
    uint64_t abase, bbase = 1;
    int bshift = 0;
    if (q >= shift) {
        abase = gl::mul(gl::pow(sa, 1ull << q), gl::pow(sb, 1ull << (q - shift)));
    } else {
        abase = gl::pow(sa, 1ull << q); bbase = sb; bshift = shift - q;
    }
    // the table depends on (size, scale parameters) only: a prover asks for the same two or three tables proof after proof, so
    // they are built once per context and kept (a bounded number; anything beyond that is built per call)
    uint64_t* ktab = nullptr;
    {
        const std::vector<uint64_t> key{(uint64_t)r1, abase, bbase, (uint64_t)bshift, gl::mul(c0, ninv)};
        auto it = ktab_cache.find(key);
        if (it != ktab_cache.end()) ktab = it->second;
        else {
            const bool keep = ktab_cache.size() < 64;
            ktab = keep ? (uint64_t*)dev_alloc(((size_t)1 << r1) * 8) : (uint64_t*)scratch_alloc(((size_t)1 << r1) * 8);
            AERO_LAUNCH(this, "fill_pow_bitrev", 0, fill_pow_bitrev, dim3(((1u << r1) + 255) / 256), dim3(256), 0, ktab, r1, abase, bbase, bshift, key[4]);
            if (keep) ktab_cache[key] = ktab;
        }
    }
    for (size_t qi = plan.size(); qi-- > 0;) {
        PassArgs a{};
        a.in = data; a.out = data; a.in_col_stride = stride; a.out_col_stride = stride;
        a.log_n = log_n; a.log_s = plan[qi].log_s; a.log_r = plan[qi].log_r; a.log_tl = plan[qi].log_tl;
        a.first = (qi + 1 == plan.size());
        if (a.first && src) { a.in = src; a.in_col_stride = src_stride ? src_stride : stride; }
        a.bad = a.first ? bad : nullptr;      // the pass that reads the caller's values
        a.tw_r = tw4096_inv; a.tw_lo = t->lo_inv; a.tw_hi = t->hi_inv; a.tw_h = t->h;
        if (qi == 0) { a.ktab = ktab; a.sc_a = sa; a.sc_b = sb; a.sc_shift = shift; }
        if (qi > 0 && reg_passes) {
            if (!a.first) a.tw_pass = pass_twiddles(log_n, a.log_s, a.log_r, true);
            dim3 rgrid((unsigned)((((size_t)1 << log_n) >> a.log_r) / 256), ncols);
            const size_t abytes = (size_t)ncols * 16 * ((size_t)1 << log_n);
            const char* nm = pass_names ? (a.log_r == 6 ? "ntt_inv_reg6" : a.log_r == 5 ? "ntt_inv_reg5" : a.log_r == 4 ? "ntt_inv_reg4" : "ntt_inv_reg123") : "ntt_inv_pass";
            switch (a.log_r) {
                case 6: AERO_LAUNCH(this, nm, abytes, ntt_inv_strided_reg<6>, rgrid, dim3(256), 0, a); break;
                case 5: AERO_LAUNCH(this, nm, abytes, ntt_inv_strided_reg<5>, rgrid, dim3(256), 0, a); break;
                case 4: AERO_LAUNCH(this, nm, abytes, ntt_inv_strided_reg<4>, rgrid, dim3(256), 0, a); break;
                case 3: AERO_LAUNCH(this, nm, abytes, ntt_inv_strided_reg<3>, rgrid, dim3(256), 0, a); break;
                case 2: AERO_LAUNCH(this, nm, abytes, ntt_inv_strided_reg<2>, rgrid, dim3(256), 0, a); break;
                default: AERO_LAUNCH(this, nm, abytes, ntt_inv_strided_reg<1>, rgrid, dim3(256), 0, a); break;
            }
            continue;
        }
        const bool last12 = this->reg_passes && qi == 0 && a.log_r == 12 && a.log_s == 0 && a.log_tl == 0 && log_n >= 13;
        if ((inv2p && qi == 0 && a.log_r == 11) || last12) {
            // per-tile factors a^rev(b) * b^(rev(b) >> shift): built once per (size, scale parameters) and kept
            const int bbits = log_n - a.log_r;
            const std::vector<uint64_t> key{(uint64_t)(0x100 + bbits), sa, sb, (uint64_t)shift, 0};
            uint64_t* bftab = nullptr;
            auto it = ktab_cache.find(key);
            if (it != ktab_cache.end()) bftab = it->second;
            else {
                const bool keep = ktab_cache.size() < 64;
                bftab = keep ? (uint64_t*)dev_alloc(((size_t)1 << bbits) * 8) : (uint64_t*)scratch_alloc(((size_t)1 << bbits) * 8);
                AERO_LAUNCH(this, "fill_block_factors", 0, fill_block_factors, dim3(((1u << bbits) + 255) / 256), dim3(256), 0, bftab, bbits, sa, sb, shift);
                if (keep) ktab_cache[key] = bftab;
            }
            if (last12) {
                AERO_LAUNCH(this, pass_names ? "ntt_inv_last12" : "ntt_inv_pass", (size_t)ncols * 16 * ((size_t)1 << log_n), ntt_inv_last_pass_12, dim3((unsigned)(((size_t)1 << log_n) >> 12), ncols),
                            dim3(256), 0, a, (const uint64_t*)bftab, (const uint64_t*)twmt12_inv);
                continue;
            }
            a.tw_mt = twmt_inv;
            const size_t tiles = ((size_t)1 << log_n) >> 11;
            AERO_LAUNCH(this, pass_names ? "ntt_inv_last11" : "ntt_inv_pass", (size_t)ncols * 16 * ((size_t)1 << log_n), ntt_inv_last_pass_11, dim3((unsigned)tiles, ncols),
                        dim3(128), 0, a, (const uint64_t*)bftab);
            continue;
        }
        size_t E = (size_t)1 << (a.log_r + a.log_tl);
        dim3 grid((unsigned)(((size_t)1 << log_n) / E), ncols);
        const char* lnm = pass_names ? (a.log_s ? (lds_wide(ncols, log_n) ? "ntt_inv_lds_strided_512" : "ntt_inv_lds_strided_256") : (lds_wide(ncols, log_n) ? "ntt_inv_lds_512" : "ntt_inv_lds_256")) : "ntt_inv_pass";
        if (lds_wide(ncols, log_n)) AERO_LAUNCH(this, lnm, (size_t)ncols * 16 * ((size_t)1 << log_n), ntt_inv_pass<512>, grid, dim3(512), 0, a);
        else AERO_LAUNCH(this, lnm, (size_t)ncols * 16 * ((size_t)1 << log_n), ntt_inv_pass<256>, grid, dim3(256), 0, a);
    }
    check_launch("ntt_inverse");
}

}  // namespace aero
