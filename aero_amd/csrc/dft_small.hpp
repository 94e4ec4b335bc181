// Small in-register DFTs over Goldilocks (shared by the NTT rounds and the FRI fold).
#pragma once
#include "gl_field.hpp"

namespace aero {
using gl::add;
using gl::mul;
using gl::sub;

// Small-root constants of this field: w_4 = 2^48, w_8 = 2^120 = -2^24, w_8^3 = 2^168 = -2^72 (2 has order 192).
// Products with them are written as multiplications by the constants; the rest of a radix-8 butterfly is adds.
// x * 2^c for the three constants the radix-8 butterflies need, written as shifts + one short reduction instead of a
// full 64x64 multiply (measured ~37 VALU issue slots for gl::mul vs 14-22 here). With 2^64 = 2^32 - 1 =: EPS and
// 2^96 = -1 (mod p):
//   x * 2^24 = lo + hi * EPS                       lo = x << 24 (64 bits), hi = x >> 40
//   x * 2^48 = b0 * 2^32 + b1 * EPS - b2           (b0, b1, b2) = 32-bit limbs of x << 16 (b2 < 2^16)
//   x * 2^72 = c0 * EPS - c1 - c2 * 2^32           (c0, c1, c2) = 32-bit limbs of x << 8  (c2 < 2^8)
// Every intermediate that enters gl::add / gl::sub is canonical (< p) by construction.
__device__ __forceinline__ uint64_t mul_2_24(uint64_t x) {
    const uint64_t lo = x << 24, hi = x >> 40;
    const uint64_t t1 = (hi << 32) - hi;                                   // hi * EPS < 2^56
    uint32_t c, c1, c2;
    uint32_t s0 = __builtin_addc((uint32_t)lo, (uint32_t)t1, 0u, &c), s1 = __builtin_addc((uint32_t)(lo >> 32), (uint32_t)(t1 >> 32), c, &c1);
    const uint32_t m = 0u - c1;                                            // wrapped past 2^64: + EPS (cannot wrap again: s < 2^56)
    s0 = __builtin_addc(s0, m, 0u, &c); s1 = __builtin_addc(s1, 0u, c, &c);
    const uint32_t t0 = __builtin_addc(s0, 0xFFFFFFFFu, 0u, &c), t1h = __builtin_addc(s1, 0u, c, &c2);   // >= p <=> + EPS wraps
    return c2 ? gl::mk64(t0, t1h) : gl::mk64(s0, s1);
}
__device__ __forceinline__ uint64_t mul_w4(uint64_t x) {                   // x * 2^48
    const uint32_t b0 = (uint32_t)(x << 16), b1 = (uint32_t)(x >> 16), b2 = (uint32_t)(x >> 48);
    const uint64_t pos = add(gl::mk64(0u, b0), ((uint64_t)b1 << 32) - b1); // b0 * 2^32 <= p - 1, b1 * EPS < p
    return sub(pos, (uint64_t)b2);
}
__device__ __forceinline__ uint64_t mul_2_72(uint64_t x) {
    const uint32_t c0 = (uint32_t)(x << 8), c1 = (uint32_t)(x >> 24), c2 = (uint32_t)(x >> 56);
    const uint64_t pos = ((uint64_t)c0 << 32) - c0;                        // c0 * EPS < p
    const uint64_t neg = (uint64_t)c1 + ((uint64_t)c2 << 32);              // < 2^41
    return sub(pos, neg);
}

// Plain DFT of 2^RB points held in registers: decimation in time, bit-reversed input -> natural output,
// root w_(2^RB) = w_4096^(4096 >> RB). The butterflies below ARE the radix-2 stages, with the constant twiddles folded:
//   x * w_8 = -(x * 2^24), x * w_4 = x * 2^48, x * w_8^3 = -(x * 2^72)  (negations swap the add and the sub).
template <int RB> __device__ __forceinline__ void dft_dit(uint64_t (&y)[1 << RB]) {
    if constexpr (RB >= 1) {
#pragma unroll
        for (int i = 0; i < (1 << RB); i += 2) { uint64_t u = y[i], v = y[i + 1]; y[i] = add(u, v); y[i + 1] = sub(u, v); }
    }
    if constexpr (RB >= 2) {
#pragma unroll
        for (int i = 0; i < (1 << RB); i += 4) {
            uint64_t u = y[i], v = y[i + 2]; y[i] = add(u, v); y[i + 2] = sub(u, v);
            u = y[i + 1]; v = mul_w4(y[i + 3]); y[i + 1] = add(u, v); y[i + 3] = sub(u, v);
        }
    }
    if constexpr (RB >= 3) {
        uint64_t u, v;
        u = y[0]; v = y[4];            y[0] = add(u, v); y[4] = sub(u, v);
        u = y[1]; v = mul_2_24(y[5]);  y[1] = sub(u, v); y[5] = add(u, v);      // * w_8   = -2^24
        u = y[2]; v = mul_w4(y[6]);    y[2] = add(u, v); y[6] = sub(u, v);      // * w_8^2 =  2^48
        u = y[3]; v = mul_2_72(y[7]);  y[3] = sub(u, v); y[7] = add(u, v);      // * w_8^3 = -2^72
    }
}
// Exact inverse up to the factor 2^RB: decimation in frequency with inverse roots, natural input -> bit-reversed output.
//   w_8^-1 = 2^72, w_4^-1 = -2^48, w_8^-3 = 2^24.
template <int RB> __device__ __forceinline__ void dft_dif_inv(uint64_t (&y)[1 << RB]) {
    if constexpr (RB >= 3) {
        uint64_t u, v;
        u = y[0]; v = y[4]; y[0] = add(u, v); y[4] = sub(u, v);
        u = y[1]; v = y[5]; y[1] = add(u, v); y[5] = mul_2_72(sub(u, v));       // * w_8^-1
        u = y[2]; v = y[6]; y[2] = add(u, v); y[6] = mul_w4(sub(v, u));         // * w_8^-2 = -2^48
        u = y[3]; v = y[7]; y[3] = add(u, v); y[7] = mul_2_24(sub(u, v));       // * w_8^-3
    }
    if constexpr (RB >= 2) {
#pragma unroll
        for (int i = 0; i < (1 << RB); i += 4) {
            uint64_t u = y[i], v = y[i + 2]; y[i] = add(u, v); y[i + 2] = sub(u, v);
            u = y[i + 1]; v = y[i + 3]; y[i + 1] = add(u, v); y[i + 3] = mul_w4(sub(v, u));   // * w_4^-1 = -2^48
        }
    }
    if constexpr (RB >= 1) {
#pragma unroll
        for (int i = 0; i < (1 << RB); i += 2) { uint64_t u = y[i], v = y[i + 1]; y[i] = add(u, v); y[i + 1] = sub(u, v); }
    }
}

// ---- DFTs of up to 64 points entirely in registers -----------------------------------------------------------------------
// 2 has order 192 in this field and the 64-th root of unity the NTT uses is w_64 = 2^39 (w_32 = 2^78, w_16 = 2^156 = -2^60,
// w_8 = 2^120 = -2^24, w_4 = 2^48): every twiddle of a transform of up to 64 points is a power of two, i.e. a shift plus one
// short reduction. x * 2^K for a compile-time K in [0, 96), x canonical, with phi = 2^32, phi^2 = phi - 1, phi^3 = -1:
//   (y0, y1, y2) = 32-bit limbs of x << (K mod 32)            (y2 < 2^(K mod 32))
//   K < 32:  y0 + y1 phi + y2 (phi - 1)            K < 64:  y0 phi + y1 (phi - 1) - y2            K < 96:  y0 (phi - 1) - y1 - y2 phi
template <int K> __device__ __forceinline__ uint64_t mul_pow2(uint64_t x) {
    static_assert(K >= 0 && K < 96, "mul_pow2: exponent out of range");
    if constexpr (K == 0) return x;
    else if constexpr (K == 24) return mul_2_24(x);
    else if constexpr (K == 48) return mul_w4(x);
    else if constexpr (K == 72) return mul_2_72(x);
    else {
        constexpr int q = K / 32, r = K % 32;
        const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32);
        const uint32_t y0 = x0 << r;
        const uint32_t y1 = r ? ((x0 >> ((32 - r) & 31)) | (x1 << r)) : x1;
        const uint32_t y2 = r ? (x1 >> ((32 - r) & 31)) : 0u;
        if constexpr (q == 0) {
            // (y1:y0) may exceed p, but (y1:y0) + y2 EPS < 2^64 + 2^63 < 2p: gl::add's single conditional "+ EPS" (taken when the sum wraps past
            // 2^64 or lands in [p, 2^64)) already returns the canonical residue - no canonicalisation of the first operand (4 instructions)
            return add(gl::mk64(y0, y1), ((uint64_t)y2 << 32) - y2);
        } else if constexpr (q == 1) {
            return sub(add(gl::mk64(0u, y0), ((uint64_t)y1 << 32) - y1), (uint64_t)y2);
        } else {
            return sub(((uint64_t)y0 << 32) - y0, gl::mk64(y1, y2));
        }
    }
}
// v * w_64^E for a compile-time E in [0, 64) folded into the butterfly: returns (u + t, u - t) with t = v * 2^(39 E mod 192);
// exponents >= 96 are a negation (2^96 = -1), which swaps the two outputs instead of costing anything.
template <int E> __device__ __forceinline__ void bfly_w64(uint64_t& u, uint64_t& v) {
    constexpr int K = (39 * E) % 192;
    if constexpr (K < 96) { const uint64_t t = mul_pow2<K>(v); const uint64_t a = add(u, t), b = sub(u, t); u = a; v = b; }
    else { const uint64_t t = mul_pow2<K - 96>(v); const uint64_t a = sub(u, t), b = add(u, t); u = a; v = b; }
}
// inverse direction: (u, v) -> (u + v, (u - v) * w_64^-E); w_64^-E = 2^(192 - 39 E mod 192)
template <int E> __device__ __forceinline__ void bfly_w64_inv(uint64_t& u, uint64_t& v) {
    constexpr int K = (192 - (39 * E) % 192) % 192;
    const uint64_t a = add(u, v);
    if constexpr (K < 96) v = mul_pow2<K>(sub(u, v));
    else v = mul_pow2<K - 96>(sub(v, u));
    u = a;
}
// One radix-2 stage T (sub-transform size 2^T) of an in-register transform of 2^LOG points, fully unrolled at compile time.
template <int LOG, int T, int I, bool INV> struct StageUnroll {
    static __device__ __forceinline__ void run(uint64_t (&y)[1 << LOG]) {
        constexpr int half = 1 << (T - 1);
        constexpr int j = I & (half - 1), grp = I >> (T - 1);
        constexpr int lo = (grp << T) + j, hi = lo + half;
        constexpr int E = j * (64 >> T);          // w_(2^T)^j = w_64^(j * 64 / 2^T)
        if constexpr (!INV) bfly_w64<E>(y[lo], y[hi]);
        else bfly_w64_inv<E>(y[lo], y[hi]);
        if constexpr (I + 1 < (1 << (LOG - 1))) StageUnroll<LOG, T, I + 1, INV>::run(y);
    }
};
// Forward: decimation in time, bit-reversed input -> natural output (same convention as dft_dit, any LOG <= 6).
template <int LOG, int T = 1> __device__ __forceinline__ void dft_dit_reg(uint64_t (&y)[1 << LOG]) {
    static_assert(LOG >= 1 && LOG <= 6, "register DFT of up to 64 points");
    StageUnroll<LOG, T, 0, false>::run(y);
    if constexpr (T < LOG) dft_dit_reg<LOG, T + 1>(y);
}
// Inverse up to the factor 2^LOG: decimation in frequency with inverse roots, natural input -> bit-reversed output.
template <int LOG, int T = LOG> __device__ __forceinline__ void dft_dif_inv_reg(uint64_t (&y)[1 << LOG]) {
    static_assert(LOG >= 1 && LOG <= 6, "register DFT of up to 64 points");
    StageUnroll<LOG, T, 0, true>::run(y);
    if constexpr (T > 1) dft_dif_inv_reg<LOG, T - 1>(y);
}

}  // namespace aero
