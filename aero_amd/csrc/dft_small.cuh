// Small in-register DFTs over Goldilocks (shared by the NTT rounds and the FRI fold).
#pragma once
#include "gl.cuh"

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


}  // namespace aero
