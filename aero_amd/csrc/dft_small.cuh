// Small in-register DFTs over Goldilocks (shared by the NTT rounds and the FRI fold).
#pragma once
#include "gl.cuh"

namespace aero {
using gl::add;
using gl::mul;
using gl::sub;

// Small-root constants of this field: w_4 = 2^48, w_8 = 2^120 = -2^24, w_8^3 = 2^168 = -2^72 (2 has order 192).
// Products with them are written as multiplications by the constants; the rest of a radix-8 butterfly is adds.
__device__ __forceinline__ uint64_t mul_w4(uint64_t x) { return mul(x, 1ull << 48); }
__device__ __forceinline__ uint64_t mul_2_24(uint64_t x) { return mul(x, 1ull << 24); }
__device__ __forceinline__ uint64_t mul_2_72(uint64_t x) { return mul(x, 0xFFFFFFFF00ull); }   // 2^72 = 2^8 * (2^32 - 1)

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
