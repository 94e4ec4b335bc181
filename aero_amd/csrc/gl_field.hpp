// Goldilocks field arithmetic for gfx950 (and the same code for the host side of the prover).
//
// Replaces winter-math 0.4 `fields::f64::BaseElement` / `QuadExtension<BaseElement>` as used by the reference
// through `miden_core::Felt` (/root/reference/miden-proof-generator/src/main.rs:3,
// aero-sdk/miden-wasm/src/utils.rs:365-409 canonical `as_int()` <-> `new()`); constants from
// /root/reference/src/utils/math_goldilocks.cairo:4 and src/stark_verifier/fri/fri_verifier.cairo:154-155.
//
// gfx950 has no 64x64 multiplier: a field multiply is four 32x32->64 multiply-adds (v_mad_u64_u32) plus the
// 2^64 = 2^32 - 1 (mod p) fold. MFMA is deliberately not used: 64-bit modular mul-add is not a dense fp
// contraction. All values are kept canonical (< p) between operations so results are bit-exact by construction.
#pragma once
#ifndef __HIPCC_RTC__          // compiled at run time too (air_jit.hip embeds this file): hiprtc brings the HIP built-ins itself
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
typedef unsigned long long uint64_t;
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned char uint8_t;
typedef unsigned long size_t;
#endif

#define GL_HD __host__ __device__ __forceinline__

namespace gl {

constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
constexpr uint64_t EPS = 0xFFFFFFFFULL;           // 2^64 mod p
constexpr uint64_t GEN = 7;                       // multiplicative generator = LDE / FRI domain offset
constexpr uint64_t ROOT_2_32 = 1753635133440165772ULL;
constexpr int TWO_ADICITY = 32;

// Device code spells the carry chains out on 32-bit limbs (v_add_co/v_addc_co, no 64-bit compares): measured on MI355X
// (tools/ubench_field.hip) add 6.7 T/s vs 4.8, sub 7.7 vs 6.2, mul 1.65 vs 1.34 T/s for the plain u64 formulation.
GL_HD uint64_t mk64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
GL_HD uint64_t add(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(GL_LAZY_ADD_UNSAFE)
    // MEASUREMENT ONLY (tools/ubench_dft.hip): one wrap correction, no canonicalisation, a second wrap goes unnoticed - not a field add.
    uint32_t c, c1;
    uint32_t s0 = __builtin_addc((uint32_t)a, (uint32_t)b, 0u, &c), s1 = __builtin_addc((uint32_t)(a >> 32), (uint32_t)(b >> 32), c, &c1);
    const uint32_t m = 0u - c1;
    s0 = __builtin_addc(s0, m, 0u, &c); s1 = __builtin_addc(s1, 0u, c, &c);
    return mk64(s0, s1);
#elif defined(__HIP_DEVICE_COMPILE__) && !defined(GL_PLAIN_ADD)
    // a + b < 2p. Result is s + EPS (mod 2^64) iff the sum wrapped past 2^64 or s >= p (<=> s + EPS wraps).
    uint32_t c, c1, c2;
    uint32_t s0 = __builtin_addc((uint32_t)a, (uint32_t)b, 0u, &c), s1 = __builtin_addc((uint32_t)(a >> 32), (uint32_t)(b >> 32), c, &c1);
    uint32_t t0 = __builtin_addc(s0, 0xFFFFFFFFu, 0u, &c), t1 = __builtin_addc(s1, 0u, c, &c2);
    return (c1 | c2) ? mk64(t0, t1) : mk64(s0, s1);
#else
    uint64_t s = a + b;
    uint64_t c = (s < a) ? EPS : 0;               // wrapped past 2^64: + (2^64 mod p)
    s += c;
    return s >= P ? s - P : s;
#endif
}
// NOTE: the carry-chain form of sub is NOT the default: ROCm 7.2's LLVM folds its high limb `d1 - borrow` into the high limb of a
// following carry-chain add as `d1 + sext(borrow) + carry` and keeps using THAT instruction's carry-out, which is a different
// function of the operands (profiles/r3_carry_sub_miscompile.md has the instruction-level account). GL_CARRY_SUB selects it with an
// optimisation barrier on the high limb that keeps the combiner from looking through the subtraction.
GL_HD uint64_t sub(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(GL_CARRY_SUB)
    uint32_t c, c1;
    uint32_t d0 = __builtin_subc((uint32_t)a, (uint32_t)b, 0u, &c), d1 = __builtin_subc((uint32_t)(a >> 32), (uint32_t)(b >> 32), c, &c1);
    uint32_t t0 = __builtin_subc(d0, 0xFFFFFFFFu, 0u, &c), t1 = __builtin_subc(d1, 0u, c, &c);   // d + p = d - EPS (mod 2^64)
    uint32_t hi = c1 ? t1 : d1;
    asm volatile("" : "+v"(hi));
    return mk64(c1 ? t0 : d0, hi);
#else
    uint64_t d = a - b;
    return a < b ? d + P : d;                     // a - b + p, computed mod 2^64
#endif
}
GL_HD uint64_t neg(uint64_t a) { return a ? P - a : 0; }
GL_HD uint64_t dbl(uint64_t a) { return add(a, a); }

GL_HD void mul_wide(uint64_t a, uint64_t b, uint64_t& lo, uint64_t& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    lo = a * b;
    hi = __umul64hi(a, b);
#else
    unsigned __int128 x = (unsigned __int128)a * b;
    lo = (uint64_t)x;
    hi = (uint64_t)(x >> 64);
#endif
}
// hi*2^64 + lo  ->  canonical residue. Uses 2^64 = 2^32 - 1, 2^96 = -1 (mod p).
GL_HD uint64_t reduce128(uint64_t lo, uint64_t hi) {
    uint64_t hh = hi >> 32, hl = hi & EPS;
    uint64_t t0 = lo - hh;
    if (lo < hh) t0 -= EPS;                       // borrow: adding p back == subtracting 2^32 - 1 (mod 2^64)
    uint64_t t1 = hl * EPS;                       // < 2^64
    uint64_t r = t0 + t1;
    if (r < t1) r += EPS;
    return r >= P ? r - P : r;
}
GL_HD uint64_t mul(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GL_PLAIN_MUL)
    // 4 x v_mad_u64_u32 schoolbook product (x0..x3), then x0 + x1 2^32 + x2 (2^32 - 1) - x3 on carry chains
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t t = (uint64_t)a0 * b0;
    const uint64_t u = (uint64_t)a0 * b1 + (t >> 32);
    const uint64_t v = (uint64_t)a1 * b0 + (uint32_t)u;
    const uint64_t w = (uint64_t)a1 * b1 + ((u >> 32) + (v >> 32));
    const uint32_t x0 = (uint32_t)t, x1 = (uint32_t)v, x2 = (uint32_t)w, x3 = (uint32_t)(w >> 32);
    uint32_t c, bo, ca, c2;
    uint32_t l0 = __builtin_subc(x0, x3, 0u, &c), l1 = __builtin_subc(x1, 0u, c, &bo);      // lo - x3
    const uint32_t m = 0u - bo;                                                             // borrow: - EPS
    l0 = __builtin_subc(l0, m, 0u, &c); l1 = __builtin_subc(l1, 0u, c, &c);
    const uint32_t e0 = __builtin_subc(0u, x2, 0u, &c), e1 = __builtin_subc(x2, 0u, c, &c);  // x2 * EPS
    uint32_t r0 = __builtin_addc(l0, e0, 0u, &c), r1 = __builtin_addc(l1, e1, c, &ca);
    const uint32_t m2 = 0u - ca;                                                            // carry: + EPS
    r0 = __builtin_addc(r0, m2, 0u, &c); r1 = __builtin_addc(r1, 0u, c, &c);
    const uint32_t t0 = __builtin_addc(r0, 0xFFFFFFFFu, 0u, &c), t1 = __builtin_addc(r1, 0u, c, &c2);   // r >= p <=> r + EPS wraps
    return c2 ? mk64(t0, t1) : mk64(r0, r1);
#else
    uint64_t lo, hi;
    mul_wide(a, b, lo, hi);
    return reduce128(lo, hi);
#endif
}
GL_HD uint64_t sqr(uint64_t a) { return mul(a, a); }

// Sum of products as a 160-bit integer (up to 2^32 terms), reduced ONCE: the constraint and DEEP kernels spend most of their
// multiplications on sum_k coefficient_k * value_k, and a term costs 4 multiply-adds + 5 carry adds here against a whole field
// multiplication + addition (about 27 instructions) when every product is reduced. 2^64 = 2^32 - 1, 2^128 = -2^32 (mod p).
struct Wide { uint32_t l0, l1, l2, l3, l4; };
GL_HD Wide wzero() { return Wide{0u, 0u, 0u, 0u, 0u}; }
GL_HD void wmac(Wide& w, uint64_t x, uint64_t y) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), y0 = (uint32_t)y, y1 = (uint32_t)(y >> 32);
    const uint64_t t = (uint64_t)x0 * y0;
    const uint64_t u = (uint64_t)x0 * y1 + (t >> 32);
    const uint64_t v = (uint64_t)x1 * y0 + (uint32_t)u;
    const uint64_t z = (uint64_t)x1 * y1 + ((u >> 32) + (v >> 32));
    uint32_t c;
    w.l0 = __builtin_addc(w.l0, (uint32_t)t, 0u, &c);
    w.l1 = __builtin_addc(w.l1, (uint32_t)v, c, &c);
    w.l2 = __builtin_addc(w.l2, (uint32_t)z, c, &c);
    w.l3 = __builtin_addc(w.l3, (uint32_t)(z >> 32), c, &c);
    w.l4 += c;
#else
    uint64_t lo, hi;
    mul_wide(x, y, lo, hi);
    unsigned __int128 s = ((unsigned __int128)mk64(w.l2, w.l3) << 64 | mk64(w.l0, w.l1));
    const unsigned __int128 q = ((unsigned __int128)hi << 64) | lo, r = s + q;
    w.l4 += r < s ? 1u : 0u;
    w.l0 = (uint32_t)r; w.l1 = (uint32_t)(r >> 32); w.l2 = (uint32_t)(r >> 64); w.l3 = (uint32_t)(r >> 96);
#endif
}
GL_HD uint64_t wreduce(const Wide& w) {
    const uint64_t r = reduce128(mk64(w.l0, w.l1), mk64(w.l2, w.l3));
    return sub(r, mul((uint64_t)w.l4, 1ULL << 32));
}
GL_HD uint64_t pow(uint64_t b, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = mul(r, b);
        b = sqr(b);
        e >>= 1;
    }
    return r;
}
// a^(p-2), p - 2 = 2^64 - 2^32 - 1 = (2^32 - 2) * 2^32 + (2^32 - 1). inv(0) = 0 (winter-math convention).
GL_HD uint64_t sqr_n(uint64_t x, int n) {
#pragma unroll 1
    for (int i = 0; i < n; i++) x = sqr(x);
    return x;
}
// a^(p - 2), p - 2 = (2^32 - 2) 2^32 + (2^32 - 1): a^(2^31 - 1) by the chain 1, 2, 4, 5, 10, 15, 30, 31 (exponents 2^k - 1), then 33 squarings:
// 63 squarings + 9 multiplications (the square-and-multiply ladder this replaces took 63 + 32)
GL_HD uint64_t inv(uint64_t a) {
    const uint64_t x2 = mul(sqr(a), a);
    const uint64_t x4 = mul(sqr_n(x2, 2), x2);
    const uint64_t x5 = mul(sqr(x4), a);
    const uint64_t x10 = mul(sqr_n(x5, 5), x5);
    const uint64_t x15 = mul(sqr_n(x10, 5), x5);
    const uint64_t x30 = mul(sqr_n(x15, 15), x15);
    const uint64_t x31 = mul(sqr(x30), a);         // a^(2^31 - 1)
    uint64_t y = sqr(x31);                         // a^(2^32 - 2)
    const uint64_t t = mul(y, a);                  // a^(2^32 - 1)
    y = sqr_n(y, 32);                              // a^((2^32 - 2) * 2^32)
    return mul(y, t);
}
GL_HD uint64_t root_of_unity(int log_n) { return pow(ROOT_2_32, 1ULL << (TWO_ADICITY - log_n)); }

// ---- quadratic extension F_p[phi]/(phi^2 - phi + 2) ------------------------------------------------
struct E2 {
    uint64_t a0, a1;
};
GL_HD E2 e2(uint64_t a) { return E2{a, 0}; }
GL_HD E2 add(E2 a, E2 b) { return E2{add(a.a0, b.a0), add(a.a1, b.a1)}; }
GL_HD E2 sub(E2 a, E2 b) { return E2{sub(a.a0, b.a0), sub(a.a1, b.a1)}; }
GL_HD E2 mul(E2 a, E2 b) {
    uint64_t a0b0 = mul(a.a0, b.a0), a1b1 = mul(a.a1, b.a1);
    uint64_t s = mul(add(a.a0, a.a1), add(b.a0, b.a1));
    return E2{sub(a0b0, dbl(a1b1)), sub(s, a0b0)};
}
GL_HD E2 mulb(E2 a, uint64_t b) { return E2{mul(a.a0, b), mul(a.a1, b)}; }
GL_HD E2 conj(E2 a) { return E2{add(a.a0, a.a1), neg(a.a1)}; }
GL_HD E2 inv(E2 a) {
    uint64_t n = add(add(sqr(a.a0), mul(a.a0, a.a1)), dbl(sqr(a.a1)));
    uint64_t ni = inv(n);
    E2 c = conj(a);
    return E2{mul(c.a0, ni), mul(c.a1, ni)};
}

// ---- uniform interface over E = F_p and E = F_p^2 (kernels are templated on one of these) ----------
struct FB {
    typedef uint64_t T;
    static constexpr int DEG = 1;
    GL_HD static T zero() { return 0; }
    GL_HD static T one() { return 1; }
    GL_HD static T from(uint64_t b) { return b; }
    GL_HD static T add(T a, T b) { return gl::add(a, b); }
    GL_HD static T sub(T a, T b) { return gl::sub(a, b); }
    GL_HD static T mul(T a, T b) { return gl::mul(a, b); }
    GL_HD static T mulb(T a, uint64_t b) { return gl::mul(a, b); }
    GL_HD static T inv(T a) { return gl::inv(a); }
    GL_HD static T conj(T a) { return a; }
    GL_HD static uint64_t comp(T a, int) { return a; }
    GL_HD static T make(uint64_t c0, uint64_t) { return c0; }
};
struct FQ {
    typedef E2 T;
    static constexpr int DEG = 2;
    GL_HD static T zero() { return E2{0, 0}; }
    GL_HD static T one() { return E2{1, 0}; }
    GL_HD static T from(uint64_t b) { return E2{b, 0}; }
    GL_HD static T add(T a, T b) { return gl::add(a, b); }
    GL_HD static T sub(T a, T b) { return gl::sub(a, b); }
    GL_HD static T mul(T a, T b) { return gl::mul(a, b); }
    GL_HD static T mulb(T a, uint64_t b) { return gl::mulb(a, b); }
    GL_HD static T inv(T a) { return gl::inv(a); }
    GL_HD static T conj(T a) { return gl::conj(a); }
    GL_HD static uint64_t comp(T a, int i) { return i ? a.a1 : a.a0; }
    GL_HD static T make(uint64_t c0, uint64_t c1) { return E2{c0, c1}; }
};
template <class F> GL_HD typename F::T fpow(typename F::T b, uint64_t e) {
    typename F::T r = F::one();
    while (e) {
        if (e & 1) r = F::mul(r, b);
        b = F::mul(b, b);
        e >>= 1;
    }
    return r;
}

GL_HD uint32_t bitrev(uint32_t x, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
#endif
}

}  // namespace gl
