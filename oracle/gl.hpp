// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped MI355X path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Goldilocks field F_p, p = 2^64 - 2^32 + 1, plus the quadratic extension F_p[phi]/(phi^2 - phi + 2).
// CPU restatement written against:
//   * /root/reference/src/utils/math_goldilocks.cairo:4 (PG = 18446744069414584321), :7-148 (add/sub/mul/inv/pow)
//   * /root/reference/src/stark_verifier/fri/fri_verifier.cairo:154-168 (2^32-th root 1753635133440165772,
//     get_root_of_unity(n) = G^(2^(32-n)))
//   * third-party: winter-math 0.4 `fields::f64::BaseElement` and `QuadExtension` (submodule absent from the
//     reference mount; published algorithm restated: canonical u64 < p on serialisation, generator 7,
//     quadratic extension x^2 - x + 2, mul = [a0b0 - 2 a1b1, (a0+a1)(b0+b1) - a0b0]).
// This file uses unsigned __int128 products (host-only) and shares no code with the 32-bit-limb device
// implementation in aero_amd/csrc/.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>

namespace orc {

typedef unsigned __int128 u128;
static const uint64_t P = 0xFFFFFFFF00000001ULL;   // math_goldilocks.cairo:4
static const uint64_t GEN = 7;                     // fri_verifier.cairo:23, composer.cairo:24
static const uint64_t ROOT_2_32 = 1753635133440165772ULL;  // fri_verifier.cairo:155
static const int TWO_ADICITY = 32;                 // fri_verifier.cairo:154

// Inputs are canonical (< p); outputs are canonical.
static inline uint64_t gl_add(uint64_t a, uint64_t b) { uint64_t s = a + b; return (s < a || s >= P) ? s - P : s; }
static inline uint64_t gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (P - b); }
// Reference multiply: 128-bit product, generic remainder. Slow; kept as the cross-check for gl_mul.
static inline uint64_t gl_mul_slow(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % P); }
// Fast multiply: 2^64 = 2^32 - 1 and 2^96 = -1 (mod p), so hi*2^64 + lo = lo - (hi >> 32) + (hi & M)*M, M = 2^32 - 1.
// Always returns the canonical representative (< p). tests/test_oracle_field.py checks it against gl_mul_slow.
static inline uint64_t gl_mul(uint64_t a, uint64_t b) {
    u128 x = (u128)a * b;
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hh = hi >> 32, hl = hi & 0xFFFFFFFFULL;
    uint64_t t0 = lo - hh;
    if (lo < hh) t0 -= 0xFFFFFFFFULL;            // borrow: add p back (== subtract 2^32 - 1 mod 2^64)
    uint64_t t1 = hl * 0xFFFFFFFFULL;
    uint64_t r = t0 + t1;
    if (r < t1) r += 0xFFFFFFFFULL;              // carry: 2^64 = 2^32 - 1
    return r >= P ? r - P : r;
}
static inline uint64_t gl_neg(uint64_t a) { return a ? P - a : 0; }
static inline uint64_t gl_pow(uint64_t b, uint64_t e) {
    uint64_t r = 1;
    while (e) { if (e & 1) r = gl_mul(r, b); b = gl_mul(b, b); e >>= 1; }
    return r;
}
static inline uint64_t gl_inv(uint64_t a) { return gl_pow(a, P - 2); }   // inv(0) = 0 as in winter-math
static inline uint64_t gl_div(uint64_t a, uint64_t b) { return gl_mul(a, gl_inv(b)); }
// fri_verifier.cairo:157-168 / air_instance.cairo:85-90
static inline uint64_t gl_root_of_unity(int log_n) { return gl_pow(ROOT_2_32, 1ULL << (TWO_ADICITY - log_n)); }

// Quadratic extension element a0 + a1*phi, phi^2 = phi - 2.
struct Fe2 { uint64_t a0, a1; };
static inline Fe2 e2(uint64_t a) { return Fe2{a, 0}; }
static inline bool e2_eq(Fe2 a, Fe2 b) { return a.a0 == b.a0 && a.a1 == b.a1; }
static inline Fe2 e2_add(Fe2 a, Fe2 b) { return Fe2{gl_add(a.a0, b.a0), gl_add(a.a1, b.a1)}; }
static inline Fe2 e2_sub(Fe2 a, Fe2 b) { return Fe2{gl_sub(a.a0, b.a0), gl_sub(a.a1, b.a1)}; }
static inline Fe2 e2_mul(Fe2 a, Fe2 b) {
    uint64_t a0b0 = gl_mul(a.a0, b.a0), a1b1 = gl_mul(a.a1, b.a1);
    uint64_t s = gl_mul(gl_add(a.a0, a.a1), gl_add(b.a0, b.a1));
    return Fe2{gl_sub(a0b0, gl_add(a1b1, a1b1)), gl_sub(s, a0b0)};
}
static inline Fe2 e2_mulb(Fe2 a, uint64_t b) { return Fe2{gl_mul(a.a0, b), gl_mul(a.a1, b)}; }
static inline Fe2 e2_conj(Fe2 a) { return Fe2{gl_add(a.a0, a.a1), gl_neg(a.a1)}; }   // Frobenius
static inline Fe2 e2_inv(Fe2 a) {
    // norm = a * conj(a) lies in the base field: a0^2 + a0 a1 + 2 a1^2
    uint64_t n = gl_add(gl_add(gl_mul(a.a0, a.a0), gl_mul(a.a0, a.a1)), gl_mul(2, gl_mul(a.a1, a.a1)));
    uint64_t ni = gl_inv(n);
    Fe2 c = e2_conj(a);
    return Fe2{gl_mul(c.a0, ni), gl_mul(c.a1, ni)};
}
static inline Fe2 e2_pow(Fe2 b, uint64_t e) {
    Fe2 r = e2(1);
    while (e) { if (e & 1) r = e2_mul(r, b); b = e2_mul(b, b); e >>= 1; }
    return r;
}

}  // namespace orc
