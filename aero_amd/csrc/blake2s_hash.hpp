// BLAKE2s-256 compression for gfx950 (and the host side of the transcript).
//
// Replaces winter-crypto 0.4 `Blake2s_256<Felt>` as called by the reference at
// /root/reference/aero-sdk/miden-wasm/src/hashing_worker.rs:16 (`Blake2s_256::hash_elements(row)`), with the fork's
// element serialisation mirrored in /root/reference/src/stark_verifier/crypto/random.cairo:93-104: every u64
// element occupies 32 bytes (8 data bytes, 24 zero bytes), so a 64-byte block carries two elements and message
// words 2..7 and 10..15 are always zero. The compression below is specialised on that fact: the zero words are
// compile-time zeros and their additions vanish.
// One thread = one hash state; the 16-word working vector stays in VGPRs (fully unrolled rounds, constant
// sigma indices), rotations are v_alignbit_b32, a + b + m is v_add3_u32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define B2_HD __host__ __device__ __forceinline__

namespace b2s {

constexpr uint32_t IV0 = 0x6A09E667u, IV1 = 0xBB67AE85u, IV2 = 0x3C6EF372u, IV3 = 0xA54FF53Au, IV4 = 0x510E527Fu,
                   IV5 = 0x9B05688Cu, IV6 = 0x1F83D9ABu, IV7 = 0x5BE0CD19u;
constexpr uint32_t PARAM0 = 0x01010020u;   // digest 32 bytes, no key, fanout 1, depth 1

B2_HD uint32_t rotr(uint32_t x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(x, x, n);
#else
    return (x >> n) | (x << (32 - n));
#endif
}

struct State {
    uint32_t h[8];
};
B2_HD void init(State& s) {
    s.h[0] = IV0 ^ PARAM0; s.h[1] = IV1; s.h[2] = IV2; s.h[3] = IV3; s.h[4] = IV4; s.h[5] = IV5; s.h[6] = IV6; s.h[7] = IV7;
}

#define B2_G(a, b, c, d, x, y)                 \
    a = a + b + (x); d = rotr(d ^ a, 16);      \
    c = c + d;       b = rotr(b ^ c, 12);      \
    a = a + b + (y); d = rotr(d ^ a, 8);       \
    c = c + d;       b = rotr(b ^ c, 7);

#define B2_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    B2_G(v0, v4, v8, v12, m[s0], m[s1]);   B2_G(v1, v5, v9, v13, m[s2], m[s3]);         \
    B2_G(v2, v6, v10, v14, m[s4], m[s5]);  B2_G(v3, v7, v11, v15, m[s6], m[s7]);        \
    B2_G(v0, v5, v10, v15, m[s8], m[s9]);  B2_G(v1, v6, v11, v12, m[s10], m[s11]);      \
    B2_G(v2, v7, v8, v13, m[s12], m[s13]); B2_G(v3, v4, v9, v14, m[s14], m[s15]);

// Generic compression of one 64-byte block given as 16 words.
B2_HD void compress(State& s, const uint32_t m[16], uint32_t t_lo, uint32_t t_hi, bool last) {
    uint32_t v0 = s.h[0], v1 = s.h[1], v2 = s.h[2], v3 = s.h[3], v4 = s.h[4], v5 = s.h[5], v6 = s.h[6], v7 = s.h[7];
    uint32_t v8 = IV0, v9 = IV1, v10 = IV2, v11 = IV3, v12 = IV4 ^ t_lo, v13 = IV5 ^ t_hi, v14 = last ? ~IV6 : IV6, v15 = IV7;
    B2_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    B2_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    B2_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    B2_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    B2_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    B2_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    B2_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    B2_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    B2_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    B2_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    s.h[0] ^= v0 ^ v8;  s.h[1] ^= v1 ^ v9;  s.h[2] ^= v2 ^ v10; s.h[3] ^= v3 ^ v11;
    s.h[4] ^= v4 ^ v12; s.h[5] ^= v5 ^ v13; s.h[6] ^= v6 ^ v14; s.h[7] ^= v7 ^ v15;
}

// Block made of two 32-byte-padded field elements (e1 ignored and the block is 32 bytes long when !two).
// `t` = bytes hashed so far including this block.
B2_HD void compress_elems(State& s, uint64_t e0, uint64_t e1, bool two, uint32_t t, bool last) {
    uint32_t m[16] = {(uint32_t)e0, (uint32_t)(e0 >> 32), 0, 0, 0, 0, 0, 0,
                      two ? (uint32_t)e1 : 0u, two ? (uint32_t)(e1 >> 32) : 0u, 0, 0, 0, 0, 0, 0};
    compress(s, m, t, 0, last);
}

struct Digest {
    uint32_t w[8];
};

// node = BLAKE2s(left || right): one final compression of a full 64-byte block
// (/root/reference/src/stark_verifier/channel.cairo:157-175, random.cairo:330-342).
B2_HD Digest merge(const Digest& l, const Digest& r) {
    State s; init(s);
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { m[i] = l.w[i]; m[8 + i] = r.w[i]; }
    compress(s, m, 64, 0, true);
    Digest d;
#pragma unroll
    for (int i = 0; i < 8; i++) d.w[i] = s.h[i];
    return d;
}
// BLAKE2s(seed || LE64(v)) — 40 bytes (random.cairo:67-91)
B2_HD Digest merge_with_int(const Digest& seed, uint64_t v) {
    State s; init(s);
    uint32_t m[16] = {seed.w[0], seed.w[1], seed.w[2], seed.w[3], seed.w[4], seed.w[5], seed.w[6], seed.w[7],
                      (uint32_t)v, (uint32_t)(v >> 32), 0, 0, 0, 0, 0, 0};
    compress(s, m, 40, 0, true);
    Digest d;
#pragma unroll
    for (int i = 0; i < 8; i++) d.w[i] = s.h[i];
    return d;
}
// BLAKE2s of a 32-byte string (random_coin_new: random.cairo:31-37)
B2_HD Digest hash32(const Digest& x) {
    State s; init(s);
    uint32_t m[16] = {x.w[0], x.w[1], x.w[2], x.w[3], x.w[4], x.w[5], x.w[6], x.w[7], 0, 0, 0, 0, 0, 0, 0, 0};
    compress(s, m, 32, 0, true);
    Digest d;
#pragma unroll
    for (int i = 0; i < 8; i++) d.w[i] = s.h[i];
    return d;
}
// hash_elements over a small array (host transcript and small device rows)
B2_HD Digest hash_elements(const uint64_t* e, uint32_t n) {
    State s; init(s);
    if (n == 0) {
        uint32_t m[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        compress(s, m, 0, 0, true);
    }
    for (uint32_t i = 0; i < n; i += 2) {
        bool two = i + 1 < n;
        uint32_t t = (two ? i + 2 : i + 1) * 32;
        compress_elems(s, e[i], two ? e[i + 1] : 0, two, t, t == n * 32);
    }
    Digest d;
    for (int i = 0; i < 8; i++) d.w[i] = s.h[i];
    return d;
}

}  // namespace b2s
