// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/gl.hpp header).
//
// BLAKE2s-256 (RFC 7693, unkeyed, 32-byte digest) and the element-hashing convention of the reference's
// Winterfell fork:
//   * hash_elements: every u64 field element is serialised as a 32-byte little-endian value (8 bytes of
//     data + 24 zero bytes) — /root/reference/src/stark_verifier/crypto/random.cairo:93-104
//     (`blake2s_add_felts(..., bigend=0)`, `n_bytes = n_elements * 32`); call site on the prover side
//     /root/reference/aero-sdk/miden-wasm/src/hashing_worker.rs:16 (`Blake2s_256::hash_elements(row)`).
//   * merge(a, b) = BLAKE2s(a || b), 64 bytes, one compression — random.cairo:330-342, channel.cairo:168.
//   * merge_with_int(seed, v) = BLAKE2s(seed || LE64(v)), 40 bytes — random.cairo:67-91.
// Third-party note: the hash itself lives in the `blake2` crate used by winter-crypto (absent from the
// mount); RFC 7693 is the published algorithm restated here.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>

namespace orc {

static const uint32_t B2S_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                   0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t B2S_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};

static inline uint32_t b2s_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

// One compression: h updated in place with 64-byte block m (as 16 LE words), byte counter t, final flag.
static inline void b2s_compress(uint32_t h[8], const uint32_t m[16], uint64_t t, bool last) {
    uint32_t v[16];
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = B2S_IV[i]; }
    v[12] ^= (uint32_t)t;
    v[13] ^= (uint32_t)(t >> 32);
    if (last) v[14] = ~v[14];
#define ORC_G(a, b, c, d, x, y)                                                     \
    v[a] = v[a] + v[b] + (x); v[d] = b2s_rotr(v[d] ^ v[a], 16);                     \
    v[c] = v[c] + v[d];       v[b] = b2s_rotr(v[b] ^ v[c], 12);                     \
    v[a] = v[a] + v[b] + (y); v[d] = b2s_rotr(v[d] ^ v[a], 8);                      \
    v[c] = v[c] + v[d];       v[b] = b2s_rotr(v[b] ^ v[c], 7);
    for (int r = 0; r < 10; r++) {
        const uint8_t* s = B2S_SIGMA[r];
        ORC_G(0, 4, 8, 12, m[s[0]], m[s[1]]);
        ORC_G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        ORC_G(2, 6, 10, 14, m[s[4]], m[s[5]]);
        ORC_G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        ORC_G(0, 5, 10, 15, m[s[8]], m[s[9]]);
        ORC_G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        ORC_G(2, 7, 8, 13, m[s[12]], m[s[13]]);
        ORC_G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef ORC_G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}

struct Digest {
    uint8_t b[32];
    bool operator==(const Digest& o) const { return memcmp(b, o.b, 32) == 0; }
    bool operator!=(const Digest& o) const { return !(*this == o); }
};

// General byte-string BLAKE2s-256.
static inline Digest blake2s(const uint8_t* data, size_t len) {
    uint32_t h[8];
    for (int i = 0; i < 8; i++) h[i] = B2S_IV[i];
    h[0] ^= 0x01010020u;   // digest_length=32, key_length=0, fanout=1, depth=1
    size_t off = 0;
    uint32_t m[16];
    while (len - off > 64) {
        memcpy(m, data + off, 64);
        off += 64;
        b2s_compress(h, m, off, false);
    }
    uint8_t blk[64];
    memset(blk, 0, 64);
    memcpy(blk, data + off, len - off);
    memcpy(m, blk, 64);
    b2s_compress(h, m, len, true);   // empty input: one all-zero final block with t = 0
    Digest d;
    memcpy(d.b, h, 32);
    return d;
}

// hash_elements over base-field elements: 32-byte padded LE serialisation each (random.cairo:93-104).
static inline Digest hash_elements(const uint64_t* e, size_t n) {
    if (n == 0) return blake2s(nullptr, 0);
    uint32_t h[8];
    for (int i = 0; i < 8; i++) h[i] = B2S_IV[i];
    h[0] ^= 0x01010020u;
    size_t total = n * 32, done = 0;
    for (size_t i = 0; i < n; i += 2) {
        uint32_t m[16];
        memset(m, 0, sizeof m);
        m[0] = (uint32_t)e[i]; m[1] = (uint32_t)(e[i] >> 32);
        size_t take = 32;
        if (i + 1 < n) { m[8] = (uint32_t)e[i + 1]; m[9] = (uint32_t)(e[i + 1] >> 32); take = 64; }
        done += take;
        b2s_compress(h, m, done, done == total);
    }
    Digest d;
    memcpy(d.b, h, 32);
    return d;
}

static inline Digest merge(const Digest& a, const Digest& b) {   // random.cairo:330-342
    uint8_t buf[64];
    memcpy(buf, a.b, 32); memcpy(buf + 32, b.b, 32);
    return blake2s(buf, 64);
}
static inline Digest merge_with_int(const Digest& seed, uint64_t v) {   // random.cairo:67-91
    uint8_t buf[40];
    memcpy(buf, seed.b, 32);
    for (int i = 0; i < 8; i++) buf[32 + i] = (uint8_t)(v >> (8 * i));
    return blake2s(buf, 40);
}

}  // namespace orc
