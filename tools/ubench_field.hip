// Throughput of candidate Goldilocks add / sub / mul implementations on gfx950 (8 independent chains per thread).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_field.hip -o tools/ubench_field
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../aero_amd/csrc/gl_field.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef uint32_t u32; typedef uint64_t u64;
static constexpr u32 EPS32 = 0xFFFFFFFFu;
__device__ __forceinline__ u64 mk(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
namespace B {
__device__ __forceinline__ u64 add(u64 a, u64 b) {
    u32 c1, c2, c;
    u32 s0 = __builtin_addc((u32)a, (u32)b, 0u, &c); u32 s1 = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c, &c1);
    u32 t0 = __builtin_addc(s0, EPS32, 0u, &c); u32 t1 = __builtin_addc(s1, 0u, c, &c2);
    return (c1 | c2) ? mk(t0, t1) : mk(s0, s1);
}
__device__ __forceinline__ u64 sub(u64 a, u64 b) {
    u32 c, c1;
    u32 d0 = __builtin_subc((u32)a, (u32)b, 0u, &c); u32 d1 = __builtin_subc((u32)(a >> 32), (u32)(b >> 32), c, &c1);
    u32 t0 = __builtin_subc(d0, EPS32, 0u, &c); u32 t1 = __builtin_subc(d1, 0u, c, &c);
    return c1 ? mk(t0, t1) : mk(d0, d1);
}
__device__ __forceinline__ u64 mul(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 t = (u64)a0 * b0;
    u64 u = (u64)a0 * b1 + (t >> 32);
    u64 v = (u64)a1 * b0 + (u32)u;
    u64 w = (u64)a1 * b1 + ((u >> 32) + (v >> 32));
    u32 x0 = (u32)t, x1 = (u32)v, x2 = (u32)w, x3 = (u32)(w >> 32);
    u32 c, bo;
    u32 l0 = __builtin_subc(x0, x3, 0u, &c); u32 l1 = __builtin_subc(x1, 0u, c, &bo);
    u32 m = 0u - bo;
    l0 = __builtin_subc(l0, m, 0u, &c); l1 = __builtin_subc(l1, 0u, c, &c);
    u32 e0 = __builtin_subc(0u, x2, 0u, &c); u32 e1 = __builtin_subc(x2, 0u, c, &c);
    u32 ca;
    u32 r0 = __builtin_addc(l0, e0, 0u, &c); u32 r1 = __builtin_addc(l1, e1, c, &ca);
    u32 m2 = 0u - ca;
    r0 = __builtin_addc(r0, m2, 0u, &c); r1 = __builtin_addc(r1, 0u, c, &c);
    u32 c2;
    u32 t0 = __builtin_addc(r0, EPS32, 0u, &c); u32 t1 = __builtin_addc(r1, 0u, c, &c2);
    return c2 ? mk(t0, t1) : mk(r0, r1);
}
}
namespace D {   // product by the compiler (4 x v_mad_u64_u32), reduction + canonicalisation hand-written on VOP2 carry chains
__device__ __forceinline__ u64 mul(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t = (u64)a0 * b0;
    const u64 u = (u64)a0 * b1 + (t >> 32);
    const u64 v = (u64)a1 * b0 + (u32)u;
    const u64 w = (u64)a1 * b1 + ((u >> 32) + (v >> 32));
    const u32 x0 = (u32)t, x1 = (u32)v, x2 = (u32)w, x3 = (u32)(w >> 32), z = 0;
    u32 r0, r1, l0, l1, m, e0, e1, t0, t1;
    asm("v_sub_co_u32 %[l0], vcc, %[x0], %[x3]\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %[l1], vcc, 0, %[x1], vcc\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %[m], vcc, 0, %[z], vcc\n\t"
        "v_sub_co_u32 %[l0], vcc, %[l0], %[m]\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %[l1], vcc, 0, %[l1], vcc\n\t"
        "v_sub_co_u32 %[e0], vcc, 0, %[x2]\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %[e1], vcc, 0, %[x2], vcc\n\t"
        "v_add_co_u32 %[r0], vcc, %[l0], %[e0]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[r1], vcc, %[l1], %[e1], vcc\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %[m], vcc, 0, %[z], vcc\n\t"
        "v_add_co_u32 %[r0], vcc, %[r0], %[m]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[r1], vcc, 0, %[r1], vcc\n\t"
        "v_add_co_u32 %[t0], vcc, -1, %[r0]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[t1], vcc, 0, %[r1], vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %[r0], %[r0], %[t0], vcc\n\t"
        "v_cndmask_b32 %[r1], %[r1], %[t1], vcc"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [l0] "=&v"(l0), [l1] "=&v"(l1), [m] "=&v"(m), [e0] "=&v"(e0), [e1] "=&v"(e1), [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [z] "v"(z)
        : "vcc");
    return mk(r0, r1);
}
}
namespace ASM {   // round 5 (commit d4aaeb6, -DGL_ASM): add / sub / mul on VCC-only VOP2 carry chains, wait states spelled out. 13 % fewer VOP2-equivalents
                  // in the 32-point register transform, no faster there (profiles/r5_glasm_ubench.txt) and slower inside the NTT passes
                  // (profiles/r5_glasm_ab.txt): the opaque blocks cost the scheduler its interleaving and the kernels registers.
using gl::mk64;
__device__ __forceinline__ u64 add(u64 a, u64 b) {
    uint32_t s0, s1, t0, t1;
    uint64_t sv;
    asm("v_add_co_u32 %[s0], vcc, %[a0], %[b0]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[s1], vcc, %[a1], %[b1], vcc\n\t"
        "s_mov_b64 %[sv], vcc\n\t"
        "v_add_co_u32 %[t0], vcc, -1, %[s0]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[t1], vcc, 0, %[s1], vcc\n\t"
        "s_or_b64 vcc, vcc, %[sv]\n\t"
        "v_cndmask_b32 %[s0], %[s0], %[t0], vcc\n\t"
        "v_cndmask_b32 %[s1], %[s1], %[t1], vcc"
        : [s0] "=&v"(s0), [s1] "=&v"(s1), [t0] "=&v"(t0), [t1] "=&v"(t1), [sv] "=&s"(sv)
        : [a0] "v"((uint32_t)a), [a1] "v"((uint32_t)(a >> 32)), [b0] "v"((uint32_t)b), [b1] "v"((uint32_t)(b >> 32))
        : "vcc", "scc");      // s_or_b64 writes SCC
    return mk64(s0, s1);
}
__device__ __forceinline__ u64 sub(u64 a, u64 b) {
    uint32_t d0, d1, m;
    asm("v_sub_co_u32 %[d0], vcc, %[a0], %[b0]\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32 %[d1], vcc, %[a1], %[b1], vcc\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %[m], vcc, 0, %[z], vcc\n\t"
        "v_sub_co_u32 %[d0], vcc, %[d0], %[m]\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %[d1], vcc, 0, %[d1], vcc"
        : [d0] "=&v"(d0), [d1] "=&v"(d1), [m] "=&v"(m)
        : [a0] "v"((uint32_t)a), [a1] "v"((uint32_t)(a >> 32)), [b0] "v"((uint32_t)b), [b1] "v"((uint32_t)(b >> 32)), [z] "v"(0u)
        : "vcc");
    return mk64(d0, d1);
}
__device__ __forceinline__ u64 mul(u64 a, u64 b) {
    // product by the compiler (4 x v_mad_u64_u32), x0 + x1 2^32 + x2 (2^32 - 1) - x3 and the canonical form on VOP2 carry chains
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t t = (uint64_t)a0 * b0;
    const uint64_t u = (uint64_t)a0 * b1 + (t >> 32);
    const uint64_t v = (uint64_t)a1 * b0 + (uint32_t)u;
    const uint64_t w = (uint64_t)a1 * b1 + ((u >> 32) + (v >> 32));
    uint32_t r0, r1, l0, l1, m, e0, e1, t0, t1;
    uint64_t sc;
    asm("v_sub_co_u32 %[l0], vcc, %[x0], %[x3]\n\t"
        "v_sub_co_u32 %[e0], %[sc], 0, %[x2]\n\t"
        "s_nop 0\n\t"
        "v_subbrev_co_u32 %[l1], vcc, 0, %[x1], vcc\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %[m], vcc, 0, %[z], vcc\n\t"
        "v_sub_co_u32 %[l0], vcc, %[l0], %[m]\n\t"
        "v_subbrev_co_u32 %[e1], %[sc], 0, %[x2], %[sc]\n\t"
        "s_nop 0\n\t"
        "v_subbrev_co_u32 %[l1], vcc, 0, %[l1], vcc\n\t"
        "v_add_co_u32 %[r0], vcc, %[l0], %[e0]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[r1], vcc, %[l1], %[e1], vcc\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %[m], vcc, 0, %[z], vcc\n\t"
        "v_add_co_u32 %[r0], vcc, %[r0], %[m]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[r1], vcc, 0, %[r1], vcc\n\t"
        "v_add_co_u32 %[t0], vcc, -1, %[r0]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[t1], vcc, 0, %[r1], vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %[r0], %[r0], %[t0], vcc\n\t"
        "v_cndmask_b32 %[r1], %[r1], %[t1], vcc"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [l0] "=&v"(l0), [l1] "=&v"(l1), [m] "=&v"(m), [e0] "=&v"(e0), [e1] "=&v"(e1), [t0] "=&v"(t0), [t1] "=&v"(t1), [sc] "=&s"(sc)
        : [x0] "v"((uint32_t)t), [x1] "v"((uint32_t)v), [x2] "v"((uint32_t)w), [x3] "v"((uint32_t)(w >> 32)), [z] "v"(0u)
        : "vcc");
    return mk64(r0, r1);
}
}
namespace Cc {   // u64-typed with 128-bit product, reduction written on u64 with __builtin overflow
__device__ __forceinline__ u64 mul(u64 a, u64 b) {
    u64 lo = a * b, hi = __umul64hi(a, b);
    u64 hh = hi >> 32, hl = hi & 0xFFFFFFFFull;
    u64 t0; bool bo = __builtin_usubl_overflow(lo, hh, &t0);
    t0 -= bo ? 0xFFFFFFFFull : 0;
    u64 t1 = (hl << 32) - hl;
    u64 r; bool ca = __builtin_uaddl_overflow(t0, t1, &r);
    r += ca ? 0xFFFFFFFFull : 0;
    u64 t; bool c2 = __builtin_uaddl_overflow(r, 0xFFFFFFFFull, &t);
    return c2 ? t : r;
}
}
#define BENCH(NAME, MULF, ADDF, SUBF, KIND)                                                        \
__global__ __launch_bounds__(256) void NAME(u64* out, int iters) {                                 \
    u64 x[8], k = (blockIdx.x * 0xD1B54A32D192ED03ull + 3) % gl::P;                                \
    for (int u = 0; u < 8; u++) x[u] = (threadIdx.x * 0x9E3779B97F4A7C15ull + u) % gl::P;         \
    for (int i = 0; i < iters; i++) {                                                              \
        _Pragma("unroll") for (int r = 0; r < 4; r++)                                              \
        _Pragma("unroll") for (int u = 0; u < 8; u++) {                                            \
            if (KIND == 0) x[u] = MULF(x[u], k); if (KIND == 1) x[u] = ADDF(x[u], k); if (KIND == 2) x[u] = SUBF(x[u], k); } } \
    u64 r = 0; for (int u = 0; u < 8; u++) r ^= x[u];                                              \
    out[blockIdx.x * 256 + threadIdx.x] = r; }
BENCH(a_mul, gl::mul, gl::add, gl::sub, 0) BENCH(a_add, gl::mul, gl::add, gl::sub, 1) BENCH(a_sub, gl::mul, gl::add, gl::sub, 2)
BENCH(b_mul, B::mul, B::add, B::sub, 0) BENCH(b_add, B::mul, B::add, B::sub, 1) BENCH(b_sub, B::mul, B::add, B::sub, 2)
BENCH(c_mul, Cc::mul, B::add, B::sub, 0)
BENCH(d_mul, D::mul, B::add, B::sub, 0)
BENCH(e_mul, ASM::mul, ASM::add, ASM::sub, 0) BENCH(e_add, ASM::mul, ASM::add, ASM::sub, 1) BENCH(e_sub, ASM::mul, ASM::add, ASM::sub, 2)
__global__ void check(u64* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    u64 a = (i * 0x9E3779B97F4A7C15ull) % gl::P, b = ((i + 7) * 0xD1B54A32D192ED03ull) % gl::P;
    if (i < 4) { a = gl::P - 1 - i; b = gl::P - 1; } if (i == 5) { a = 0; } if (i == 6) { a = 0xFFFFFFFFull; b = 0xFFFFFFFF00000000ull; }
    bool ok = B::mul(a, b) == gl::mul(a, b) && B::add(a, b) == gl::add(a, b) && B::sub(a, b) == gl::sub(a, b) && B::sub(b, a) == gl::sub(b, a) && Cc::mul(a, b) == gl::mul(a, b) && D::mul(a, b) == gl::mul(a, b)
        && ASM::mul(a, b) == gl::mul(a, b) && ASM::add(a, b) == gl::add(a, b) && ASM::sub(a, b) == gl::sub(a, b) && ASM::sub(b, a) == gl::sub(b, a);
    u64 c3 = (a ^ (b >> 3)) % gl::P;
    ok = ok && B::mul(B::sub(a, b), c3) == gl::mul(gl::sub(a, b), c3) && B::sub(B::mul(a, b), c3) == gl::sub(gl::mul(a, b), c3) && B::sub(c3, B::mul(a, b)) == gl::sub(c3, gl::mul(a, b))
        && B::add(B::mul(B::sub(a, c3), b), B::mul(B::sub(b, c3), a)) == gl::add(gl::mul(gl::sub(a, c3), b), gl::mul(gl::sub(b, c3), a));
    if (!ok) atomicAdd((unsigned long long*)out, 1ull);
}
int main() {
    const int blocks = 256 * 16, iters = 128;
    u64* out; CK(hipMalloc(&out, blocks * 256 * 8));
    CK(hipMemset(out, 0, 8)); hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, out, 1 << 20); u64 bad; CK(hipMemcpy(&bad, out, 8, hipMemcpyDeviceToHost)); printf("mismatches vs gl::  : %llu\n", (unsigned long long)bad);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto kern) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters); CK(hipDeviceSynchronize());
        float best = 1e9;
        for (int r = 0; r < 3; r++) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
        printf("%-20s %8.3f ms  %9.1f Gop/s\n", name, best, iters * 32.0 * blocks * 256.0 / (best * 1e-3) / 1e9);
    };
    timeit("A mul (current)", a_mul); timeit("B mul (carry chain)", b_mul); timeit("C mul (u64 ovf)", c_mul); timeit("D mul (asm reduce)", d_mul); timeit("E mul (asm, interleaved e chain)", e_mul);
    timeit("A add (current)", a_add); timeit("B add (carry chain)", b_add);
    timeit("A sub (current)", a_sub); timeit("B sub (carry chain)", b_sub);
    timeit("E add (asm VCC chain)", e_add); timeit("E sub (asm VCC chain)", e_sub);
    return 0;
}
