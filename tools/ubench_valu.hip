// Micro-benchmark: integer VALU issue rates and in-register BLAKE2s / Goldilocks-multiply throughput on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_valu.hip -o /tmp/ubench_valu   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../aero_amd/csrc/blake2s_hash.hpp"
#include "../aero_amd/csrc/gl_field.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int OP> __global__ __launch_bounds__(256) void k_int(uint32_t* out, int iters) {
    uint32_t a = threadIdx.x, b = blockIdx.x * 7 + 1, c = a ^ b, d = a + b, e = a * 3, f = b * 5, g = a + 11, h = b + 13;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == 0) { a ^= b; c ^= d; e ^= f; g ^= h; b ^= c; d ^= e; f ^= g; h ^= a; }
            if (OP == 1) { a += b; c += d; e += f; g += h; b += c; d += e; f += g; h += a; }
            if (OP == 2) { a = __builtin_amdgcn_alignbit(a, a, 7); c = __builtin_amdgcn_alignbit(c, c, 12); e = __builtin_amdgcn_alignbit(e, e, 8); g = __builtin_amdgcn_alignbit(g, g, 16);
                           b = __builtin_amdgcn_alignbit(b, b, 7); d = __builtin_amdgcn_alignbit(d, d, 12); f = __builtin_amdgcn_alignbit(f, f, 8); h = __builtin_amdgcn_alignbit(h, h, 16); }
            if (OP == 3) { a = a + b + c; c = c + d + e; e = e + f + g; g = g + h + a; b = b + c + d; d = d + e + f; f = f + g + h; h = h + a + b; }
            if (OP == 4) { a = __builtin_amdgcn_perm(a, a, 0x01000302); c = __builtin_amdgcn_perm(c, c, 0x00030201); e = __builtin_amdgcn_perm(e, e, 0x01000302); g = __builtin_amdgcn_perm(g, g, 0x00030201);
                           b = __builtin_amdgcn_perm(b, b, 0x01000302); d = __builtin_amdgcn_perm(d, d, 0x00030201); f = __builtin_amdgcn_perm(f, f, 0x01000302); h = __builtin_amdgcn_perm(h, h, 0x00030201); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
__global__ __launch_bounds__(256) void k_blake(uint32_t* out, int iters) {
    b2s::Digest d;
    for (int i = 0; i < 8; i++) d.w[i] = threadIdx.x * 31 + i + blockIdx.x;
    for (int i = 0; i < iters; i++) d = b2s::merge(d, d);
    uint32_t x = 0;
    for (int i = 0; i < 8; i++) x ^= d.w[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_blake_elems(uint32_t* out, int iters) {
    uint64_t e0 = threadIdx.x * 31 + blockIdx.x, e1 = e0 * 77 + 5;
    for (int i = 0; i < iters; i++) {
        b2s::State s; b2s::init(s);
        b2s::compress_elems(s, e0, e1, true, 64, true);
        e0 = ((uint64_t)s.h[1] << 32) | s.h[0]; e1 = ((uint64_t)s.h[3] << 32) | s.h[2];
    }
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(e0 ^ e1);
}
__global__ __launch_bounds__(256) void k_glmul(uint64_t* out, int iters) {
    uint64_t a = threadIdx.x * 0x9E3779B97F4A7C15ull + 1, b = blockIdx.x * 0xD1B54A32D192ED03ull + 3, c = a ^ 0x1234567, d = b ^ 0x7654321;
    a %= gl::P; b %= gl::P; c %= gl::P; d %= gl::P;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) { a = gl::mul(a, b); c = gl::mul(c, d); b = gl::mul(b, c); d = gl::mul(d, a); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d;
}
__global__ __launch_bounds__(256) void k_gladd(uint64_t* out, int iters) {
    uint64_t a = threadIdx.x * 0x9E3779B97F4A7C15ull + 1, b = blockIdx.x * 0xD1B54A32D192ED03ull + 3, c = a ^ 0x1234567, d = b ^ 0x7654321;
    a %= gl::P; b %= gl::P; c %= gl::P; d %= gl::P;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) { a = gl::add(a, b); c = gl::sub(c, d); b = gl::add(b, c); d = gl::sub(d, a); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d;
}

int main() {
    const int blocks = 256 * 16, iters = 512;
    uint32_t* out; CK(hipMalloc(&out, blocks * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch, double ops_per_thread) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double tot = ops_per_thread * blocks * 256.0;
        printf("%-22s %8.3f ms  %10.2f Gop/s\n", name, ms, tot / (ms * 1e-3) / 1e9);
        return 0;
    };
    timeit("v_xor_b32", [&] { hipLaunchKernelGGL((k_int<0>), dim3(blocks), dim3(256), 0, 0, out, iters); }, iters * 16.0 * 8);
    timeit("v_add_u32", [&] { hipLaunchKernelGGL((k_int<1>), dim3(blocks), dim3(256), 0, 0, out, iters); }, iters * 16.0 * 8);
    timeit("v_alignbit_b32", [&] { hipLaunchKernelGGL((k_int<2>), dim3(blocks), dim3(256), 0, 0, out, iters); }, iters * 16.0 * 8);
    timeit("v_add3_u32", [&] { hipLaunchKernelGGL((k_int<3>), dim3(blocks), dim3(256), 0, 0, out, iters); }, iters * 16.0 * 8);
    timeit("v_perm_b32", [&] { hipLaunchKernelGGL((k_int<4>), dim3(blocks), dim3(256), 0, 0, out, iters); }, iters * 16.0 * 8);
    timeit("blake2s merge (compr/s)", [&] { hipLaunchKernelGGL(k_blake, dim3(blocks), dim3(256), 0, 0, out, 256); }, 256.0);
    timeit("blake2s elems (compr/s)", [&] { hipLaunchKernelGGL(k_blake_elems, dim3(blocks), dim3(256), 0, 0, out, 256); }, 256.0);
    timeit("goldilocks mul", [&] { hipLaunchKernelGGL(k_glmul, dim3(blocks), dim3(256), 0, 0, (uint64_t*)out, iters); }, iters * 8.0 * 4);
    timeit("goldilocks add/sub", [&] { hipLaunchKernelGGL(k_gladd, dim3(blocks), dim3(256), 0, 0, (uint64_t*)out, iters); }, iters * 8.0 * 4);
    return 0;
}
