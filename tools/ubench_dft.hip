// What would lazy reduction buy the NTT's register transforms? The register phase of ntt_fwd_first_pass_8 - one
// table multiplication per element, the 32-point transform with shift twiddles (dft_small.hpp), the pass-boundary twiddle as a
// geometric progression (2 multiplications per element) - and the 64-point transform of ntt_fwd_strided_reg6x2's column phase, with no
// memory traffic at all, built twice: with the product's canonical gl::add / gl::sub, and with -DGL_LAZY_ADD_UNSAFE, which replaces
// gl::add by the cheapest conceivable lazy form (one wrap correction, no >= p select, NO second correction: values drift out of
// [0, p) and a double wrap is not caught - the results are NOT field elements, the build only bounds the instruction-count gain
// from above). gl::sub already costs what a lazy sub would (borrow -> + p, 5 issue slots).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I aero_amd/csrc tools/ubench_dft.hip -o /tmp/ubench_dft [-DGL_LAZY_ADD_UNSAFE]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "dft_small.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef uint64_t u64;
using namespace aero;

__global__ __launch_bounds__(256, 3) void phase_b32(u64* out, int iters) {
    u64 y[32], k = (blockIdx.x * 0xD1B54A32D192ED03ull + 3) % gl::P, step = (threadIdx.x * 0x9E3779B97F4A7C15ull + 11) % gl::P;
#pragma unroll
    for (int u = 0; u < 32; u++) y[u] = (threadIdx.x * 0x9E3779B97F4A7C15ull + u * 0x2545F4914F6CDD1Dull) % gl::P;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 1; i < 32; i++) y[i] = gl::mul(y[i], k);          // stands for the table twiddle
        dft_dit_reg<5>(y);
        u64 cur = k;
#pragma unroll
        for (int i = 0; i < 32; i++) { y[i] = gl::mul(y[i], cur); cur = gl::mul(cur, step); }
        k = cur;
    }
    u64 r = 0;
#pragma unroll
    for (int u = 0; u < 32; u++) r ^= y[u];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ __launch_bounds__(256, 3) void dft32_only(u64* out, int iters) {
    u64 y[32];
#pragma unroll
    for (int u = 0; u < 32; u++) y[u] = (threadIdx.x * 0x9E3779B97F4A7C15ull + u * 0x2545F4914F6CDD1Dull + blockIdx.x) % gl::P;
    for (int it = 0; it < iters; it++) dft_dit_reg<5>(y);
    u64 r = 0;
#pragma unroll
    for (int u = 0; u < 32; u++) r ^= y[u];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ __launch_bounds__(256, 3) void dft8_shift(u64* out, int iters) {      // the radix-8 butterflies of the LDS rounds
    u64 y[8];
#pragma unroll
    for (int u = 0; u < 8; u++) y[u] = (threadIdx.x * 0x9E3779B97F4A7C15ull + u * 0x2545F4914F6CDD1Dull + blockIdx.x) % gl::P;
    for (int it = 0; it < 4 * iters; it++) dft_dit<3>(y);
    u64 r = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) r ^= y[u];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
    const int blocks = 256 * 12, iters = 64;
    u64* out; CK(hipMalloc(&out, blocks * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#ifdef GL_ASM
    printf("variant: hand-written VOP2 carry chains (GL_ASM)\n");
#elif defined(GL_LAZY_ADD_UNSAFE)
    printf("variant: gl::add WITHOUT canonicalisation (upper bound on the gain, results are not field elements)\n");
#else
    printf("variant: the product's canonical gl::add / gl::sub\n");
#endif
    auto timeit = [&](const char* name, auto kern, double elems_per_thread_iter) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters); CK(hipDeviceSynchronize());
        float best = 1e9;
        for (int r = 0; r < 3; r++) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
        static u64 host[256 * 12 * 256];
        CK(hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost));
        u64 sum = 0; for (size_t i = 0; i < sizeof(host) / 8; i++) sum = sum * 0x9E3779B97F4A7C15ull + host[i];
        printf("%-44s %8.3f ms  %9.1f Gelem/s  checksum %016llx\n", name, best, iters * elems_per_thread_iter * blocks * 256.0 / (best * 1e-3) / 1e9, (unsigned long long)sum);
    };
    timeit("phase B (mul + dft32 + 2 mul) per element", phase_b32, 32.0);
    timeit("dft32 shift-twiddle stages only", dft32_only, 32.0);
    timeit("dft8 (radix-8 butterfly of the LDS rounds)", dft8_shift, 32.0);
    return 0;
}
