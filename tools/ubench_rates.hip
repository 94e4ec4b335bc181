// Clean issue-rate micro-benchmark: 16 independent dependency chains per thread, one instruction kind per kernel
// (inline asm so the compiler cannot fold or re-associate). Reports wave-instructions/s as lane-ops/s.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_rates.hip -o tools/ubench_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define CHAIN16(OPSTR)                                                               \
    _Pragma("unroll") for (int u = 0; u < 16; u++) asm volatile(OPSTR : "+v"(x[u]) : "v"(k0), "v"(k1));

#define DEFK(NAME, OPSTR)                                                            \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters) {          \
        uint32_t x[16];                                                              \
        for (int u = 0; u < 16; u++) x[u] = threadIdx.x * 17 + u;                    \
        uint32_t k0 = blockIdx.x + 3, k1 = threadIdx.x | 1;                          \
        for (int i = 0; i < iters; i++) { CHAIN16(OPSTR) CHAIN16(OPSTR) }            \
        uint32_t r = 0;                                                              \
        for (int u = 0; u < 16; u++) r ^= x[u];                                      \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                     \
    }

DEFK(k_xor, "v_xor_b32 %0, %0, %1")
DEFK(k_add, "v_add_u32 %0, %0, %1")
DEFK(k_alignbit, "v_alignbit_b32 %0, %0, %0, 7")
DEFK(k_alignbit2, "v_alignbit_b32 %0, %0, %1, 7")
DEFK(k_add3, "v_add3_u32 %0, %0, %1, %2")
DEFK(k_perm, "v_perm_b32 %0, %0, %0, %1")
DEFK(k_lshl_or, "v_lshl_or_b32 %0, %0, 7, %1")
DEFK(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
DEFK(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
DEFK(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEFK(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
DEFK(k_lshr, "v_lshrrev_b32 %0, 7, %0")
DEFK(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEFK(k_mov_sdwa, "v_mov_b32_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0")
DEFK(k_xor_sdwa, "v_xor_b32_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0")
DEFK(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
DEFK(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")

// 64-bit destination ops
#define DEFK64(NAME, OPSTR)                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters) {          \
        uint64_t x[8];                                                               \
        for (int u = 0; u < 8; u++) x[u] = threadIdx.x * 17 + u;                     \
        uint32_t k0 = blockIdx.x + 3, k1 = threadIdx.x | 1;                          \
        for (int i = 0; i < iters; i++) {                                            \
            _Pragma("unroll") for (int r = 0; r < 4; r++)                            \
            _Pragma("unroll") for (int u = 0; u < 8; u++) asm volatile(OPSTR : "+v"(x[u]) : "v"(k0), "v"(k1)); } \
        uint64_t r = 0;                                                              \
        for (int u = 0; u < 8; u++) r ^= x[u];                                       \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(r ^ (r >> 32));            \
    }
DEFK64(k_mad64, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
DEFK64(k_lshl64, "v_lshlrev_b64 %0, 7, %0")
DEFK64(k_lshladd64, "v_lshl_add_u64 %0, %0, 3, %0")

int main() {
    const int blocks = 256 * 16, iters = 256;
    uint32_t* out; CK(hipMalloc(&out, blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto kern, double ops_per_thread) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters); CK(hipDeviceSynchronize());
        float best = 1e9;
        for (int r = 0; r < 3; r++) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
        printf("%-28s %8.3f ms  %9.2f Tlane-op/s\n", name, best, ops_per_thread * blocks * 256.0 / (best * 1e-3) / 1e12);
    };
    double n32 = iters * 32.0;
    timeit("v_xor_b32", k_xor, n32); timeit("v_add_u32", k_add, n32); timeit("v_lshrrev_b32", k_lshr, n32);
    timeit("v_alignbit_b32 (x,x,7)", k_alignbit, n32); timeit("v_alignbit_b32 (x,k,7)", k_alignbit2, n32);
    timeit("v_lshl_or_b32", k_lshl_or, n32); timeit("v_and_or_b32", k_and_or, n32); timeit("v_bfi_b32", k_bfi, n32);
    timeit("v_cndmask_b32", k_cndmask, n32);
    timeit("v_mov_b32_sdwa", k_mov_sdwa, n32); timeit("v_xor_b32_sdwa", k_xor_sdwa, n32);
    timeit("v_mul_lo_u32", k_mul_lo, n32); timeit("v_mul_hi_u32", k_mul_hi, n32);
    timeit("v_mul_u32_u24", k_mul_u24, n32); timeit("v_mad_u32_u24", k_mad_u32_u24, n32);
    timeit("v_mad_u64_u32", k_mad64, iters * 32.0); timeit("v_lshlrev_b64", k_lshl64, iters * 32.0); timeit("v_lshl_add_u64", k_lshladd64, iters * 32.0);
    return 0;
}
