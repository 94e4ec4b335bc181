// Calibration of rocprofv3's FETCH_SIZE counter on gfx950 for the access patterns of this repo's kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o tools/ubench_fetch
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- tools/ubench_fetch
// Every kernel reads from a buffer of 2^27 u64 (1 GiB: larger than the L2s and the 256 MiB Infinity Cache) and touches a KNOWN
// number of distinct bytes; FETCH_SIZE (KiB) * 1024 / that number is the counter's scale for the pattern:
//   read16_contig   16 B per lane, lanes contiguous (the pattern the guide's x2 correction was calibrated on)
//   read8_contig     8 B per lane, lanes contiguous (column reads of every per-row kernel here)
//   read8_stride64   8 B per lane, 64-byte lane stride (every 8th row: deep_kernel)      useful = 1/8 of the span
//   read8_stride32   8 B per lane, 32-byte lane stride (every 4th row: fib_constraints)  useful = 1/4 of the span
//   read8_blocks8    a lane reads 8 consecutive u64 with 8 separate loads (merkle_leaf8's row ownership)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void read16_contig(const ulonglong2* p, size_t n16, uint64_t* sink) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t acc = 0;
    for (; i < n16; i += (size_t)gridDim.x * 256) { ulonglong2 v = p[i]; acc += v.x ^ v.y; }
    if (acc == 0x1234567) *sink = acc;
}
__global__ void read8_contig(const uint64_t* p, size_t n, uint64_t* sink) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 0x1234567) *sink = acc;
}
template <int STRIDE> __global__ void read8_strided(const uint64_t* p, size_t n, uint64_t* sink) {   // elements i * STRIDE
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t acc = 0;
    for (; i * STRIDE < n; i += (size_t)gridDim.x * 256) acc += p[i * STRIDE];
    if (acc == 0x1234567) *sink = acc;
}
__global__ void read8_blocks8(const uint64_t* p, size_t n, uint64_t* sink) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t acc = 0;
    for (; t * 8 < n; t += (size_t)gridDim.x * 256) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc += p[t * 8 + k];
    }
    if (acc == 0x1234567) *sink = acc;
}

int main() {
    const size_t n = (size_t)1 << 27;
    uint64_t *buf, *sink;
    CK(hipMalloc(&buf, n * 8));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(buf, 1, n * 8));
    const dim3 grid(256 * 16), block(256);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(read16_contig, grid, block, 0, 0, (const ulonglong2*)buf, n / 2, sink);
        hipLaunchKernelGGL(read8_contig, grid, block, 0, 0, buf, n, sink);
        hipLaunchKernelGGL(read8_strided<8>, grid, block, 0, 0, buf, n, sink);
        hipLaunchKernelGGL(read8_strided<4>, grid, block, 0, 0, buf, n, sink);
        hipLaunchKernelGGL(read8_blocks8, grid, block, 0, 0, buf, n, sink);
    }
    CK(hipDeviceSynchronize());
    printf("span bytes per kernel: %zu; useful bytes: read16_contig %zu, read8_contig %zu, read8_strided<8> %zu, read8_strided<4> %zu, read8_blocks8 %zu\n",
           n * 8, n * 8, n * 8, n, n * 2, n * 8);
    return 0;
}
