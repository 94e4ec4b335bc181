// Host round trip on one stream: a kernel produces 32 bytes, the host reads them, derives 64 bytes, the next kernel consumes them.
//   A  hipMemcpyAsync D2H + hipStreamSynchronize, hipMemcpyAsync H2D (what the prover does at every transcript point)
//   B  the kernel stores to mapped pinned memory, hipStreamSynchronize, parameters by value in the kernel arguments
//   C  as B, but the host polls a sequence word in pinned memory instead of synchronising the stream
//   D  as A with hipDeviceScheduleSpin
//   E  the kernel stores to mapped pinned memory, hipStreamWriteValue32 of a sequence word behind it, the host polls the word; parameters
//      by hipMemcpyAsync H2D as in A (what the prover can do without touching its kernels' argument lists)
//   G  the kernel stores to mapped pinned memory, hipStreamSynchronize, parameters by hipMemcpyAsync H2D
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_roundtrip.hip -o tools/ubench_roundtrip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct P64 { uint64_t v[8]; };
__global__ void produce(uint64_t* out, const uint64_t* in) { if (threadIdx.x < 4) out[threadIdx.x] = in[threadIdx.x] * 3 + 1; }
__global__ void produce_val(uint64_t* out, P64 p) { if (threadIdx.x < 4) out[threadIdx.x] = p.v[threadIdx.x] * 3 + 1; }
__global__ void produce_host(uint64_t* out, volatile uint64_t* host, volatile uint32_t* flag, uint32_t seq, P64 p) {
    if (threadIdx.x < 4) { uint64_t v = p.v[threadIdx.x] * 3 + 1; out[threadIdx.x] = v; host[threadIdx.x] = v; }
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence_system(); *flag = seq; }
}
__global__ void produce_host_plain(uint64_t* out, volatile uint64_t* host, const uint64_t* in) {      // E / G: results straight to mapped pinned memory, no flag
    if (threadIdx.x < 4) { uint64_t v = in[threadIdx.x] * 3 + 1; out[threadIdx.x] = v; host[threadIdx.x] = v; }
}
__global__ void busy(uint64_t* x, int n) { uint64_t v = x[0]; for (int i = 0; i < n; i++) v = v * 6364136223846793005ull + 1; x[1] = v; }
int main(int argc, char** argv) {
    const int iters = 300;
    if (argc > 1 && argv[1][0] == 's') CK(hipSetDeviceFlags(hipDeviceScheduleSpin));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint64_t *d_out, *d_in, *h_out, *h_in; uint32_t* h_flag;
    CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&d_in, 64));
    CK(hipHostMalloc((void**)&h_out, 64, hipHostMallocDefault)); CK(hipHostMalloc((void**)&h_in, 64, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&h_flag, 64, hipHostMallocDefault));
    memset(h_in, 0, 64); *h_flag = 0;
    CK(hipMemcpy(d_in, h_in, 64, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 1; i <= iters; i++) {
                if (mode == 0) {
                    hipLaunchKernelGGL(produce, 1, 64, 0, s, d_out, d_in);
                    CK(hipMemcpyAsync(h_out, d_out, 32, hipMemcpyDeviceToHost, s));
                    CK(hipStreamSynchronize(s));
                    for (int k = 0; k < 8; k++) h_in[k] = h_out[k & 3] + k;
                    CK(hipMemcpyAsync(d_in, h_in, 64, hipMemcpyHostToDevice, s));
                } else if (mode == 1) {
                    P64 p; for (int k = 0; k < 8; k++) p.v[k] = h_out[k & 3] + k;
                    hipLaunchKernelGGL(produce_host, 1, 64, 0, s, d_out, (volatile uint64_t*)h_out, (volatile uint32_t*)h_flag, (uint32_t)i, p);
                    CK(hipStreamSynchronize(s));
                } else {
                    P64 p; for (int k = 0; k < 8; k++) p.v[k] = h_out[k & 3] + k;
                    hipLaunchKernelGGL(produce_host, 1, 64, 0, s, d_out, (volatile uint64_t*)h_out, (volatile uint32_t*)h_flag, (uint32_t)(i + 1000 * (rep + 1)), p);
                    while (*(volatile uint32_t*)h_flag != (uint32_t)(i + 1000 * (rep + 1))) { }
                }
            }
            CK(hipStreamSynchronize(s));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
            if (rep) printf("mode %c: %.2f us per round trip\n", "ABC"[mode], us);
        }
    }
    for (int mode = 0; mode < 2; mode++) {
        bool ok = true;
        for (int rep = 0; rep < 2 && ok; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 1; i <= iters; i++) {
                const uint32_t seq = (uint32_t)(i + 100000 * (rep + 1) + 1000000 * mode);
                hipLaunchKernelGGL(produce_host_plain, 1, 64, 0, s, d_out, (volatile uint64_t*)h_out, d_in);
                if (mode == 0) {
                    if (hipStreamWriteValue32(s, h_flag, seq, 0) != hipSuccess) { printf("mode E: hipStreamWriteValue32 refused\n"); (void)hipGetLastError(); ok = false; break; }
                    while (*(volatile uint32_t*)h_flag != seq) { }
                } else CK(hipStreamSynchronize(s));
                for (int k = 0; k < 8; k++) h_in[k] = h_out[k & 3] + k;
                CK(hipMemcpyAsync(d_in, h_in, 64, hipMemcpyHostToDevice, s));
            }
            CK(hipStreamSynchronize(s));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
            if (rep) printf("mode %c: %.2f us per round trip\n", "EG"[mode], us);
        }
    }
    // the same with a 50 us kernel in front (the wake-up after a longer wait)
    for (int mode = 0; mode < 3; mode += 2) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 1; i <= 100; i++) {
            hipLaunchKernelGGL(busy, 1, 1, 0, s, d_out + 4, 20000);
            if (mode == 0) {
                hipLaunchKernelGGL(produce, 1, 64, 0, s, d_out, d_in);
                CK(hipMemcpyAsync(h_out, d_out, 32, hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
                for (int k = 0; k < 8; k++) h_in[k] = h_out[k & 3] + k;
                CK(hipMemcpyAsync(d_in, h_in, 64, hipMemcpyHostToDevice, s));
            } else {
                P64 p; for (int k = 0; k < 8; k++) p.v[k] = h_out[k & 3] + k;
                hipLaunchKernelGGL(produce_host, 1, 64, 0, s, d_out, (volatile uint64_t*)h_out, (volatile uint32_t*)h_flag, (uint32_t)(i + 5000 + mode), p);
                while (*(volatile uint32_t*)h_flag != (uint32_t)(i + 5000 + mode)) { }
            }
        }
        CK(hipStreamSynchronize(s));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 100;
        printf("with a busy kernel in front, mode %c: %.2f us per iteration\n", "ABC"[mode], us);
    }
    return 0;
}
