// Host-to-device rate of this box's link, as the prover's hand-over sees it: hipMemcpyAsync from pinned (hipHostMalloc) and from pageable
// memory, one copy at a time and two copies on two streams, sizes from 16 MiB to 1 GiB, and the same while a kernel keeps the CUs busy.
// The result bounds every H2D-inclusive figure of bench.py (config.h2d_included): cells/s <= rate / 8 B.   VERDICT r4 item 2.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_h2d.hip -o tools/ubench_h2d ; run: tools/ubench_h2d [json-out]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(unsigned long long* out, int iters) {
    unsigned long long x = threadIdx.x + blockIdx.x;
    for (int i = 0; i < iters; i++) x = x * 6364136223846793005ull + 1442695040888963407ull;
    if (x == 42) out[0] = x;
}
int main(int argc, char** argv) {
    const size_t MAXB = (size_t)1 << 30;
    char *pin = nullptr, *pin2 = nullptr, *dev = nullptr, *dev2 = nullptr;
    CK(hipHostMalloc((void**)&pin, MAXB, hipHostMallocDefault)); CK(hipHostMalloc((void**)&pin2, MAXB, hipHostMallocDefault));
    CK(hipMalloc((void**)&dev, MAXB)); CK(hipMalloc((void**)&dev2, MAXB));
    char* page = (char*)malloc(MAXB);
    memset(pin, 1, MAXB); memset(pin2, 2, MAXB); memset(page, 3, MAXB);
    hipStream_t s0, s1, sk; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    unsigned long long* sink; CK(hipMalloc((void**)&sink, 8));
    auto rate = [&](const char* src0, const char* src1, size_t bytes, bool busy, bool d2h) {
        double best = 0;
        for (int r = 0; r < 4; r++) {
            if (busy) hipLaunchKernelGGL(spin, dim3(256 * 8), dim3(256), 0, sk, sink, 400000);
            CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
            const double t0 = now();
            if (d2h) CK(hipMemcpyAsync((void*)src0, dev, bytes, hipMemcpyDeviceToHost, s0));
            else CK(hipMemcpyAsync(dev, src0, bytes, hipMemcpyHostToDevice, s0));
            if (src1) CK(hipMemcpyAsync(dev2, src1, bytes, hipMemcpyHostToDevice, s1));
            CK(hipStreamSynchronize(s0)); if (src1) CK(hipStreamSynchronize(s1));
            const double dt = now() - t0;
            const double g = (src1 ? 2.0 : 1.0) * bytes / dt / 1e9;
            if (g > best) best = g;
            if (busy) CK(hipStreamSynchronize(sk));
        }
        return best;
    };
    std::vector<std::pair<const char*, double>> rows;
    char buf[16][96]; int nb = 0;
    auto put = [&](const char* fmt, size_t mib, double g) { snprintf(buf[nb], 96, fmt, mib); printf("%-58s %7.2f GB/s\n", buf[nb], g); rows.push_back({buf[nb], g}); nb++; };
    for (size_t mib : {16, 64, 256, 604, 1024}) put("h2d pinned, one copy, %zu MiB", mib, rate(pin, nullptr, mib << 20, false, false));
    put("h2d pinned, two copies on two streams, %zu MiB each", 256, rate(pin, pin2, (size_t)256 << 20, false, false));
    put("h2d pinned, one copy, %zu MiB, CUs busy", 604, rate(pin, nullptr, (size_t)604 << 20, true, false));
    put("h2d pageable, one copy, %zu MiB", 256, rate(page, nullptr, (size_t)256 << 20, false, false));
    put("d2h pinned, one copy, %zu MiB", 256, rate(pin, nullptr, (size_t)256 << 20, false, true));
    if (argc > 1) {
        FILE* f = fopen(argv[1], "w");
        fprintf(f, "{\n");
        for (size_t i = 0; i < rows.size(); i++) fprintf(f, " \"%s\": %.2f%s\n", rows[i].first, rows[i].second, i + 1 < rows.size() ? "," : "");
        fprintf(f, "}\n"); fclose(f);
    }
    return 0;
}
