// Experiment harness: variants of the fused leaf-hash + 3-level Merkle kernel on a 2-column, 2^23-row matrix.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bench_hash.hip -o tools/bench_hash
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../aero_amd/csrc/blake2s_hash.hpp"
using b2s::Digest;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void store_digest(Digest* dst, const Digest& d) {
    uint4* p = reinterpret_cast<uint4*>(dst);
    p[0] = make_uint4(d.w[0], d.w[1], d.w[2], d.w[3]); p[1] = make_uint4(d.w[4], d.w[5], d.w[6], d.w[7]);
}
__device__ __forceinline__ Digest load_digest(const Digest* src) {
    const uint4* p = reinterpret_cast<const uint4*>(src); uint4 a = p[0], b = p[1]; Digest d;
    d.w[0] = a.x; d.w[1] = a.y; d.w[2] = a.z; d.w[3] = a.w; d.w[4] = b.x; d.w[5] = b.y; d.w[6] = b.z; d.w[7] = b.w; return d;
}
__device__ __forceinline__ Digest leaf2(uint64_t e0, uint64_t e1) {
    b2s::State st; b2s::init(st); b2s::compress_elems(st, e0, e1, true, 64, true);
    Digest d;
#pragma unroll
    for (int i = 0; i < 8; i++) d.w[i] = st.h[i];
    return d;
}
template <int N> struct Slots {
    Digest d[N];
    __device__ __forceinline__ void put(int i, const Digest& v) {
#pragma unroll
        for (int s = 0; s < N; s++) { const bool hit = (s == i);
#pragma unroll
            for (int k = 0; k < 8; k++) d[s].w[k] = hit ? v.w[k] : d[s].w[k]; }
    }
    __device__ __forceinline__ Digest get(int i) const {
        Digest r = d[0];
#pragma unroll
        for (int s = 1; s < N; s++) { const bool hit = (s == i);
#pragma unroll
            for (int k = 0; k < 8; k++) r.w[k] = hit ? d[s].w[k] : r.w[k]; }
        return r;
    }
};
__device__ __forceinline__ void build3(Slots<8>& sl, Digest* nodes, size_t child_base, int min_store_h) {
#pragma unroll 1
    for (int it = 0; it < 7; it++) {
        const int h = it < 4 ? 1 : (it < 6 ? 2 : 3);
        const int p = it < 4 ? it : (it < 6 ? it - 4 : 0);
        Digest l = sl.get(2 * p), r = sl.get(2 * p + 1);
        Digest m = b2s::merge(l, r);
        sl.put(p, m);
        if (h >= min_store_h) store_digest(&nodes[(child_base >> h) + p], m);
    }
}
// V0: current product kernel (selects, direct 8-byte loads at 64-byte lane stride)
__global__ __launch_bounds__(256) void v0(const uint64_t* c0, const uint64_t* c1, Digest* nodes, size_t n) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; if (t >= n / 8) return;
    size_t first = t * 8; Slots<8> sl;
#pragma unroll 1
    for (int i = 0; i < 8; i++) sl.put(i, leaf2(c0[first + i], c1[first + i]));
    build3(sl, nodes, n + first, 3);
}
// V1: fully unrolled, no selects
__global__ __launch_bounds__(256) void v1(const uint64_t* c0, const uint64_t* c1, Digest* nodes, size_t n) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; if (t >= n / 8) return;
    size_t first = t * 8;
    Digest d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = leaf2(c0[first + i], c1[first + i]);
#pragma unroll
    for (int i = 0; i < 4; i++) d[i] = b2s::merge(d[2 * i], d[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 2; i++) d[i] = b2s::merge(d[2 * i], d[2 * i + 1]);
    store_digest(&nodes[(n + first) >> 3], b2s::merge(d[0], d[1]));
}
// V2: selects + preloaded rows with 16-byte loads
__global__ __launch_bounds__(256) void v2(const uint64_t* c0, const uint64_t* c1, Digest* nodes, size_t n) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; if (t >= n / 8) return;
    size_t first = t * 8; Slots<8> sl;
    uint64_t a[8], b[8];
    const ulonglong2* p0 = reinterpret_cast<const ulonglong2*>(c0 + first); const ulonglong2* p1 = reinterpret_cast<const ulonglong2*>(c1 + first);
#pragma unroll
    for (int i = 0; i < 4; i++) { ulonglong2 x = p0[i], y = p1[i]; a[2 * i] = x.x; a[2 * i + 1] = x.y; b[2 * i] = y.x; b[2 * i + 1] = y.y; }
#pragma unroll 1
    for (int i = 0; i < 8; i++) {
        uint64_t e0 = a[0], e1 = b[0];
#pragma unroll
        for (int s = 1; s < 8; s++) { e0 = (s == i) ? a[s] : e0; e1 = (s == i) ? b[s] : e1; }
        sl.put(i, leaf2(e0, e1));
    }
    build3(sl, nodes, n + first, 3);
}
// V3: plain local array (scratch)
__global__ __launch_bounds__(256) void v3(const uint64_t* c0, const uint64_t* c1, Digest* nodes, size_t n) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; if (t >= n / 8) return;
    size_t first = t * 8;
    Digest d[8];
#pragma unroll 1
    for (int i = 0; i < 8; i++) d[i] = leaf2(c0[first + i], c1[first + i]);
#pragma unroll 1
    for (int it = 0; it < 7; it++) {
        const int p = it < 4 ? it : (it < 6 ? it - 4 : 0);
        Digest m = b2s::merge(d[2 * p], d[2 * p + 1]);
        d[p] = m;
        if (it == 6) store_digest(&nodes[(n + first) >> 3], m);
    }
}
// V4: coalesced leaf pass only (lane <-> row), digests stored
__global__ __launch_bounds__(256) void v4(const uint64_t* c0, const uint64_t* c1, Digest* nodes, size_t n) {
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; if (j >= n) return;
    store_digest(&nodes[n + j], leaf2(c0[j], c1[j]));
}
// V5: 3 levels from stored digests (selects)
__global__ __launch_bounds__(256) void v5(Digest* nodes, size_t m) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; if (t >= m) return;
    size_t child_base = (m + t) * 8; Slots<8> sl;
#pragma unroll
    for (int i = 0; i < 8; i++) sl.d[i] = load_digest(&nodes[child_base + i]);
    build3(sl, nodes, child_base, 1);
}
// V6: 3 levels from stored digests, fully unrolled
__global__ __launch_bounds__(256) void v6(Digest* nodes, size_t m) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; if (t >= m) return;
    size_t cb = (m + t) * 8; Digest d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = load_digest(&nodes[cb + i]);
#pragma unroll
    for (int i = 0; i < 4; i++) { d[i] = b2s::merge(d[2 * i], d[2 * i + 1]); store_digest(&nodes[(cb >> 1) + i], d[i]); }
#pragma unroll
    for (int i = 0; i < 2; i++) { d[i] = b2s::merge(d[2 * i], d[2 * i + 1]); store_digest(&nodes[(cb >> 2) + i], d[i]); }
    store_digest(&nodes[cb >> 3], b2s::merge(d[0], d[1]));
}
// V7: single level
__global__ __launch_bounds__(256) void v7(Digest* nodes, size_t m) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; if (t >= m) return;
    size_t i = m + t;
    store_digest(&nodes[i], b2s::merge(load_digest(&nodes[2 * i]), load_digest(&nodes[2 * i + 1])));
}
// V8: 2 leaves per thread + 1 level, lane pairs rows (2t, 2t+1): 16-byte loads, no selects, 3 compress bodies
__global__ __launch_bounds__(256) void v8(const uint64_t* c0, const uint64_t* c1, Digest* nodes, size_t n) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; if (t >= n / 2) return;
    ulonglong2 x = reinterpret_cast<const ulonglong2*>(c0)[t], y = reinterpret_cast<const ulonglong2*>(c1)[t];
    Digest a = leaf2(x.x, y.x), b = leaf2(x.y, y.y);
    store_digest(&nodes[(n >> 1) + t], b2s::merge(a, b));
}

int main() {
    const size_t n = (size_t)1 << 23;
    uint64_t *c0, *c1; Digest* nodes;
    CK(hipMalloc(&c0, n * 8)); CK(hipMalloc(&c1, n * 8)); CK(hipMalloc(&nodes, 2 * n * 32));
    std::vector<uint64_t> h(n); for (size_t i = 0; i < n; i++) h[i] = i * 0x9E3779B97F4A7C15ull;
    CK(hipMemcpy(c0, h.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(c1, h.data(), n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch, double compress) {
        launch(); CK(hipDeviceSynchronize());
        float best = 1e9;
        for (int r = 0; r < 5; r++) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
        printf("%-44s %8.3f ms  %7.2f G compress/s\n", name, best, compress / (best * 1e-3) / 1e9);
    };
    double cl8 = n * (1.0 + 7.0 / 8);
    timeit("v0 leaf8 selects direct loads", [&] { hipLaunchKernelGGL(v0, dim3(n / 8 / 256), dim3(256), 0, 0, c0, c1, nodes, n); }, cl8);
    timeit("v1 leaf8 fully unrolled", [&] { hipLaunchKernelGGL(v1, dim3(n / 8 / 256), dim3(256), 0, 0, c0, c1, nodes, n); }, cl8);
    timeit("v2 leaf8 selects + 16B preload", [&] { hipLaunchKernelGGL(v2, dim3(n / 8 / 256), dim3(256), 0, 0, c0, c1, nodes, n); }, cl8);
    timeit("v3 leaf8 scratch array", [&] { hipLaunchKernelGGL(v3, dim3(n / 8 / 256), dim3(256), 0, 0, c0, c1, nodes, n); }, cl8);
    timeit("v4 leaf pass coalesced (stores leaves)", [&] { hipLaunchKernelGGL(v4, dim3(n / 256), dim3(256), 0, 0, c0, c1, nodes, n); }, (double)n);
    timeit("v5 up3 selects (from leaves)", [&] { hipLaunchKernelGGL(v5, dim3(n / 8 / 256), dim3(256), 0, 0, nodes, n / 8); }, n * 7.0 / 8);
    timeit("v6 up3 unrolled (from leaves)", [&] { hipLaunchKernelGGL(v6, dim3(n / 8 / 256), dim3(256), 0, 0, nodes, n / 8); }, n * 7.0 / 8);
    timeit("v7 single level (n/2 nodes)", [&] { hipLaunchKernelGGL(v7, dim3(n / 2 / 256), dim3(256), 0, 0, nodes, n / 2); }, n / 2.0);
    timeit("v8 leaf2 + 1 level, 16B loads", [&] { hipLaunchKernelGGL(v8, dim3(n / 2 / 256), dim3(256), 0, 0, c0, c1, nodes, n); }, n * 1.5);
    return 0;
}
