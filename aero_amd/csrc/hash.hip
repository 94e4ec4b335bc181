// BLAKE2s row hashing and Merkle construction for gfx950.
//
// Replaces the reference's hashing seam and tree build:
//   /root/reference/aero-sdk/miden-wasm/src/hashing_worker.rs:12-26   for row in rows: Blake2s_256::hash_elements(row)
//   /root/reference/aero-sdk/miden-wasm/src/proving_worker.rs:287-300 row gather `read_row_into` from the column-major LDE
//   /root/reference/aero-sdk/miden-wasm/src/proving_worker.rs:161-162 MerkleTree::new(trace_row_hashes)
//   FRI layer leaves: hash of `fold` transposed evaluations (mirror: src/stark_verifier/channel.cairo:102-133)
// Node convention: nodes[1] = root, children of i are 2i and 2i+1, leaves at nodes[n + j]
// (src/stark_verifier/channel.cairo:136-175).
//
// One thread owns one hash state (16-word working vector in VGPRs). Column-major input makes the row gather a
// coalesced read: lane j reads element j of each column. These kernels are integer-VALU bound (about 10^3 32-bit
// ops per compression while only 16 of the 64 hashed bytes per block come from HBM), not HBM bound.
#include "aero_internal.hpp"

namespace aero {

__device__ __forceinline__ void store_digest(Digest* dst, const b2s::State& s) {
    uint4* p = reinterpret_cast<uint4*>(dst);
    p[0] = make_uint4(s.h[0], s.h[1], s.h[2], s.h[3]);
    p[1] = make_uint4(s.h[4], s.h[5], s.h[6], s.h[7]);
}
__device__ __forceinline__ Digest load_digest(const Digest* src) {
    const uint4* p = reinterpret_cast<const uint4*>(src);
    uint4 a = p[0], b = p[1];
    Digest d;
    d.w[0] = a.x; d.w[1] = a.y; d.w[2] = a.z; d.w[3] = a.w; d.w[4] = b.x; d.w[5] = b.y; d.w[6] = b.z; d.w[7] = b.w;
    return d;
}

__global__ __launch_bounds__(256) void hash_rows_kernel(const uint64_t* __restrict__ cols, size_t col_stride, int ncols, size_t rows,
                                                          Digest* __restrict__ leaves) {
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= rows) return;
    b2s::State s;
    b2s::init(s);
    const uint32_t total = (uint32_t)ncols * 32;
    for (int c = 0; c < ncols; c += 2) {
        bool two = c + 1 < ncols;
        uint64_t e0 = cols[(size_t)c * col_stride + j];
        uint64_t e1 = two ? cols[(size_t)(c + 1) * col_stride + j] : 0;
        uint32_t t = (uint32_t)(two ? c + 2 : c + 1) * 32;
        b2s::compress_elems(s, e0, e1, two, t, t == total);
    }
    store_digest(&leaves[j], s);
}

// FRI rows: element q of row i (q < fold*deg) = comp[q % deg][i + (q / deg) * rows]
__global__ __launch_bounds__(256) void hash_fri_rows_kernel(const uint64_t* __restrict__ c0, const uint64_t* __restrict__ c1, int deg,
                                                              size_t rows, int fold, Digest* __restrict__ leaves) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    b2s::State s;
    b2s::init(s);
    const int nel = fold * deg;
    const uint32_t total = (uint32_t)nel * 32;
    for (int q = 0; q < nel; q += 2) {
        uint64_t e0, e1;
        if (deg == 1) {
            e0 = c0[i + (size_t)q * rows];
            e1 = c0[i + (size_t)(q + 1) * rows];
        } else {
            e0 = c0[i + (size_t)(q >> 1) * rows];
            e1 = c1[i + (size_t)(q >> 1) * rows];
        }
        uint32_t t = (uint32_t)(q + 2) * 32;
        b2s::compress_elems(s, e0, e1, true, t, t == total);
    }
    store_digest(&leaves[i], s);
}

// one tree level: nodes[i] = merge(nodes[2i], nodes[2i+1]) for i in [m, 2m)
__global__ __launch_bounds__(256) void merkle_level_kernel(Digest* nodes, size_t m) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= m) return;
    size_t i = m + t;
    Digest l = load_digest(&nodes[2 * i]), r = load_digest(&nodes[2 * i + 1]);
    Digest d = b2s::merge(l, r);
    uint4* p = reinterpret_cast<uint4*>(&nodes[i]);
    p[0] = make_uint4(d.w[0], d.w[1], d.w[2], d.w[3]);
    p[1] = make_uint4(d.w[4], d.w[5], d.w[6], d.w[7]);
}

// top of the tree in one workgroup: level `m` (<= 256 nodes) down to the root, staged through LDS
__global__ __launch_bounds__(256) void merkle_top_kernel(Digest* nodes, int m) {
    __shared__ Digest buf[512];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * m; i += 256) buf[i] = nodes[2 * m + i];   // children of level m: nodes[2m .. 4m)
    __syncthreads();
    // buf[0 .. 2w) holds the children of the current level (width w); results overwrite buf[0 .. w) after a sync
    for (int w = m; w >= 1; w >>= 1) {
        Digest d;
        if (tid < w) d = b2s::merge(buf[2 * tid], buf[2 * tid + 1]);
        __syncthreads();
        if (tid < w) { buf[tid] = d; nodes[w + tid] = d; }
        __syncthreads();
    }
}

void Context::hash_rows(const uint64_t* cols, size_t col_stride, int ncols, size_t rows, Digest* leaves) {
    if (ncols < 1) fail("hash_rows: empty rows");
    AERO_LAUNCH(this, "hash_rows_kernel", rows * ((size_t)ncols * 8 + 32), hash_rows_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, cols, col_stride, ncols, rows, leaves);
    check_launch("hash_rows");
}

void Context::hash_fri_rows(const uint64_t* const comp[2], int deg, size_t rows, int fold, Digest* leaves) {
    if ((fold * deg) & 1) fail("hash_fri_rows: odd element count");
    AERO_LAUNCH(this, "hash_fri_rows_kernel", rows * ((size_t)fold * deg * 8 + 32), hash_fri_rows_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, comp[0], comp[1], deg, rows, fold, leaves);
    check_launch("hash_fri_rows");
}

void Context::merkle_build(Digest* nodes, size_t n) {
    if (n < 2 || (n & (n - 1))) fail("merkle_build: leaf count must be a power of two >= 2");
    size_t m = n / 2;
    for (; m > 256; m >>= 1)
        AERO_LAUNCH(this, "merkle_level_kernel", m * 96, merkle_level_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, nodes, m);
    AERO_LAUNCH(this, "merkle_top_kernel", m * 128, merkle_top_kernel, dim3(1), dim3(256), 0, nodes, (int)m);
    check_launch("merkle_build");
}

}  // namespace aero
