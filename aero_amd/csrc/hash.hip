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
// These kernels are integer-VALU bound (about 1336 VALU issue slots per compression: 656 full-rate xor/add plus 320
// rotations and 20 three-input adds that issue at half rate; only 16 of the 64 hashed bytes of a leaf block come from
// HBM). The design therefore minimises everything that is NOT a compression:
//   * one thread owns one hash state (16-word working vector in VGPRs, fully unrolled rounds);
//   * large narrow matrices (<= 4 columns, >= 2^18 rows) use the FUSED kernel: a thread hashes 8 consecutive rows and
//     builds the 3 tree levels above them in registers (no idle lanes at any level, one launch);
//   * everything else hashes one row per lane (coalesced column reads) and then builds the tree;
//   * upper levels are built 3 levels per launch (8 child digests -> 7 nodes per thread), the top 9 levels by one
//     workgroup staged through LDS: a compression has ~1.2 us of serial latency, so small trees are bound by
//     (levels x latency + launch boundaries), and 3 levels per launch removes two of every three boundaries;
//   * trees of large narrow matrices (the fused kernel's) do not store their three lowest levels (skip = 3: 7/8 of the digest writes and
//     of the tree's memory, 0.1 ms per 2^23-leaf tree) - the openings recompute the few low nodes they need from the committed matrix
//     (merkle_recompute_kernel, 8 lanes per node). Prover::low_level_skip, on by default (prover.hip: commit_to_rows).
// Measured on MI355X (tools/bench_hash.hip, 2^23 x 2 matrix): every variant lands at 30-36 G compressions/s against
// 40-42 G/s for compressions that never leave registers (tools/ubench_valu.hip).
#include <type_traits>

#include "aero_internal.hpp"
#include "dft_small.hpp"

namespace aero {

__device__ __forceinline__ void store_digest(Digest* dst, const Digest& d) {
    uint4* p = reinterpret_cast<uint4*>(dst);
    p[0] = make_uint4(d.w[0], d.w[1], d.w[2], d.w[3]);
    p[1] = make_uint4(d.w[4], d.w[5], d.w[6], d.w[7]);
}
__device__ __forceinline__ Digest load_digest(const Digest* src) {
    const uint4* p = reinterpret_cast<const uint4*>(src);
    uint4 a = p[0], b = p[1];
    Digest d;
    d.w[0] = a.x; d.w[1] = a.y; d.w[2] = a.z; d.w[3] = a.w; d.w[4] = b.x; d.w[5] = b.y; d.w[6] = b.z; d.w[7] = b.w;
    return d;
}
__device__ __forceinline__ Digest state_digest(const b2s::State& s) {
    Digest d;
#pragma unroll
    for (int i = 0; i < 8; i++) d.w[i] = s.h[i];
    return d;
}

// Workgroups are dispatched round-robin over the 8 XCDs. XCD = 1 hands XCD x the x-th contiguous eighth of a launch's blocks (as the NTT
// passes do, ntt.hip: xcd_tile), so that the rows one XCD has in flight are neighbours in memory; 0 = the dispatch order.
template <int XCD> __device__ __forceinline__ size_t hash_block(uint32_t bid, uint32_t nblocks) {
    if (XCD && !(nblocks & 7u)) return (size_t)(bid & 7u) * (nblocks >> 3) + (bid >> 3);
    return bid;
}

// ---- leaf sources ---------------------------------------------------------------------------------
__device__ __forceinline__ Digest leaf_digest(const RowSrc& s, size_t j) {
    b2s::State st;
    b2s::init(st);
    const uint32_t total = (uint32_t)s.ncols * 32;
    for (int c = 0; c < s.ncols; c += 2) {
        bool two = c + 1 < s.ncols;
        uint64_t e0 = s.cols[(size_t)c * s.stride + j];
        uint64_t e1 = two ? s.cols[(size_t)(c + 1) * s.stride + j] : 0;
        uint32_t t = (uint32_t)(two ? c + 2 : c + 1) * 32;
        b2s::compress_elems(st, e0, e1, two, t, t == total);
    }
    return state_digest(st);
}
// rows of a worker message (row-major, each with its own length; an element >= p stands for its residue like Felt::new reads it;
// a row without elements is BLAKE2s of the empty string, as hash_elements(&[]) is)
__device__ __forceinline__ Digest leaf_digest(const MsgSrc& s, size_t i) {
    b2s::State st;
    b2s::init(st);
    const uint64_t* row = s.words + s.offs[i];
    const uint32_t n = (uint32_t)row[0];
    const uint32_t total = n * 32;
    if (n == 0) b2s::compress_elems(st, 0, 0, false, 0, true);
    for (uint32_t c = 0; c < n; c += 2) {
        const bool two = c + 1 < n;
        uint64_t e0 = row[1 + c], e1 = two ? row[2 + c] : 0;
        e0 = e0 >= gl::P ? e0 - gl::P : e0;
        e1 = e1 >= gl::P ? e1 - gl::P : e1;
        const uint32_t t = (two ? c + 2 : c + 1) * 32;
        b2s::compress_elems(st, e0, e1, two, t, t == total);
    }
    return state_digest(st);
}
// FRI rows: element q of row i (q < fold*deg) = comp[q % deg][i + (q / deg) * rows]
__device__ __forceinline__ Digest leaf_digest(const FriSrc& s, size_t i) {
    b2s::State st;
    b2s::init(st);
    const int nel = s.fold * s.deg;
    const uint32_t total = (uint32_t)nel * 32;
    for (int q = 0; q < nel; q += 2) {
        uint64_t e0, e1;
        if (s.deg == 1) {
            e0 = s.c0[i + (size_t)q * s.rows];
            e1 = s.c0[i + (size_t)(q + 1) * s.rows];
        } else {
            e0 = s.c0[i + (size_t)(q >> 1) * s.rows];
            e1 = s.c1[i + (size_t)(q >> 1) * s.rows];
        }
        uint32_t t = (uint32_t)(q + 2) * 32;
        b2s::compress_elems(st, e0, e1, true, t, t == total);
    }
    return state_digest(st);
}

// Straight-line leaf hash for a compile-time column count (no loop: lets the 8-leaf kernel keep its digests in VGPRs;
// with the runtime-width loop the compiler declined the outer unroll and spilled the digest array to scratch, which
// showed up as 270 MB of extra HBM writes per 2^23-row tree in the PMC counters).
template <int NC> __device__ __forceinline__ Digest leaf_digest_fixed(const RowSrc& s, size_t j) {
    b2s::State st;
    b2s::init(st);
#pragma unroll
    for (int c = 0; c < NC; c += 2) {
        const bool two = c + 1 < NC;
        const uint64_t e0 = s.cols[(size_t)c * s.stride + j];
        const uint64_t e1 = two ? s.cols[(size_t)(c + 1) * s.stride + j] : 0;
        const uint32_t t = (uint32_t)(two ? c + 2 : c + 1) * 32;
        b2s::compress_elems(st, e0, e1, two, t, t == (uint32_t)NC * 32);
    }
    return state_digest(st);
}

// Builds the 3 levels above 8 digests (children left to right, heap index of child 0 = child_base). The node at height
// h above the children and position p inside this subtree goes to nodes[(child_base >> h) + p]; only heights >=
// min_store_h are written. Fully unrolled: 7 compression bodies, every digest stays in VGPRs.
__device__ __forceinline__ void build3(Digest (&d)[8], Digest* nodes, size_t child_base, int min_store_h) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        d[i] = b2s::merge(d[2 * i], d[2 * i + 1]);
        if (min_store_h <= 1) store_digest(&nodes[(child_base >> 1) + i], d[i]);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        d[i] = b2s::merge(d[2 * i], d[2 * i + 1]);
        if (min_store_h <= 2) store_digest(&nodes[(child_base >> 2) + i], d[i]);
    }
    store_digest(&nodes[child_base >> 3], b2s::merge(d[0], d[1]));
}

// Fused: thread t hashes leaves [8t, 8t+8) and builds the 3 levels above them. `skip` = number of lowest levels that
// are not stored (0: store everything incl. leaves; 3: store only the subtree roots at height 3). Used for large
// narrow matrices only: it divides the thread count by 8, which makes small trees latency-bound.
template <class Src> __global__ __launch_bounds__(256) void merkle_leaf8_kernel(Src src, Digest* nodes, size_t n, int skip) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n / 8) return;
    const size_t first = t * 8;
    Digest d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        d[i] = leaf_digest(src, first + i);
        if (skip == 0) store_digest(&nodes[n + first + i], d[i]);
    }
    build3(d, nodes, n + first, skip > 1 ? skip : 1);
}
// The 8 leaf hashes of a thread as straight-line code by construction: with two compressions per leaf (3 or 4 columns) the
// compiler declines `#pragma unroll` on the leaf loop (23 compression bodies) and the digest array goes to scratch.
template <int NC, int I> struct LeafUnroll {
    static __device__ __forceinline__ void run(const uint64_t (&e)[NC][8], Digest (&d)[8], Digest* nodes, size_t n, size_t first, int skip) {
        b2s::State st;
        b2s::init(st);
#pragma unroll
        for (int c = 0; c < NC; c += 2) {
            const bool two = c + 1 < NC;
            const uint32_t t = (uint32_t)(two ? c + 2 : c + 1) * 32;
            b2s::compress_elems(st, e[c][I], two ? e[c + 1 < NC ? c + 1 : c][I] : 0, two, t, t == (uint32_t)NC * 32);
        }
        d[I] = state_digest(st);
        if (skip == 0) store_digest(&nodes[n + first + I], d[I]);
        if constexpr (I + 1 < 8) LeafUnroll<NC, I + 1>::run(e, d, nodes, n, first, skip);
    }
};
// A thread's 8 rows are 64 contiguous bytes per column: they are fetched up front with four 16-byte loads per column, so every
// sector that comes in from HBM is consumed at once. (With the loads left next to the compression that uses them the PMC
// counters showed 229 MB fetched per 2^23-row tree for 134 MB of data: by the time a lane came back for the next element of a
// line - one 960-instruction compression later - the line had often left the L2.)
template <int NC, int XCD = 0> __global__ __launch_bounds__(256) void merkle_leaf8_rows_kernel(RowSrc src, Digest* nodes, size_t n, int skip) {
    const size_t t = hash_block<XCD>(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (t >= n / 8) return;
    const size_t first = t * 8;
    uint64_t e[NC][8];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const ulonglong2* p = reinterpret_cast<const ulonglong2*>(src.cols + (size_t)c * src.stride + first);
#pragma unroll
        for (int k = 0; k < 4; k++) { const ulonglong2 v = p[k]; e[c][2 * k] = v.x; e[c][2 * k + 1] = v.y; }
    }
    Digest d[8];
    LeafUnroll<NC, 0>::run(e, d, nodes, n, first, skip);
    build3(d, nodes, n + first, skip > 1 ? skip : 1);
}

// 3 levels per launch from a stored level: thread t owns node m + t and its 8 descendants three levels below.
// Three levels per launch also cut the serial latency chain of small trees (one launch boundary per 3 levels).
__global__ __launch_bounds__(256) void merkle_up3_kernel(Digest* nodes, size_t m) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= m) return;
    const size_t child_base = (m + t) * 8;
    Digest d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = load_digest(&nodes[child_base + i]);
    build3(d, nodes, child_base, 1);
}

// Same, for the subtree a rank of a sharded proof builds from exchanged leaf digests: the leaf level holds 2^log_parts
// pieces (one per source rank, in arrival order) instead of the interleaved natural order, i.e. leaf u sits in slot
// n + (u mod parts) * (n / parts) + u / parts. Reading through that permutation saves a reordering pass over the leaves.
__global__ __launch_bounds__(256) void merkle_up3_parts_kernel(Digest* nodes, size_t n, int log_parts) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n / 8) return;
    const size_t piece = n >> log_parts, pmask = ((size_t)1 << log_parts) - 1;
    Digest d[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const size_t u = t * 8 + i;
        d[i] = load_digest(&nodes[n + (u & pmask) * piece + (u >> log_parts)]);
    }
    build3(d, nodes, n + t * 8, 1);
}

// one plain row-hash pass (wide matrices, and the C-ABI hashing seam): lane <-> row, coalesced column reads
template <class Src> __global__ __launch_bounds__(256) void hash_rows_kernel(Src src, size_t rows, Digest* leaves) {
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= rows) return;
    store_digest(&leaves[j], leaf_digest(src, j));
}

// L <= 9 levels in one launch, one workgroup per subtree: workgroup b owns node r = m + b and its 2^L descendants L
// levels below (heap indices (r << L) + j), staged through LDS. Lane utilisation is poor (2^L - 1 compressions on
// L * 2^(L-1) lane slots), so this is used only where a level has too few nodes to fill the chip anyway: there the cost
// is the serial latency of one compression per level (~2.4 us with one wave per SIMD) and one launch instead of L.
// reseed with the root, then draw one element of E (deg components); same retry rule as the host coin (prover.hip HostCoin)
__device__ __forceinline__ void coin_step(const CoinStep& cs, const Digest& root) {
    if (cs.root_out) *cs.root_out = root;
    const Digest seed = b2s::merge(*cs.seed_io, root);
    *cs.seed_io = seed;
    if (cs.seed_out) *cs.seed_out = seed;
    for (uint64_t ctr = 1; ctr < 1000; ctr++) {
        const Digest d = b2s::merge_with_int(seed, ctr);
        const uint64_t v0 = (uint64_t)d.w[0] | ((uint64_t)d.w[1] << 32), v1 = (uint64_t)d.w[2] | ((uint64_t)d.w[3] << 32);
        if (v0 < gl::P && (cs.deg == 1 || v1 < gl::P)) { cs.alpha_out[0] = v0; if (cs.deg > 1) cs.alpha_out[1] = v1; return; }
    }
    cs.alpha_out[0] = 0;
    if (cs.deg > 1) cs.alpha_out[1] = 0;   // unreachable in practice (the host replay of the transcript would fail the same way)
}
__global__ __launch_bounds__(256) void merkle_multi_kernel(Digest* nodes, size_t m, int L, CoinStep cs) {
    __shared__ Digest buf[512];
    const int tid = threadIdx.x;
    const size_t r = m + blockIdx.x;
    const int nchild = 1 << L;
    for (int i = tid; i < nchild; i += 256) buf[i] = load_digest(&nodes[(r << L) + i]);
    __syncthreads();
    for (int d = L - 1; d >= 0; d--) {
        const int w = 1 << d;
        Digest v;
        if (tid < w) v = b2s::merge(buf[2 * tid], buf[2 * tid + 1]);
        __syncthreads();
        if (tid < w) { buf[tid] = v; store_digest(&nodes[(r << d) + tid], v); }
        __syncthreads();
    }
    if (m == 1 && tid == 0) {
        if (cs.seed_io) coin_step(cs, buf[0]);
        else if (cs.root_out) *cs.root_out = buf[0];       // no coin step: only the root, once more, where the host reads it
        if (cs.flag_out) { __threadfence_system(); *(volatile uint32_t*)cs.flag_out = cs.flag_seq; }      // everything above is thread 0's own
    }
}

// Same subtree build with FOUR lanes per compression (a quad holds one column of the 4 x 4 BLAKE2s state each: lane j has
// v[j], v[4+j], v[8+j], v[12+j]; the diagonal step rotates rows 1..3 across the quad with DPP quad_perm moves). A level then
// costs ~1 us instead of the ~2.4 us of a whole compression on one lane - this kernel is latency-bound by construction (see
// above), so shortening the dependent instruction chain is what counts. Message words are read from the LDS copy of the two
// children through a per-lane, per-round index table held in registers.
__constant__ uint8_t QUAD_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
#define QUAD_ROT1(x) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(x), 0x39, 0xf, 0xf, true))   // lane i <- lane (i + 1) & 3
#define QUAD_ROT2(x) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(x), 0x4E, 0xf, 0xf, true))   // lane i <- lane (i + 2) & 3
#define QUAD_ROT3(x) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(x), 0x93, 0xf, 0xf, true))   // lane i <- lane (i + 3) & 3
#define QUAD_G(a, b, c, d, x, y)                          \
    a = a + b + (x); d = b2s::rotr(d ^ a, 16);            \
    c = c + d;       b = b2s::rotr(b ^ c, 12);            \
    a = a + b + (y); d = b2s::rotr(d ^ a, 8);             \
    c = c + d;       b = b2s::rotr(b ^ c, 7);
// BLAKE2s(left || right) of the 16 message words at msg[0..16) (LDS), computed by the 4 lanes of a quad; lane j returns words
// j (lo) and 4 + j (hi) of the digest. Every lane of the quad must call it (DPP moves), idle quads may pass any valid msg.
__device__ __forceinline__ void quad_merge(const uint32_t* msg, int j, const uint32_t (&pk)[10], uint32_t& lo, uint32_t& hi, uint32_t tlen = 64u) {
    const uint32_t iv_lo[4] = {b2s::IV0, b2s::IV1, b2s::IV2, b2s::IV3}, iv_hi[4] = {b2s::IV4, b2s::IV5, b2s::IV6, b2s::IV7};
    const uint32_t h_lo = (j == 0 ? b2s::IV0 ^ b2s::PARAM0 : j == 1 ? iv_lo[1] : j == 2 ? iv_lo[2] : iv_lo[3]);
    const uint32_t h_hi = (j == 0 ? iv_hi[0] : j == 1 ? iv_hi[1] : j == 2 ? iv_hi[2] : iv_hi[3]);
    uint32_t a = h_lo, b = h_hi;
    uint32_t c = (j == 0 ? iv_lo[0] : j == 1 ? iv_lo[1] : j == 2 ? iv_lo[2] : iv_lo[3]);
    uint32_t d = (j == 0 ? iv_hi[0] ^ tlen : j == 1 ? iv_hi[1] : j == 2 ? ~iv_hi[2] : iv_hi[3]);   // t = tlen bytes (one block), last block
    // the 4 message words of each round come from LDS through this lane's packed index table; none of the 40 reads depends
    // on the G chain, so they are all in flight before it starts
    uint32_t mw[10][4];
#pragma unroll
    for (int r = 0; r < 10; r++) {
#pragma unroll
        for (int k = 0; k < 4; k++) mw[r][k] = msg[(pk[r] >> (4 * k)) & 15u];
    }
#pragma unroll
    for (int r = 0; r < 10; r++) {
        QUAD_G(a, b, c, d, mw[r][0], mw[r][1])
        b = QUAD_ROT1(b); c = QUAD_ROT2(c); d = QUAD_ROT3(d);
        QUAD_G(a, b, c, d, mw[r][2], mw[r][3])
        b = QUAD_ROT3(b); c = QUAD_ROT2(c); d = QUAD_ROT1(d);
    }
    lo = h_lo ^ a ^ c;
    hi = h_hi ^ b ^ d;
}
// The transcript step of a FRI commitment (reseed with the root, draw the folding challenge) with quad compressions: every thread
// of the workgroup runs it in lockstep on the same 24 LDS words (the quads all compute the same digests, thread 0..3 store) - a
// lone lane needs 2.4 us per compression and the step is 2 of them on the critical path of every layer.
// scratch: 24 words of LDS, words [0, 8) = the root on entry. Returns the challenge (both components) to every thread.
__device__ __forceinline__ void coin_step_quad(Digest* seed_io, int deg, uint32_t* scratch, int tid, int j, const uint32_t (&pk)[10],
                                               uint64_t& a0, uint64_t& a1, Digest* seed_out = nullptr) {
    const uint32_t r_lo = scratch[j], r_hi = scratch[4 + j];
    const uint32_t s_lo = seed_io->w[j], s_hi = seed_io->w[4 + j];
    __syncthreads();
    if (tid < 4) { scratch[j] = s_lo; scratch[4 + j] = s_hi; scratch[8 + j] = r_lo; scratch[12 + j] = r_hi; }
    __syncthreads();
    uint32_t lo, hi;
    quad_merge(scratch, j, pk, lo, hi);                       // seed <- BLAKE2s(seed || root)
    __syncthreads();
    if (tid < 4) {
        scratch[j] = lo; scratch[4 + j] = hi; scratch[8 + j] = 0; scratch[12 + j] = 0;
        seed_io->w[j] = lo; seed_io->w[4 + j] = hi;
        if (seed_out) { seed_out->w[j] = lo; seed_out->w[4 + j] = hi; }
    }
    a0 = a1 = 0;
    for (uint32_t ctr = 1; ctr < 1000; ctr++) {               // draw: first 8 (16) bytes of BLAKE2s(seed || LE64(ctr)), retried while >= p
        if (tid == 0) scratch[8] = ctr;
        __syncthreads();
        quad_merge(scratch, j, pk, lo, hi, 40u);
        if (tid < 4) scratch[16 + j] = lo;
        __syncthreads();
        const uint64_t v0 = (uint64_t)scratch[16] | ((uint64_t)scratch[17] << 32), v1 = (uint64_t)scratch[18] | ((uint64_t)scratch[19] << 32);
        if (v0 < gl::P && (deg == 1 || v1 < gl::P)) { a0 = v0; a1 = deg > 1 ? v1 : 0; break; }
        __syncthreads();
    }
}
// Levels with more than 64 nodes use one lane per node (256 lanes already fill the 4 SIMDs of the CU: more lanes per node
// would only add instructions); from 64 nodes down a quad per node shortens the chain.
__global__ __launch_bounds__(256) void merkle_multi_quad_kernel(Digest* nodes, size_t m, int L, CoinStep cs) {
    __shared__ __attribute__((aligned(16))) Digest buf[512];
    const int tid = threadIdx.x, q = tid >> 2, j = tid & 3;
    const size_t r = m + blockIdx.x;
    const int nchild = 1 << L;
    for (int i = tid; i < nchild; i += 256) buf[i] = load_digest(&nodes[(r << L) + i]);
    uint32_t pk[10];
#pragma unroll
    for (int rd = 0; rd < 10; rd++)
        pk[rd] = (uint32_t)QUAD_SIGMA[rd][2 * j] | ((uint32_t)QUAD_SIGMA[rd][2 * j + 1] << 4) | ((uint32_t)QUAD_SIGMA[rd][8 + 2 * j] << 8) |
                 ((uint32_t)QUAD_SIGMA[rd][8 + 2 * j + 1] << 12);
    __syncthreads();
    uint32_t* words = reinterpret_cast<uint32_t*>(buf);
    for (int d = L - 1; d >= 0; d--) {
        const int w = 1 << d;
        if (w > 64) {
            Digest v;
            if (tid < w) v = b2s::merge(buf[2 * tid], buf[2 * tid + 1]);
            __syncthreads();
            if (tid < w) { buf[tid] = v; store_digest(&nodes[(r << d) + tid], v); }
            __syncthreads();
        } else {
            const bool live = q < w;
            uint32_t lo, hi;
            quad_merge(words + (live ? 16 * q : 0), j, pk, lo, hi);
            __syncthreads();
            if (live) {
                words[8 * q + j] = lo; words[8 * q + 4 + j] = hi;
                uint32_t* dst = reinterpret_cast<uint32_t*>(&nodes[(r << d) + q]);
                dst[j] = lo; dst[4 + j] = hi;
            }
            __syncthreads();
        }
    }
    if (m == 1 && tid == 0 && cs.root_out) store_digest(cs.root_out, buf[0]);      // also without a coin step: the root, once more, where the host reads it (mapped memory)
    if (cs.seed_io && m == 1) {       // uniform over the (single) workgroup
        uint64_t a0, a1;
        coin_step_quad(cs.seed_io, cs.deg, words, tid, j, pk, a0, a1, cs.seed_out);
        if (tid == 0) { cs.alpha_out[0] = a0; if (cs.deg > 1) cs.alpha_out[1] = a1; }
    }
    if (m == 1 && cs.flag_out) {      // uniform: the host's completion word, behind everything this workgroup stored for the host
        __threadfence_system();
        __syncthreads();
        if (tid == 0) *(volatile uint32_t*)cs.flag_out = cs.flag_seq;
    }
}

// ------------------------------------------------------------------------------------------------
// FRI tail (FriTailArgs): one workgroup of 512 threads, the current layer's evaluations and the tree level under construction
// live in LDS; global memory only receives what the opening phase reads later (evaluations, tree nodes, roots).
template <int LOGF> __global__ __launch_bounds__(512) void fri_tail_kernel(FriTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t tail_lds[];
    constexpr int FD = 1 << LOGF;
    const int tid = threadIdx.x, q = tid >> 2, j = tid & 3;
    const int deg = a.deg;
    uint64_t* v0 = tail_lds;
    uint64_t* v1 = tail_lds + (deg > 1 ? a.dom0 : 0);
    Digest* dig = reinterpret_cast<Digest*>(tail_lds + (size_t)deg * a.dom0);     // up to 512 digests (+ 1 spare for single-leaf layers)
    __shared__ uint64_t s_alpha[2];
    for (uint32_t i = tid; i < a.dom0 * (uint32_t)deg; i += 512) tail_lds[i] = a.vals0[i];
    uint32_t pk[10];
#pragma unroll
    for (int rd = 0; rd < 10; rd++)
        pk[rd] = (uint32_t)QUAD_SIGMA[rd][2 * j] | ((uint32_t)QUAD_SIGMA[rd][2 * j + 1] << 4) | ((uint32_t)QUAD_SIGMA[rd][8 + 2 * j] << 8) |
                 ((uint32_t)QUAD_SIGMA[rd][8 + 2 * j + 1] << 12);
    __syncthreads();
    uint32_t dom = a.dom0;
    uint64_t w_inv = a.w_inv0;
    uint32_t* words = reinterpret_cast<uint32_t*>(dig);
    for (int L = 0; L < a.n_layers; L++) {
        const uint32_t rows = dom >> LOGF;
        Digest* nodes = a.nodes[L];
        // leaves: row t = (v[t + j rows])_j, each value deg components
        const FriSrc src{v0, v1, deg, rows, FD};
        for (uint32_t t = tid; t < rows; t += 512) {
            const Digest d = leaf_digest(src, t);
            dig[t] = d;
            store_digest(&nodes[rows > 1 ? rows + t : 1], d);
        }
        __syncthreads();
        // tree: one lane per node while a level has more than 64 nodes, a quad per node below that (merkle_multi_quad_kernel)
        for (uint32_t w = rows >> 1; w >= 1; w >>= 1) {
            if (w > 64) {
                Digest v;
                if ((uint32_t)tid < w) v = b2s::merge(dig[2 * tid], dig[2 * tid + 1]);
                __syncthreads();
                if ((uint32_t)tid < w) { dig[tid] = v; store_digest(&nodes[w + tid], v); }
                __syncthreads();
            } else {
                const bool live = (uint32_t)q < w;
                uint32_t lo, hi;
                quad_merge(words + (live ? 16 * q : 0), j, pk, lo, hi);
                __syncthreads();
                if (live) {
                    words[8 * q + j] = lo; words[8 * q + 4 + j] = hi;
                    uint32_t* dst = reinterpret_cast<uint32_t*>(&nodes[w + q]);
                    dst[j] = lo; dst[4 + j] = hi;
                }
                __syncthreads();
            }
        }
        // transcript: reseed with the root, draw the folding challenge
        {
            if (tid == 0) store_digest(&a.roots_out[L], dig[0]);
            uint64_t a0, a1;
            coin_step_quad(a.seed_io, deg, words, tid, j, pk, a0, a1, a.seed_out);     // words [0, 8) = the root; the level buffer is free again
            if (tid == 0) {
                s_alpha[0] = a0; s_alpha[1] = a1;
                a.alphas_out[(size_t)L * deg] = a0;
                if (deg > 1) a.alphas_out[(size_t)L * deg + 1] = a1;
            }
        }
        __syncthreads();
        if (L + 1 == a.n_layers) break;
        // fold (stark.hip fri_fold_fft_kernel): inverse DFT of the row in registers, Horner in alpha / x_t, x_t = offset * w_dom^t
        uint64_t r0 = 0, r1 = 0;
        const bool act = (uint32_t)tid < rows;
        if (act) {
            uint64_t y0[FD], y1[FD];
#pragma unroll
            for (int k = 0; k < FD; k++) { y0[k] = v0[tid + k * rows]; y1[k] = deg > 1 ? v1[tid + k * rows] : 0; }
            dft_dif_inv<LOGF>(y0);
            if (deg > 1) dft_dif_inv<LOGF>(y1);
            const uint64_t xinv = gl::mul(a.gen_inv, gl::pow(w_inv, (uint64_t)tid));
            if (deg == 1) {
                const uint64_t r = gl::mul(s_alpha[0], xinv);
                uint64_t acc = 0;
#pragma unroll
                for (int k = FD - 1; k >= 0; k--) acc = gl::add(gl::mul(acc, r), y0[(int)gl::bitrev((uint32_t)k, LOGF)]);
                r0 = gl::mul(acc, a.fold_inv);
            } else {
                const gl::E2 r = gl::mulb(gl::E2{s_alpha[0], s_alpha[1]}, xinv);
                gl::E2 acc{0, 0};
#pragma unroll
                for (int k = FD - 1; k >= 0; k--) {
                    const int p = (int)gl::bitrev((uint32_t)k, LOGF);
                    acc = gl::add(gl::mul(acc, r), gl::E2{y0[p], y1[p]});
                }
                acc = gl::mulb(acc, a.fold_inv);
                r0 = acc.a0; r1 = acc.a1;
            }
        }
        __syncthreads();          // every row has been read: the next layer may overwrite the buffer
        // layout [deg][dom] for every layer: the next layer's second component sits right behind its first
        uint64_t* nv1 = v0 + rows;
        if (act) {
            v0[tid] = r0;
            a.vals_out[L][tid] = r0;
            if (deg > 1) { nv1[tid] = r1; a.vals_out[L][rows + tid] = r1; }
        }
        v1 = deg > 1 ? nv1 : v0;
        __syncthreads();
        dom = rows;
        uint64_t wn = w_inv;
#pragma unroll
        for (int k = 0; k < LOGF; k++) wn = gl::sqr(wn);
        w_inv = wn;
    }
    if (a.flag_out) {             // the host's completion word, behind every root / seed this workgroup stored for it (Context::wait_flag)
        __threadfence_system();
        __syncthreads();
        if (tid == 0) *(volatile uint32_t*)a.flag_out = a.flag_seq;
    }
}
void Context::fri_tail(const FriTailArgs& a, int fold) {
    if (a.dom0 > (uint32_t)FRI_TAIL_MAX_DOM || a.dom0 / (uint32_t)fold > (uint32_t)FRI_TAIL_MAX_ROWS || a.n_layers < 1 || a.n_layers > FRI_TAIL_MAX_LAYERS)
        fail("fri_tail: bad shape", ST_INTERNAL);
    const size_t lds = (size_t)a.deg * a.dom0 * 8 + 513 * sizeof(Digest);
    if (!fri_tail_attr_set) {   // per context (= per device, per host thread): up to 2 x 4096 evaluations + 513 digests, above the 64 KiB default of dynamic LDS
        const int cap = 2 * FRI_TAIL_MAX_DOM * 8 + 513 * (int)sizeof(Digest);
        AERO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fri_tail_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        AERO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fri_tail_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        AERO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fri_tail_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        fri_tail_attr_set = true;
    }
    if (fold == 8) AERO_LAUNCH(this, "fri_tail_kernel", 0, fri_tail_kernel<3>, dim3(1), dim3(512), lds, a);
    else if (fold == 4) AERO_LAUNCH(this, "fri_tail_kernel", 0, fri_tail_kernel<2>, dim3(1), dim3(512), lds, a);
    else if (fold == 2) AERO_LAUNCH(this, "fri_tail_kernel", 0, fri_tail_kernel<1>, dim3(1), dim3(512), lds, a);
    else fail("fri_tail: folding factor must be 2, 4 or 8", ST_INTERNAL);
    check_launch("fri_tail");
}

// Recompute unstored low nodes for openings: out[q] = digest of heap node idx[q] (height h < 3 above the leaves).
// 8 lanes cooperate on one node: each lane hashes one leaf of the (at most 8-leaf) subtree, partners are fetched with
// wave shuffles, so the latency is one leaf hash plus h merges instead of 2^(h+1) - 1 serial compressions.
template <class Src> __global__ __launch_bounds__(256) void merkle_recompute_kernel(Src src, size_t n, const uint64_t* idx, int count, Digest* out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int q = t >> 3, lane8 = t & 7;
    const bool live = q < count;
    uint64_t node = live ? idx[q] : n;
    int h = 0;
    while ((node << h) < n) h++;              // node << h lands in [n, 2n); h <= 3
    const size_t first = (node << h) - n;
    const int cnt = 1 << h;
    Digest d = leaf_digest(src, first + (lane8 < cnt ? lane8 : 0));
    for (int lvl = 0; lvl < 3; lvl++) {       // uniform trip count: every lane takes part in the shuffles
        Digest o;
#pragma unroll
        for (int k = 0; k < 8; k++) o.w[k] = __shfl_xor((int)d.w[k], 1 << lvl);
        if (lvl < h && ((lane8 >> lvl) & 1) == 0) d = b2s::merge(d, o);
    }
    if (live && lane8 == 0) out[q] = d;
}

// Every opened value and tree node of a proof in ONE launch (round 5): the first blocks gather through the host-built address list (as
// gather_addr_kernel in stark.hip does), the blocks behind them recompute the unstored low nodes of up to OPEN_MAX_JOBS trees of row
// matrices (as merkle_recompute_kernel does, 8 lanes per node). The address list and the node indices are READ from mapped pinned memory and
// the value block is WRITTEN to mapped pinned memory: no upload, no download, one launch instead of three to four - the opening phase of one
// proof alone was a copy, a 4 us gather and two 12 us recomputations in a row, then a copy back (47 us, AERO_QUERY_TIMING).
__global__ __launch_bounds__(256) void openings_kernel(OpeningArgs a) {
    if (blockIdx.x < a.gather_blocks) {
        const uint32_t t = blockIdx.x * 256 + threadIdx.x;
        if (t < a.n_u64) {
            const uint64_t* p = reinterpret_cast<const uint64_t*>(a.addr[t]);
            a.out[t] = p ? *p : 0;
        } else if (t < a.n_u64 + a.n_dig) {
            const uint32_t i = t - a.n_u64;
            const ulonglong2* p = reinterpret_cast<const ulonglong2*>(a.addr[t]);
            ulonglong2 x = make_ulonglong2(0, 0), y = x;
            if (p) { x = p[0]; y = p[1]; }
            uint64_t* o = a.out + a.n_u64 + 4 * (size_t)i;
            o[0] = x.x; o[1] = x.y; o[2] = y.x; o[3] = y.y;
        }
        return;
    }
    uint32_t b = blockIdx.x - a.gather_blocks;
    int job = 0;
    while (job + 1 < a.n_jobs && b >= a.jobs[job].blocks) { b -= a.jobs[job].blocks; job++; }
    const OpeningArgs::Job& J = a.jobs[job];
    const int t = (int)b * 256 + (int)threadIdx.x;
    const int q = t >> 3, lane8 = t & 7;
    const bool live = q < J.count;
    const size_t n = J.n;
    uint64_t node = live ? J.idx[q] : n;
    int h = 0;
    while ((node << h) < n) h++;              // node << h lands in [n, 2n); h <= 3
    const size_t first = (node << h) - n;
    const int cnt = 1 << h;
    Digest d = leaf_digest(J.src, first + (lane8 < cnt ? lane8 : 0));
    for (int lvl = 0; lvl < 3; lvl++) {       // uniform trip count: every lane takes part in the shuffles
        Digest o;
#pragma unroll
        for (int k = 0; k < 8; k++) o.w[k] = __shfl_xor((int)d.w[k], 1 << lvl);
        if (lvl < h && ((lane8 >> lvl) & 1) == 0) d = b2s::merge(d, o);
    }
    if (live && lane8 == 0) store_digest(&J.out[q], d);
}
void Context::openings(const OpeningArgs& a) {
    uint32_t blocks = a.gather_blocks;
    for (int j = 0; j < a.n_jobs; j++) blocks += a.jobs[j].blocks;
    if (!blocks) return;
    AERO_LAUNCH(this, "openings_kernel", 0, openings_kernel, dim3(blocks), dim3(256), 0, a);
    check_launch("openings");
}

// ------------------------------------------------------------------------------------------------
void Context::hash_rows(const uint64_t* cols, size_t col_stride, int ncols, size_t rows, Digest* leaves) {
    if (ncols < 1) fail("hash_rows: empty rows");
    RowSrc src{cols, col_stride, ncols};
    // Round 6 measured three other forms of this kernel on 2^23-row matrices (profiles/r6_hash_forms.md): column loads software-pipelined one
    // chunk of 8 / 16 / 24 columns ahead, two rows per lane, R rows per thread in a loop - none faster at any width (72 columns: 0.93 of the
    // in-register BLAKE2s rate in every form, 24 columns per chunk 0.92; 8 columns 0.80 - 0.81 in every form): the plain form stays.
    AERO_LAUNCH(this, "hash_rows_kernel", rows * ((size_t)ncols * 8 + 32), (hash_rows_kernel<RowSrc>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                src, rows, leaves);
    check_launch("hash_rows");
}

void Context::hash_message_rows(const MsgSrc& src, size_t rows, Digest* leaves) {
    if (!rows) return;
    AERO_LAUNCH(this, "hash_message_rows_kernel", 0, (hash_rows_kernel<MsgSrc>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, src, rows, leaves);
    check_launch("hash_message_rows");
}
// FRI rows with a compile-time shape: the FOLD * DEG values of a row (each in its own line, FOLD rows apart) are fetched up front,
// then hashed - with the loads left next to the compression that consumes them a lane waits for memory FOLD * DEG / 2 times per row.
template <int FOLD, int DEG> __global__ __launch_bounds__(256) void hash_fri_rows_fixed_kernel(FriSrc s, Digest* leaves) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= s.rows) return;
    uint64_t e[FOLD * DEG];
#pragma unroll
    for (int j = 0; j < FOLD; j++) {
        e[j * DEG] = s.c0[i + (size_t)j * s.rows];
        if (DEG > 1) e[j * DEG + DEG - 1] = s.c1[i + (size_t)j * s.rows];
    }
    b2s::State st;
    b2s::init(st);
#pragma unroll
    for (int q = 0; q < FOLD * DEG; q += 2) b2s::compress_elems(st, e[q], e[q + 1], true, (uint32_t)(q + 2) * 32, q + 2 == FOLD * DEG);
    store_digest(&leaves[i], state_digest(st));
}
void Context::hash_fri_rows(const FriSrc& src, Digest* leaves) {
    if ((src.fold * src.deg) & 1) fail("hash_fri_rows: odd element count");
    const size_t abytes = src.rows * ((size_t)src.fold * src.deg * 8 + 32);
    const dim3 grid((unsigned)((src.rows + 255) / 256));
    const int shape = src.fold * 4 + src.deg;
    switch (shape) {
        case 8 * 4 + 1: AERO_LAUNCH(this, "hash_fri_rows_kernel", abytes, (hash_fri_rows_fixed_kernel<8, 1>), grid, dim3(256), 0, src, leaves); break;
        case 8 * 4 + 2: AERO_LAUNCH(this, "hash_fri_rows_kernel", abytes, (hash_fri_rows_fixed_kernel<8, 2>), grid, dim3(256), 0, src, leaves); break;
        case 4 * 4 + 1: AERO_LAUNCH(this, "hash_fri_rows_kernel", abytes, (hash_fri_rows_fixed_kernel<4, 1>), grid, dim3(256), 0, src, leaves); break;
        case 4 * 4 + 2: AERO_LAUNCH(this, "hash_fri_rows_kernel", abytes, (hash_fri_rows_fixed_kernel<4, 2>), grid, dim3(256), 0, src, leaves); break;
        case 2 * 4 + 1: AERO_LAUNCH(this, "hash_fri_rows_kernel", abytes, (hash_fri_rows_fixed_kernel<2, 1>), grid, dim3(256), 0, src, leaves); break;
        case 2 * 4 + 2: AERO_LAUNCH(this, "hash_fri_rows_kernel", abytes, (hash_fri_rows_fixed_kernel<2, 2>), grid, dim3(256), 0, src, leaves); break;
        default: AERO_LAUNCH(this, "hash_fri_rows_kernel", abytes, (hash_rows_kernel<FriSrc>), grid, dim3(256), 0, src, src.rows, leaves);
    }
    check_launch("hash_fri_rows");
}

// levels above a stored level of `c` nodes (heap indices [c, 2c)) up to the root
void Context::merkle_upper(Digest* nodes, size_t c, const CoinStep* coin) {
    // throughput regime: 3 levels per launch, one thread per subtree of 8
    while (c > ((size_t)1 << 17)) {
        size_t m = c / 8;
        AERO_LAUNCH(this, "merkle_up3_kernel", c * 32 + (c - m) * 32, merkle_up3_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, nodes, m);
        c = m;
    }
    // latency regime: up to 9 levels per launch, one workgroup per subtree
    while (c > 1) {
        int L = 0;
        while (L < 9 && ((size_t)1 << (L + 1)) <= c) L++;
        size_t m = c >> L;
        const CoinStep cs = (coin && m == 1) ? *coin : CoinStep{};     // the launch that produces the root also steps the coin
        if (quad_tops) AERO_LAUNCH(this, "merkle_multi_kernel", c * 64, merkle_multi_quad_kernel, dim3((unsigned)m), dim3(256), 0, nodes, m, L, cs);
        else AERO_LAUNCH(this, "merkle_multi_kernel", c * 64, merkle_multi_kernel, dim3((unsigned)m), dim3(256), 0, nodes, m, L, cs);
        c = m;
    }
    check_launch("merkle_upper");
}

void Context::merkle_build(Digest* nodes, size_t n, const CoinStep* coin) {
    if (n < 2 || (n & (n - 1))) fail("merkle_build: leaf count must be a power of two >= 2");
    merkle_upper(nodes, n, coin);
}

void Context::merkle_build_parts(Digest* nodes, size_t n, int log_parts) {
    if (n < 8 || (n & (n - 1)) || (n >> log_parts) == 0) fail("merkle_build_parts: leaf count must be a power of two >= 8 and >= parts");
    AERO_LAUNCH(this, "merkle_up3_kernel", n * 32 + (n - n / 8) * 32, merkle_up3_parts_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, nodes, n, log_parts);
    merkle_upper(nodes, n / 8);
}

template <class Src> static size_t src_leaf_bytes(const Src&);
template <> size_t src_leaf_bytes<RowSrc>(const RowSrc& s) { return (size_t)s.ncols * 8; }
template <> size_t src_leaf_bytes<FriSrc>(const FriSrc& s) { return (size_t)s.fold * s.deg * 8; }

// leaves + whole tree from a leaf source; the lowest `skip` levels (0 or 3) are not stored.
template <class Src> void Context::merkle_commit(const Src& src, size_t n, Digest* nodes, int skip, const CoinStep* coin) {
    if (n < 8 || (n & (n - 1))) fail("merkle_commit: leaf count must be a power of two >= 8");
    if (skip != 0 && skip != 3) fail("merkle_commit: skip must be 0 or 3", ST_INTERNAL);
    const size_t stored = skip ? (n / 8) * 32 : (n + n / 2 + n / 4 + n / 8) * 32;
    const dim3 grid((unsigned)((n / 8 + 255) / 256)), block(256);
    const size_t abytes = n * src_leaf_bytes(src) + stored;
    bool done = false;
    if constexpr (std::is_same<Src, RowSrc>::value) {
        done = true;
        // Stored leaf level (skip = 0: the stage entry points, trees that keep every level): the blocks of one XCD take a contiguous eighth of
        // the rows - 6 - 8 % faster on 2^23 and 2^27 rows (516 -> 480 us, 6.76 -> 6.39 ms; the launch is write-heavy: 32 B per row + 3 levels).
        // With skip = 3 (whole proofs: 4 B stored per row) the order makes no difference (398 us either way, profiles/r6_hash_forms.md).
        if (skip == 0) {
            if (src.ncols == 1) AERO_LAUNCH(this, "merkle_leaf8_kernel", abytes, (merkle_leaf8_rows_kernel<1, 1>), grid, block, 0, src, nodes, n, skip);
            else if (src.ncols == 2) AERO_LAUNCH(this, "merkle_leaf8_kernel", abytes, (merkle_leaf8_rows_kernel<2, 1>), grid, block, 0, src, nodes, n, skip);
            else if (src.ncols == 3) AERO_LAUNCH(this, "merkle_leaf8_kernel", abytes, (merkle_leaf8_rows_kernel<3, 1>), grid, block, 0, src, nodes, n, skip);
            else if (src.ncols == 4) AERO_LAUNCH(this, "merkle_leaf8_kernel", abytes, (merkle_leaf8_rows_kernel<4, 1>), grid, block, 0, src, nodes, n, skip);
            else done = false;
        }
        else if (src.ncols == 1) AERO_LAUNCH(this, "merkle_leaf8_kernel", abytes, (merkle_leaf8_rows_kernel<1>), grid, block, 0, src, nodes, n, skip);
        else if (src.ncols == 2) AERO_LAUNCH(this, "merkle_leaf8_kernel", abytes, (merkle_leaf8_rows_kernel<2>), grid, block, 0, src, nodes, n, skip);
        else if (src.ncols == 3) AERO_LAUNCH(this, "merkle_leaf8_kernel", abytes, (merkle_leaf8_rows_kernel<3>), grid, block, 0, src, nodes, n, skip);
        else if (src.ncols == 4) AERO_LAUNCH(this, "merkle_leaf8_kernel", abytes, (merkle_leaf8_rows_kernel<4>), grid, block, 0, src, nodes, n, skip);
        else done = false;
    }
    if (!done) AERO_LAUNCH(this, "merkle_leaf8_kernel", abytes, (merkle_leaf8_kernel<Src>), grid, block, 0, src, nodes, n, skip);
    merkle_upper(nodes, n / 8, coin);
}
template void Context::merkle_commit<RowSrc>(const RowSrc&, size_t, Digest*, int, const CoinStep*);
template void Context::merkle_commit<FriSrc>(const FriSrc&, size_t, Digest*, int, const CoinStep*);

template <class Src> void Context::merkle_recompute(const Src& src, size_t n, const uint64_t* idx_dev, int count, Digest* out_dev) {
    if (count <= 0) return;
    AERO_LAUNCH(this, "merkle_recompute_kernel", 0, (merkle_recompute_kernel<Src>), dim3((count * 8 + 255) / 256), dim3(256), 0, src, n, idx_dev, count, out_dev);
    check_launch("merkle_recompute");
}
template void Context::merkle_recompute<RowSrc>(const RowSrc&, size_t, const uint64_t*, int, Digest*);
template void Context::merkle_recompute<FriSrc>(const FriSrc&, size_t, const uint64_t*, int, Digest*);

}  // namespace aero
