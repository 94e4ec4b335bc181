// Internal declarations shared by the HIP translation units of libaero_stark.so.
// Nothing here is part of the public boundary (see include/aero_stark.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "blake2s_hash.hpp"
#include "gl_field.hpp"

namespace aero {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& s) : std::runtime_error(s), code(c) {}
};
// status codes of the C ABI (include/aero_stark.h)
enum { ST_OK = 0, ST_BAD_ARG = -1, ST_OOM = -2, ST_HIP = -3, ST_COMM = -4, ST_UNSUPPORTED = -5, ST_INTERNAL = -6 };

[[noreturn]] inline void fail(const std::string& s, int code = ST_BAD_ARG) { throw Error(code, s); }

#define AERO_HIP(expr)                                                                                       \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            throw ::aero::Error(e_ == hipErrorOutOfMemory ? ::aero::ST_OOM : ::aero::ST_HIP,                  \
                                std::string(#expr) + ": " + hipGetErrorString(e_));                          \
    } while (0)

// Every kernel launch goes through this macro so that it can be bracketed by HIP events on the launch stream.
// `abytes` = algorithmic HBM bytes of the launch (every input read once + every output written once).
#define AERO_LAUNCH(ctxp, name, abytes, kern, grid, block, shm, ...)                 \
    do {                                                                             \
        ::aero::Context* c_ = (ctxp);                                                \
        const bool t_ = c_->kernel_timing && c_->kt_match(name);                     \
        if (t_) c_->kt_begin(name, (size_t)(abytes));                                \
        hipLaunchKernelGGL(kern, grid, block, shm, c_->stream, __VA_ARGS__);         \
        if (t_) c_->kt_end();                                                        \
    } while (0)

struct NttTables {
    int log_n = 0, h = 0;
    uint64_t *lo_fwd = nullptr, *hi_fwd = nullptr, *lo_inv = nullptr, *hi_inv = nullptr;
};
struct NttPass {
    int log_s, log_r, log_tl;
};
// Optional second output of an LDE: rows = 0 mod 2^log_step, stored densely (row r at r >> log_step), column stride col_stride.
// With log_split = k > 0 the compact rows are de-interleaved into 2^k parts: compact row j sits at (j mod 2^k) * (rows / 2^k) +
// j / 2^k, so that a reader of every 2^k-th compact row (DEEP) finds its rows contiguous in part 0 while a reader of all
// compact rows (constraint evaluation) still reads whole sectors (a wavefront touches 2^k contiguous runs).
struct CompactOut {
    uint64_t* ptr = nullptr;
    size_t col_stride = 0;
    int log_step = 0;
    int log_split = 0;
};
std::vector<NttPass> plan_passes(int log_n, bool reg_passes, int first_bits = 12, bool radix128 = false, int lds_max_r = 8);

typedef b2s::Digest Digest;   // 8 x u32, byte order = digest byte order (little-endian words)

// Leaf sources of a Merkle commitment: leaf j = hash_elements(row j)
struct RowSrc {   // rows of a column-major matrix
    const uint64_t* cols;
    size_t stride;   // column stride in elements
    int ncols;
};
struct MsgSrc {   // rows of a bincode HashingWorkItem as they lie in the message: words[offs[i]] = length of row i, the elements follow
    const uint64_t* words;
    const uint64_t* offs;
};
struct FriSrc {   // FRI layer rows: row i = (v[i + j*rows])_{j < fold}, each value = deg base components (c0, c1)
    const uint64_t* c0;
    const uint64_t* c1;
    int deg;
    size_t rows;
    int fold;
};

// Transcript step appended to a Merkle tree build (FRI commit phase): once the root is known the same thread reseeds the coin
// with it and draws the layer's folding challenge (random.cairo:108-114,159-166) - no separate launch, no host round trip.
struct CoinStep {
    Digest* seed_io = nullptr;   // coin seed: reseeded in place with the root
    uint64_t* alpha_out = nullptr;   // `deg` u64: the drawn element
    int deg = 1;
    Digest* root_out = nullptr;  // the root once more, in the block the host reads after the last layer (mapped pinned memory)
    Digest* seed_out = nullptr;  // the reseeded coin seed once more (mapped pinned memory): the host checks its own replay against the last one
    uint32_t* flag_out = nullptr;   // mapped pinned word: the launch that produces the root stores flag_seq there behind everything else it wrote for
    uint32_t flag_seq = 0;          // the host, which polls the word instead of waiting for the stream (Context::wait_flag)
};

// The small end of the FRI commit phase in ONE launch (hash.hip: fri_tail_kernel): every layer whose domain has at most
// FRI_TAIL_MAX_DOM points - leaf hashing, tree, transcript step and fold, layer after layer, by a single workgroup working out of
// LDS. These layers are pure latency (a 2^11-point layer cannot fill one CU, let alone 256), and as separate launches each of
// them costs four kernel boundaries.
constexpr int FRI_TAIL_MAX_DOM = 4096;
constexpr int FRI_TAIL_MAX_ROWS = 512;      // rows (= leaves) of the first tail layer: one per thread of the workgroup
constexpr int FRI_TAIL_MAX_LAYERS = 13;
struct FriTailArgs {
    int deg, n_layers;                          // n_layers commitments (the last one has no fold behind it)
    uint32_t dom0;                              // domain size of the first tail layer
    const uint64_t* vals0;                      // its evaluations: [deg][dom0]
    uint64_t* vals_out[FRI_TAIL_MAX_LAYERS];    // evaluations of tail layer i + 1: [deg][dom_(i+1)], i < n_layers - 1
    Digest* nodes[FRI_TAIL_MAX_LAYERS];         // tree of tail layer i: 2 * rows_i slots (rows_i = 1: the root in slot 1)
    Digest* roots_out;                          // n_layers roots
    Digest* seed_io;                            // coin seed, reseeded with every root
    Digest* seed_out;                           // the seed once more after every step (mapped pinned memory; may be null)
    uint32_t* flag_out;                         // mapped pinned word, receives flag_seq at the very end (may be null): Context::wait_flag
    uint32_t flag_seq;
    uint64_t* alphas_out;                       // deg u64 per layer
    uint64_t gen_inv, fold_inv, w_inv0;         // 1 / domain offset, 1 / fold, w_dom0^-1
};

// Arguments of the fused opening launch (hash.hip: openings_kernel)
constexpr int OPEN_MAX_JOBS = 4;
struct OpeningArgs {
    const uint64_t* addr = nullptr;      // n_u64 element addresses, then n_dig digest addresses (0 = leave zeros); mapped pinned memory
    uint32_t n_u64 = 0, n_dig = 0, gather_blocks = 0;
    uint64_t* out = nullptr;             // value block: n_u64 u64, then 4 u64 per digest; mapped pinned memory
    struct Job {                         // unstored low nodes of one row-matrix tree, recomputed from the matrix
        RowSrc src;
        size_t n = 0;                    // leaves of the tree
        const uint64_t* idx = nullptr;   // heap indices (mapped pinned memory)
        int count = 0;
        uint32_t blocks = 0;
        Digest* out = nullptr;           // `count` digests (mapped pinned memory)
    } jobs[OPEN_MAX_JOBS];
    int n_jobs = 0;
};

// One context = one device + one stream. Not thread-safe: one host thread drives it (SURVEY 8b "Threading").
class Context {
public:
    int device = 0;
    hipStream_t stream = nullptr;
    std::string last_error;

    explicit Context(int dev);
    ~Context();
    // Several contexts of one pool copy their traces from host memory over the same PCIe link. Left alone, proofs that start together
    // copy together (each at 1 / k of the link) and then compute together: the copies hide behind nothing. The gate orders the copies
    // on the device - a context's copy waits for the previous context's copy - so the proofs fall out of phase and every copy runs
    // under the other proofs' kernels (2^22 x (72 + 9), 3 in flight: 420 -> 3xx ms per round).
    struct CopyGate {
        std::mutex mu;
        hipEvent_t last = nullptr;
    };
    std::shared_ptr<CopyGate> copy_gate;   // shared with the pool: a context kept alive by one of its matrices may outlive aero_pool_destroy
    hipEvent_t gate_event = nullptr;       // this context's last gated copy
    // Pool prefetch (capi.hip: aero_pool::worker): while a slot proves trace r, the copy of trace r + 1 runs on its copy stream into the other of
    // two landing buffers. The next host-trace proof on this context then finds its trace at `dev` once `ready` has fired - the resident path
    // plus an event wait and the canonical-form check of the host path (Prover::set_landed_trace). Cleared by the worker after the call.
    struct LandedTrace {
        const uint64_t* dev = nullptr;
        hipEvent_t ready = nullptr;
        size_t bytes = 0;                // of the landing buffer's trace: a call that picks it up checks it against its own shape
    } landed;
    // second stream + events for the host-to-device copy of a wide trace: column group g + 1 travels while group g is transformed
    hipStream_t copy_stream = nullptr;
    hipStream_t get_copy_stream();
    hipEvent_t sync_event(size_t i);

    // ---- memory: size-keyed free lists so steady-state proving performs no hipMalloc/hipFree ----
    void* pool_alloc(size_t bytes);
    void pool_free(void* p);
    void* dev_alloc(size_t bytes);       // lives until the context dies (tables)
    void* scratch_alloc(size_t bytes);   // released by scratch_reset() (stream-ordered reuse)
    void poison_block(void* p, size_t bytes);   // AERO_POISON_ALLOC=1: fill a block that is being handed out (diagnosis of reads of unwritten words)
    void scratch_reset();
    // pinned host staging (bump allocator) for small async H2D parameter blocks and D2H results; a block stays valid
    // until stage_reset(), which callers issue only after a stream synchronisation
    void* stage_alloc(size_t bytes);
    // Completion word in mapped pinned memory. A single-workgroup launch at a transcript point (tree top, FRI tail) stores a sequence number there
    // behind its results (system-scope fence); the host polls the word instead of waiting for the stream's completion signal - 5 us less per
    // round trip (tools/ubench_roundtrip.hip, modes B / C). wait_flag() also returns when the stream has drained (hipStreamQuery every so often):
    // correctness never rests on the word, a kernel that died shows up as the stream's error.
    uint32_t* flag_host = nullptr;
    uint32_t* flag_dev = nullptr;
    uint32_t flag_seq = 0;
    uint32_t next_flag();                 // a fresh sequence number (allocates the word at first use)
    void wait_flag(uint32_t seq);
    unsigned long long* grind_slots = nullptr;   // two words the grinding launches take their minimum in, in turn (stark.hip: run_grind)
    int grind_parity = 0;
    void stage_reset() { stage_off = 0; }
    // The staging block is mapped pinned memory: a kernel can read and write it through this alias (hipHostGetDevicePointer). The few bytes a
    // proof hands to the host at each transcript point (roots, the OOD frame, the opened values) are stored there by the kernels that
    // produce them - the host waits for the stream and finds them, no copy kernel in between (round 5; tools/ubench_roundtrip.hip: 18.6 ->
    // 16.1 us per round trip, and one blit launch less on the proof's own timeline). Only for data the device writes once and never reads.
    template <class T> T* stage_dev(T* host_ptr) { return reinterpret_cast<T*>(stage_dev_base + (reinterpret_cast<uint8_t*>(host_ptr) - stage_base)); }
    unsigned int* pinned_word();         // one pinned 32-bit word that outlives stage_reset() (deferred input-check verdict)
    // One pinned host block kept by the context (grow-only; no allocation per proof) for host steps that exchange whole columns with the
    // device (air_host.hip: general auxiliary recurrences). The caller orders its own reuse: copies still in flight from an earlier use
    // must sit on this context's streams in front of whatever touches the block next. Growing it drains both streams first.
    uint64_t* host_arena(size_t bytes);
    size_t bytes_in_use = 0, bytes_peak = 0;
    // AERO_POOL_GUARD=1 (diagnosis): every block is mapped by itself at the END of its own virtual-address reservation with an unmapped
    // granule behind it (hipMemAddressReserve / hipMemMap), nothing is reused - a kernel that reads or writes one byte past a buffer
    // takes a GPU memory fault at that access instead of silently touching a neighbour. Slow; for the test-suite, not for proving.
    bool guard_mode = false;
    struct GuardBlock { void* va; size_t va_bytes, map_bytes; hipMemGenericAllocationHandle_t handle; size_t user_bytes; };
    std::map<void*, GuardBlock> guard_blocks;
    void* guard_alloc(size_t bytes);
    void guard_free(void* p);

    void check_launch(const char* what);
    void sync();
    // Small device -> host read on the critical path of the transcript: copy into pinned staging + ONE stream synchronisation.
    // (Measured and dropped: a one-workgroup kernel that stores into mapped pinned memory followed by a sequence word the host
    // spins on. With the producing kernel storing directly that saves 10 us per round trip - tools/ubench_roundtrip.hip - but as
    // a separate launch it is 1 % SLOWER per proof than the runtime's own copy + synchronise, which already spins on a signal.)
    void fetch(void* dst, const void* dev_src, size_t bytes);

    // ---- per-kernel HIP-event timing on this context's stream (off by default) ----
    bool kernel_timing = false;
    std::string kt_filter;       // empty = every kernel, otherwise only launches of this kernel are bracketed
    bool kt_match(const char* name) const { return kt_filter.empty() || kt_filter == name; }
    void kt_begin(const char* name, size_t abytes);
    void kt_end();
    std::string kt_report();   // "name calls total_ms algorithmic_bytes" lines sorted by total; resets the counters

    // ---- NTT (ntt.hip) ----
    NttTables* ntt_tables(int log_n);
    void ensure_small_twiddles();
    // pass-boundary twiddle table of a register-only strided pass (ntt.hip), cached per (size, stride, radix, direction)
    const uint64_t* pass_twiddles(int log_n, int log_s, int log_r, bool inverse);
    bool quad_tops = true;     // latency-bound tree tops: four lanes per BLAKE2s compression (AERO_QUAD_TOPS=0: one lane)
    bool pass_names = false;   // AERO_NTT_NAMES=1: forward passes are timed under per-variant names (diagnosis)
    bool two_phase = true;     // contiguous pass of the blowup-8 LDE as two register transforms around one LDS exchange (AERO_NTT_2PHASE=0: LDS rounds)
    bool reg_passes = true;    // strided passes of radix 16..64 run entirely in registers (AERO_NTT_REG=0: LDS passes only)
    bool radix128 = true;      // forward transforms take radix-128 two-lane passes where that saves a pass (AERO_NTT_R128=0: radix <= 64)
    bool fwd_two_phase(int log_out, int log_pad) const;
    // returns true when `compact` was requested AND written (the last pass must be a strided register pass: transforms that fit the
    // contiguous pass alone do not produce it and the caller falls back to strided reads of the full matrix)
    bool ntt_forward(const uint64_t* in, size_t in_stride, uint64_t* out, size_t out_stride, int ncols, int log_out, int log_pad,
                     const CompactOut* compact = nullptr);
    // bad (optional, device): 1 is ORed in when an input element is >= p (checked by the pass that reads the input)
    // `src` (optional): where the values are read from - the pass that reads them writes into `data`, the rest runs in place there (saves the
    // copy a caller would make to keep its evaluations)
    void ntt_inverse(uint64_t* data, size_t stride, int ncols, int log_n, uint64_t c0, uint64_t sa, uint64_t sb, int shift, unsigned int* bad = nullptr,
                     const uint64_t* src = nullptr, size_t src_stride = 0);

    // ---- hashing (hash.hip) ----
    // leaf j = hash_elements(row j) of a column-major matrix (ncols columns, `rows` rows, column stride in elements)
    void hash_rows(const uint64_t* cols, size_t col_stride, int ncols, size_t rows, Digest* leaves);
    // FRI layer rows: row i = (v[i + j*rows] for j < fold), each value having `deg` base components stored as
    // component columns comp[k] (k < deg): leaf i = hash_elements(flattened row)
    void hash_fri_rows(const FriSrc& src, Digest* leaves);
    void hash_message_rows(const MsgSrc& src, size_t rows, Digest* leaves);   // rows of any length, values >= p reduced
    // nodes[n + i] already hold the leaves; fills nodes[1 .. n-1]
    void merkle_build(Digest* nodes, size_t n, const CoinStep* coin = nullptr);
    // same when the leaf level holds 2^log_parts pieces in arrival order (leaf u in slot n + (u mod parts)*(n/parts) + u/parts)
    void merkle_build_parts(Digest* nodes, size_t n, int log_parts);
    // fold = 2, 4 or 8; see FriTailArgs
    void fri_tail(const FriTailArgs& a, int fold);
    // levels above a stored level of c nodes (heap indices [c, 2c)) up to the root
    void merkle_upper(Digest* nodes, size_t c, const CoinStep* coin = nullptr);
    void openings(const OpeningArgs& a);
    // fused leaf hashing + whole tree; the lowest `skip` (0 or 3) levels are not stored (nodes holds 2n >> skip slots)
    template <class Src> void merkle_commit(const Src& src, size_t n, Digest* nodes, int skip = 0, const CoinStep* coin = nullptr);
    // digests of unstored low nodes (heap indices >= 2n >> skip), recomputed from the leaf source
    template <class Src> void merkle_recompute(const Src& src, size_t n, const uint64_t* idx_dev, int count, Digest* out_dev);

    // internal state
    // AIR programs compiled at run time (air_jit.hip): AERO_AIR_JIT=0 keeps the interpreter; modules loaded on this context's device
    bool air_jit = true;
    std::map<uint64_t, void*> jit_funcs;
    std::vector<hipModule_t> jit_modules;
    static void unload_jit_modules(Context* ctx);   // air_jit.hip: hipModuleUnload under the lock that serialises module loading
    std::vector<std::shared_ptr<void>> jit_blobs;
    bool fri_tail_attr_set = false;   // the opt-in for > 64 KiB of dynamic LDS was made on this context's device
    std::map<int, NttTables> ntt_tabs;
    std::map<uint64_t, uint64_t*> pass_tabs;
    std::map<std::vector<uint64_t>, uint64_t*> ktab_cache;
    // FibAir constraint evaluation: the divisor inverses and degree-adjustment powers of a constraint domain (5 x rows words), built by
    // the first proof of a shape and kept (a context keeps the two shapes it used last, up to 2^25 rows each; AERO_CONS_INV_TABLE=0: every proof inverts per thread)
    // One table per (device, shape) for ALL contexts of the process (the eight slots of a pool used to hold eight copies: 8 x 5 x 2^23 x 8 B):
    // a context keeps the two shapes it used last (least recently used out first), a table lives while any context holds it, its bytes count
    // in bytes_in_use / bytes_peak of every holder, and an allocation that fails is not an error - the kernel then inverts per row.
    struct SharedTable {
        int device = 0;
        uint64_t* ptr = nullptr;
        size_t bytes = 0;
        hipEvent_t ready = nullptr;      // recorded behind the build on the building context's stream; other contexts wait for it once
        ~SharedTable();
    };
    std::vector<std::pair<std::vector<uint64_t>, std::shared_ptr<SharedTable>>> cons_inv_cache;     // most recently used first
    const uint64_t* cons_inv_table_for(const std::vector<uint64_t>& key, size_t bytes, const std::function<void(uint64_t*)>& build);   // nullptr: no table
    bool cons_inv_table = true;
    bool deep_coeff = true;    // DEEP composition in coefficient form (base field, one GPU); AERO_DEEP_COEFF=0: evaluated on the trace-length coset and interpolated
   // final-pass scale tables of ntt_inverse, keyed by their parameters
    uint64_t *tw4096_fwd = nullptr, *tw4096_inv = nullptr;
    uint64_t *twmt_fwd = nullptr, *twmt_inv = nullptr, *twmt12_inv = nullptr;   // [r * 64 + k] = w_2048^(+-r k), r < 32

private:
    struct KtRec { const char* name; size_t abytes; hipEvent_t start, stop; };
    std::vector<KtRec> kt_recs;
    std::vector<hipEvent_t> kt_pool;
    hipEvent_t kt_event();
    std::multimap<size_t, void*> free_blocks;
    std::map<void*, size_t> live_blocks;
    std::vector<void*> persistent, scratch;
    std::vector<hipEvent_t> sync_events;
    uint8_t* stage_base = nullptr;
    uint8_t* stage_dev_base = nullptr;

    unsigned int* pinned_flag = nullptr;
    uint64_t* arena_base = nullptr;
    size_t arena_cap = 0;
    size_t stage_cap = 0, stage_off = 0;
};

// RAII device buffer from the context pool
template <class T> struct DevBuf {
    Context* ctx = nullptr;
    T* p = nullptr;
    size_t n = 0;
    DevBuf() {}
    DevBuf(Context* c, size_t count) : ctx(c), n(count) { p = (T*)c->pool_alloc(count * sizeof(T)); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : ctx(o.ctx), p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); ctx = o.ctx; p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() { if (p) { ctx->pool_free(p); p = nullptr; n = 0; } }
    T* get() const { return p; }
    size_t bytes() const { return n * sizeof(T); }
};

}  // namespace aero
