// Host pipeline of the MI355X STARK prover (see prover.hpp for the reference interface it mirrors).
#include "prover.hpp"

#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <set>

#include "air_host.hpp"
#include "stark_kernels.hpp"

namespace aero {

using gl::FB;
using gl::FQ;

// ================================================================================================
// Context
Context::Context(int dev) : device(dev) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) throw Error(ST_HIP, "no HIP device available: libaero_stark requires an MI355X (there is no CPU fallback)");
    if (dev < 0 || dev >= count) throw Error(ST_BAD_ARG, "device id out of range");
    AERO_HIP(hipSetDevice(dev));
    AERO_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (const char* e = getenv("AERO_NTT_REG")) reg_passes = e[0] != '0';
    if (const char* e = getenv("AERO_NTT_NAMES")) pass_names = e[0] != '0';
    if (const char* e = getenv("AERO_NTT_2PHASE")) two_phase = e[0] != '0';
    if (const char* e = getenv("AERO_QUAD_TOPS")) quad_tops = e[0] != '0';
    if (const char* e = getenv("AERO_AIR_JIT")) air_jit = e[0] != '0';
    if (const char* e = getenv("AERO_NTT_R128")) radix128 = e[0] != '0';
    if (const char* e = getenv("AERO_CONS_INV_TABLE")) cons_inv_table = e[0] != '0';
    if (const char* e = getenv("AERO_DEEP_COEFF")) deep_coeff = e[0] != '0';
    if (const char* e = getenv("AERO_POOL_GUARD")) guard_mode = e[0] == '1';
}
Context::~Context() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    if (guard_mode) {
        while (!guard_blocks.empty()) guard_free(guard_blocks.begin()->first);
        free_blocks.clear(); live_blocks.clear(); persistent.clear();
    }
    for (auto& kv : free_blocks) (void)hipFree(kv.second);
    for (auto& kv : live_blocks) (void)hipFree(kv.first);
    for (void* p : persistent) (void)hipFree(p);
    unload_jit_modules(this);   // may run on whichever host thread drops the last handle: not while another thread loads a module
    if (stage_base) (void)hipHostFree(stage_base);
    if (flag_host) (void)hipHostFree(flag_host);
    if (pinned_flag) (void)hipHostFree(pinned_flag);
    if (arena_base) (void)hipHostFree(arena_base);
    for (auto& r : kt_recs) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
    for (hipEvent_t e : kt_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : sync_events) (void)hipEventDestroy(e);
    if (gate_event) {
        if (copy_gate) { std::lock_guard<std::mutex> lk(copy_gate->mu); if (copy_gate->last == gate_event) copy_gate->last = nullptr; }
        (void)hipEventDestroy(gate_event);
    }
    if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
    if (stream) (void)hipStreamDestroy(stream);
}
hipStream_t Context::get_copy_stream() {
    if (!copy_stream) AERO_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    return copy_stream;
}
hipEvent_t Context::sync_event(size_t i) {
    while (sync_events.size() <= i) {
        hipEvent_t e;
        AERO_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        sync_events.push_back(e);
    }
    return sync_events[i];
}
void* Context::guard_alloc(size_t bytes) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    AERO_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (gran < 4096) gran = 4096;
    const size_t user = (bytes + 255) & ~(size_t)255;                  // what pool_alloc hands out: an access past THIS is outside the block in normal operation too
    GuardBlock g{};
    g.user_bytes = user;
    g.map_bytes = (user + gran - 1) / gran * gran;
    g.va_bytes = g.map_bytes + gran;                                   // the last granule stays unmapped
    AERO_HIP(hipMemAddressReserve(&g.va, g.va_bytes, gran, nullptr, 0));
    hipError_t e = hipMemCreate(&g.handle, g.map_bytes, &prop, 0);
    if (e != hipSuccess) { (void)hipMemAddressFree(g.va, g.va_bytes); (void)hipGetLastError(); throw Error(ST_OOM, std::string("guard allocation failed: ") + hipGetErrorString(e)); }
    AERO_HIP(hipMemMap(g.va, g.map_bytes, 0, g.handle, 0));
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    AERO_HIP(hipMemSetAccess(g.va, g.map_bytes, &acc, 1));
    void* p = (char*)g.va + (g.map_bytes - user);                      // the block ends where the mapping ends
    guard_blocks[p] = g;
    return p;
}
void Context::guard_free(void* p) {
    auto it = guard_blocks.find(p);
    if (it == guard_blocks.end()) return;
    (void)hipStreamSynchronize(stream);
    if (copy_stream) (void)hipStreamSynchronize(copy_stream);
    const GuardBlock g = it->second;
    guard_blocks.erase(it);
    (void)hipMemUnmap(g.va, g.map_bytes);
    (void)hipMemRelease(g.handle);
    // the reservation is NOT returned: a virtual address that is handed out again for another block would let a stale translation or
    // cache line of the old mapping answer for the new one (seen on ROCm 7.2: the first kilobytes of a re-mapped range read back old
    // data); a stale pointer must fault instead. 48 bits of address space outlast any test run.
}
void* Context::pool_alloc(size_t bytes) {
    if (guard_mode) {
        void* p = guard_alloc(bytes ? bytes : 16);
        live_blocks[p] = bytes;
        bytes_in_use += bytes;
        if (bytes_in_use > bytes_peak) bytes_peak = bytes_in_use;
        return p;
    }
    if (bytes == 0) bytes = 256;
    bytes = (bytes + 255) & ~(size_t)255;
    void* p = nullptr;
    auto it = free_blocks.find(bytes);
    if (it != free_blocks.end()) {
        p = it->second;
        free_blocks.erase(it);
    } else {
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            // release cached blocks and retry once
            (void)hipStreamSynchronize(stream);
            for (auto& kv : free_blocks) (void)hipFree(kv.second);
            free_blocks.clear();
            e = hipMalloc(&p, bytes);
            if (e != hipSuccess) {
                (void)hipGetLastError();   // the failed hipMalloc must not surface again at the next launch check
                throw Error(ST_OOM, "device allocation of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
            }
        }
    }
    live_blocks[p] = bytes;
    bytes_in_use += bytes;
    if (bytes_in_use > bytes_peak) bytes_peak = bytes_in_use;
    poison_block(p, bytes);
    return p;
}
void Context::pool_free(void* p) {
    auto it = live_blocks.find(p);
    if (it == live_blocks.end()) return;
    bytes_in_use -= it->second;
    if (guard_mode) { live_blocks.erase(it); guard_free(p); return; }
    free_blocks.insert({it->second, p});   // single stream: reuse is stream-ordered
    live_blocks.erase(it);
}
void* Context::dev_alloc(size_t bytes) {
    if (guard_mode) return guard_alloc(bytes ? bytes : 16);          // released with the context
    void* p = nullptr;
    AERO_HIP(hipMalloc(&p, bytes ? bytes : 256));
    persistent.push_back(p);
    return p;
}
// ---- tables shared by the contexts of one device (aero_internal.hpp: SharedTable) ----
namespace {
std::mutex g_shared_tab_mu;
std::map<std::vector<uint64_t>, std::weak_ptr<Context::SharedTable>> g_shared_tabs;      // key + device
}
Context::SharedTable::~SharedTable() {
    // the last holder is gone; kernels of its earlier proofs may still read the table
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    if (ready) (void)hipEventDestroy(ready);
    if (ptr) (void)hipFree(ptr);
    (void)hipGetLastError();
}
const uint64_t* Context::cons_inv_table_for(const std::vector<uint64_t>& key, size_t bytes, const std::function<void(uint64_t*)>& build) {
    for (size_t i = 0; i < cons_inv_cache.size(); i++)
        if (cons_inv_cache[i].first == key) {
            if (i) std::rotate(cons_inv_cache.begin(), cons_inv_cache.begin() + i, cons_inv_cache.begin() + i + 1);
            return cons_inv_cache.front().second->ptr;
        }
    // make room first (the peak is then two tables, not three); the evicted table dies here unless another context still holds it
    while (cons_inv_cache.size() >= 2) { bytes_in_use -= cons_inv_cache.back().second->bytes; cons_inv_cache.pop_back(); }
    std::shared_ptr<SharedTable> t;
    {
        std::lock_guard<std::mutex> lk(g_shared_tab_mu);
        std::vector<uint64_t> gkey = key;
        gkey.push_back((uint64_t)device);
        auto it = g_shared_tabs.find(gkey);
        if (it != g_shared_tabs.end()) t = it->second.lock();
        if (!t) {
            static const bool refuse = getenv("AERO_TEST_TABLE_OOM") != nullptr;     // tests: the allocation fails, the proof must not
            void* p = nullptr;
            const hipError_t e = refuse ? hipErrorOutOfMemory : hipMalloc(&p, bytes);
            if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            t = std::make_shared<SharedTable>();
            t->device = device; t->ptr = (uint64_t*)p; t->bytes = bytes;
            AERO_HIP(hipEventCreateWithFlags(&t->ready, hipEventDisableTiming));
            build(t->ptr);
            AERO_HIP(hipEventRecord(t->ready, stream));
            g_shared_tabs[gkey] = t;
        }
    }
    AERO_HIP(hipStreamWaitEvent(stream, t->ready, 0));        // built on another context's stream, perhaps
    cons_inv_cache.insert(cons_inv_cache.begin(), {key, t});
    bytes_in_use += bytes;
    if (bytes_in_use > bytes_peak) bytes_peak = bytes_in_use;
    return t->ptr;
}
void* Context::scratch_alloc(size_t bytes) {
    void* p = pool_alloc(bytes);
    scratch.push_back(p);
    return p;
}
// AERO_POISON_ALLOC=1 (diagnosis): every pool block is filled with a non-canonical pattern when it is handed out, on the context's stream - a
// kernel that reads a word nobody wrote then fails the same way every time instead of depending on what the block held before
void Context::poison_block(void* p, size_t bytes) {
    static const bool on = getenv("AERO_POISON_ALLOC") != nullptr && getenv("AERO_POISON_ALLOC")[0] != '0';
    // (completed before the call returns: a block may be written next from another stream of the context that is only ordered behind EARLIER work of this one)
    if (on && p && bytes) { AERO_HIP(hipMemsetAsync(p, 0xA5, bytes, stream)); AERO_HIP(hipStreamSynchronize(stream)); }
}
void Context::scratch_reset() {
    for (void* p : scratch) pool_free(p);
    scratch.clear();
}
void* Context::stage_alloc(size_t bytes) {
    bytes = (bytes + 63) & ~(size_t)63;
    if (!stage_base) {
        stage_cap = (size_t)8 << 20;
        AERO_HIP(hipHostMalloc((void**)&stage_base, stage_cap, hipHostMallocCoherent | hipHostMallocMapped));     // kernels store results here, the host reads them behind a flag
        AERO_HIP(hipHostGetDevicePointer((void**)&stage_dev_base, stage_base, 0));
    }
    if (bytes > stage_cap) fail("staging request too large", ST_INTERNAL);
    if (stage_off + bytes > stage_cap) { sync(); stage_off = 0; }   // wrap only when every earlier copy has completed
    void* p = stage_base + stage_off;
    stage_off += bytes;
    return p;
}
uint64_t* Context::host_arena(size_t bytes) {
    if (bytes > arena_cap) {
        if (copy_stream) (void)hipStreamSynchronize(copy_stream);
        sync();
        if (arena_base) (void)hipHostFree(arena_base);
        arena_base = nullptr; arena_cap = 0;
        const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        const hipError_t e = hipHostMalloc((void**)&arena_base, want, hipHostMallocDefault);
        if (e != hipSuccess) { (void)hipGetLastError(); arena_base = nullptr; throw Error(ST_OOM, "pinned host block of " + std::to_string(want) + " bytes: " + hipGetErrorString(e)); }
        arena_cap = want;
    }
    return arena_base;
}
unsigned int* Context::pinned_word() {
    if (!pinned_flag) AERO_HIP(hipHostMalloc((void**)&pinned_flag, 64, hipHostMallocDefault));
    return pinned_flag;
}
void Context::check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw Error(ST_HIP, std::string(what) + ": kernel launch failed: " + hipGetErrorString(e));
}
void Context::sync() {
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) throw Error(ST_HIP, std::string("stream synchronize: ") + hipGetErrorString(e));
}

uint32_t Context::next_flag() {
    if (!flag_host) {
        // the protocol (a kernel's system-scope store observed by a polling host) rests on coherent, mapped host memory: asked for by name
        AERO_HIP(hipHostMalloc((void**)&flag_host, 64, hipHostMallocCoherent | hipHostMallocMapped));
        AERO_HIP(hipHostGetDevicePointer((void**)&flag_dev, flag_host, 0));
        *flag_host = 0;
    }
    if (++flag_seq == 0) ++flag_seq;
    return flag_seq;
}
void Context::wait_flag(uint32_t seq) {
    static const bool poll = !(getenv("AERO_POLL_FLAGS") && getenv("AERO_POLL_FLAGS")[0] == '0');
    if (!poll) { sync(); return; }
    // A polite spin: `pause` in every iteration (the sibling hyper-thread and the memory pipeline get the cycles); the stream's own state is
    // looked at every 2048 polls; and a wait that has lasted longer than a root's round trip ever does on an idle GPU (the word is then behind
    // OTHER proofs' kernels: a pool with more slots than the process may use CPUs would otherwise starve the slots that still have launches
    // to enqueue) gives its time slice away between polls.
    volatile uint32_t* f = flag_host;
    for (uint32_t spin = 1;; spin++) {
        if (*f == seq) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return; }
        __builtin_ia32_pause();
        if ((spin & 2047u) == 0) {
            const hipError_t e = hipStreamQuery(stream);
            if (e == hipSuccess) return;                   // drained: whatever the kernels wrote is visible
            if (e != hipErrorNotReady) throw Error(ST_HIP, std::string("stream query: ") + hipGetErrorString(e));
        }
        if (spin > 16384u) sched_yield();       // (same-box A/B against the bare loop and against default host-memory flags: no difference on the headline, profiles/r6_headline_spin_flags_ab.txt)
    }
}

void Context::fetch(void* dst, const void* dev_src, size_t bytes) {
    if (bytes == 0) { sync(); return; }
    void* h = stage_alloc(bytes);
    AERO_HIP(hipMemcpyAsync(h, dev_src, bytes, hipMemcpyDeviceToHost, stream));
    sync();
    memcpy(dst, h, bytes);
}

hipEvent_t Context::kt_event() {
    if (!kt_pool.empty()) { hipEvent_t e = kt_pool.back(); kt_pool.pop_back(); return e; }
    hipEvent_t e;
    AERO_HIP(hipEventCreate(&e));
    return e;
}
void Context::kt_begin(const char* name, size_t abytes) {
    KtRec r{name, abytes, kt_event(), kt_event()};
    AERO_HIP(hipEventRecord(r.start, stream));
    kt_recs.push_back(r);
}
void Context::kt_end() { AERO_HIP(hipEventRecord(kt_recs.back().stop, stream)); }
std::string Context::kt_report() {
    sync();
    struct Agg { int calls = 0; double ms = 0; double bytes = 0; };
    std::map<std::string, Agg> agg;
    for (auto& r : kt_recs) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, r.start, r.stop);
        auto& a = agg[r.name];
        a.calls += 1; a.ms += ms; a.bytes += (double)r.abytes;
        kt_pool.push_back(r.start); kt_pool.push_back(r.stop);
    }
    kt_recs.clear();
    std::vector<std::pair<double, std::string>> rows;
    for (auto& kv : agg) {
        char line[256];
        snprintf(line, sizeof line, "%s %d %.6f %.0f", kv.first.c_str(), kv.second.calls, kv.second.ms, kv.second.bytes);
        rows.push_back({kv.second.ms, line});
    }
    std::sort(rows.begin(), rows.end(), [](auto& a, auto& b) { return a.first > b.first; });
    std::string out;
    for (auto& r : rows) out += r.second + "\n";
    return out;
}

static int ilog2(uint64_t x) { int r = 0; while ((1ull << r) < x) r++; return r; }

// ================================================================================================
// options, coin, proof bytes
void ProofOptions::validate() const {
    if (hash_fn != HASH_BLAKE2S_256) fail("ProofOptions: only Blake2s_256 (id 4) is implemented on this path", ST_UNSUPPORTED);
    if (field_extension != EXT_NONE && field_extension != EXT_QUADRATIC) fail("ProofOptions: field extension must be None (1) or Quadratic (2)", ST_UNSUPPORTED);
    if (blowup_factor < 2 || (blowup_factor & (blowup_factor - 1)) || blowup_factor > 128) fail("ProofOptions: blowup factor must be a power of two in [2, 128]");
    if (fri_folding_factor != 2 && fri_folding_factor != 4 && fri_folding_factor != 8 && fri_folding_factor != 16) fail("ProofOptions: FRI folding factor must be 2, 4, 8 or 16");
    if (num_queries == 0) fail("ProofOptions: at least one query is required");
    if (grinding_factor > 32) fail("ProofOptions: grinding factor above 32 is not supported");
    if (fri_log_max_remainder > 16) fail("ProofOptions: FRI max remainder too large");
}

HostCoin HostCoin::from_elements(const uint64_t* e, uint32_t n) {
    HostCoin c;
    c.seed = b2s::hash32(b2s::hash_elements(e, n));   // random.cairo:254-280 then :31-37
    c.counter = 0;
    return c;
}
static inline uint64_t le64(const Digest& d, int word) { return (uint64_t)d.w[word] | ((uint64_t)d.w[word + 1] << 32); }
uint64_t HostCoin::draw_base() {
    for (int i = 0; i < 1000; i++) {
        Digest d = next();
        uint64_t v = le64(d, 0);
        if (v < gl::P) return v;
    }
    fail("random coin: failed to draw a field element", ST_INTERNAL);
}
gl::E2 HostCoin::draw_quad() {
    for (int i = 0; i < 1000; i++) {
        Digest d = next();
        uint64_t a = le64(d, 0), b = le64(d, 2);
        if (a < gl::P && b < gl::P) return gl::E2{a, b};
    }
    fail("random coin: failed to draw a field element", ST_INTERNAL);
}
std::vector<uint64_t> HostCoin::draw_integers(size_t k, uint64_t domain) {
    std::vector<uint64_t> out;
    for (int i = 0; i < 1000 && out.size() < k; i++) {
        uint64_t v = le64(next(), 0) & (domain - 1);
        if (std::find(out.begin(), out.end(), v) == out.end()) out.push_back(v);
    }
    if (out.size() != k) fail("random coin: failed to draw query positions", ST_INTERNAL);
    return out;
}

static void w8(Bytes& b, uint8_t v) { b.push_back(v); }
static void w16(Bytes& b, size_t v) {
    if (v > 0xffff) fail("proof section exceeds its u16 length prefix", ST_UNSUPPORTED);
    b.push_back(v & 0xff); b.push_back((v >> 8) & 0xff);
}
static void w32(Bytes& b, size_t v) { for (int i = 0; i < 4; i++) b.push_back((v >> (8 * i)) & 0xff); }
static void w64(Bytes& b, uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((v >> (8 * i)) & 0xff); }
static void wb(Bytes& b, const Bytes& s) { b.insert(b.end(), s.begin(), s.end()); }
static void wdigest(Bytes& b, const Digest& d) { const uint8_t* p = reinterpret_cast<const uint8_t*>(d.w); b.insert(b.end(), p, p + 32); }

Bytes StarkProof::to_bytes() const {
    Bytes b;
    w8(b, main_width); w8(b, aux_width); w8(b, aux_rands); w8(b, log_n);
    w16(b, 0);                                   // trace meta
    w8(b, 8); w64(b, gl::P);                     // field modulus
    w8(b, options.num_queries); w8(b, options.blowup_factor); w8(b, options.grinding_factor); w8(b, options.hash_fn);
    w8(b, options.field_extension); w8(b, options.fri_folding_factor); w8(b, options.fri_log_max_remainder);
    w16(b, commitments.size()); wb(b, commitments);
    for (auto& q : trace_queries) { w32(b, q.values.size()); wb(b, q.values); w32(b, q.paths.size()); wb(b, q.paths); }
    w32(b, constraint_queries.values.size()); wb(b, constraint_queries.values);
    w32(b, constraint_queries.paths.size()); wb(b, constraint_queries.paths);
    w16(b, ood_trace_states.size()); wb(b, ood_trace_states);
    w16(b, ood_evaluations.size()); wb(b, ood_evaluations);
    w8(b, (uint8_t)fri_layers.size());
    for (auto& q : fri_layers) { w32(b, q.values.size()); wb(b, q.values); w32(b, q.paths.size()); wb(b, q.paths); }
    w16(b, fri_remainder.size()); wb(b, fri_remainder);
    w8(b, 0);                                    // log2(num_partitions)
    w64(b, pow_nonce);
    return b;
}

std::vector<uint64_t> fold_positions(const std::vector<uint64_t>& positions, uint64_t source_domain, uint64_t folding_factor) {
    uint64_t target = source_domain / folding_factor;
    std::vector<uint64_t> out;
    for (uint64_t p : positions) {
        uint64_t q = p % target;
        if (std::find(out.begin(), out.end(), q) == out.end()) out.push_back(q);
    }
    return out;
}
int num_fri_layers(uint64_t domain, uint64_t fold, uint64_t max_remainder) {
    int r = 0;
    while (domain > max_remainder) { domain /= fold; r++; }
    return r;
}

std::vector<std::vector<uint64_t>> batch_proof_indices(size_t n, const std::vector<uint64_t>& positions) {
    // sorted copies instead of tree sets: this runs on the critical path of every proof (once per committed tree)
    std::vector<uint64_t> qs(positions);
    std::sort(qs.begin(), qs.end());
    if (std::adjacent_find(qs.begin(), qs.end()) != qs.end()) fail("batch opening: duplicate positions");
    if (!qs.empty() && qs.back() >= n) fail("batch opening: position out of range");
    std::vector<uint64_t> norm;                       // even partners, sorted, unique
    norm.reserve(qs.size());
    for (uint64_t p : qs) { const uint64_t e = p - (p & 1); if (norm.empty() || norm.back() != e) norm.push_back(e); }
    int depth = 0;
    while (((size_t)1 << depth) < n) depth++;
    std::vector<std::vector<uint64_t>> nodes(norm.size());
    std::vector<uint64_t> next;
    next.reserve(norm.size());
    for (size_t v = 0; v < norm.size(); v++) {
        const uint64_t e = norm[v];
        nodes[v].reserve(depth + 1);
        for (uint64_t i = e; i < e + 2; i++) if (!std::binary_search(qs.begin(), qs.end(), i)) nodes[v].push_back(n + i);
        next.push_back((e + n) >> 1);
    }
    std::vector<uint64_t> idx;
    idx.reserve(norm.size());
    for (int lvl = 1; lvl < depth; lvl++) {
        idx.swap(next);
        next.clear();
        size_t i = 0;
        while (i < idx.size()) {
            uint64_t sib = idx[i] ^ 1;
            if (i + 1 < idx.size() && idx[i + 1] == sib) i += 1;
            else nodes[i].push_back(sib);
            next.push_back(sib >> 1);
            i += 1;
        }
    }
    return nodes;
}

BatchPlan batch_proof_plan(size_t n, const std::vector<uint64_t>& positions) {
    std::vector<uint64_t> qs(positions);
    std::sort(qs.begin(), qs.end());
    if (std::adjacent_find(qs.begin(), qs.end()) != qs.end()) fail("batch opening: duplicate positions");
    if (!qs.empty() && qs.back() >= n) fail("batch opening: position out of range");
    int depth = 0;
    while (((size_t)1 << depth) < n) depth++;
    BatchPlan pl;
    pl.cap = (size_t)depth + 2;
    std::vector<uint64_t> cur, next;
    cur.reserve(qs.size()); next.reserve(qs.size());
    for (uint64_t p : qs) { const uint64_t e = p - (p & 1); if (cur.empty() || cur.back() != e) cur.push_back(e); }   // even partners
    const size_t P = cur.size();
    if (P > 255) fail("batch opening: too many paths", ST_UNSUPPORTED);
    pl.count.assign(P, 0);
    pl.idx.resize(P * pl.cap);
    for (size_t v = 0; v < P; v++) {
        const uint64_t e = cur[v];
        for (uint64_t i = e; i < e + 2; i++) if (!std::binary_search(qs.begin(), qs.end(), i)) pl.idx[v * pl.cap + pl.count[v]++] = n + i;
        cur[v] = (e + n) >> 1;
    }
    for (int lvl = 1; lvl < depth; lvl++) {
        next.clear();
        size_t i = 0;
        while (i < cur.size()) {
            const uint64_t sib = cur[i] ^ 1;
            if (i + 1 < cur.size() && cur[i + 1] == sib) i += 1;
            else pl.idx[i * pl.cap + pl.count[i]++] = sib;       // the quirk of the reference format: indexed by position in THIS level's list
            next.push_back(sib >> 1);
            i += 1;
        }
        cur.swap(next);
    }
    return pl;
}
static Bytes serialize_plan(const BatchPlan& pl, const Digest* digests) {
    Bytes out;
    out.reserve(1 + pl.paths() + 32 * pl.total());
    out.push_back((uint8_t)pl.paths());
    size_t k = 0;
    for (size_t p = 0; p < pl.paths(); p++) {
        out.push_back(pl.count[p]);
        const uint8_t* src = reinterpret_cast<const uint8_t*>(digests + k);
        out.insert(out.end(), src, src + 32 * (size_t)pl.count[p]);
        k += pl.count[p];
    }
    return out;
}

// digests for a batch opening gathered from the device tree -> serialised BatchMerkleProof nodes
static Bytes serialize_batch(const std::vector<std::vector<uint64_t>>& idx, const Digest* digests) {
    Bytes out;
    if (idx.size() > 255) fail("batch opening: too many paths", ST_UNSUPPORTED);
    out.push_back((uint8_t)idx.size());
    size_t k = 0;
    for (auto& v : idx) {
        if (v.size() > 255) fail("batch opening: too many nodes", ST_UNSUPPORTED);
        out.push_back((uint8_t)v.size());
        for (size_t i = 0; i < v.size(); i++) wdigest(out, digests[k++]);
    }
    return out;
}

Bytes open_batch(Context* ctx, const MerkleTree& tree, const std::vector<uint64_t>& positions) {
    auto idx = batch_proof_indices(tree.n, positions);
    // stored nodes are gathered, unstored low nodes are recomputed from the leaf source; both land in one buffer
    std::vector<uint64_t> flat, order;          // flat = [stored indices..., low indices...]; order[k] = slot of item k
    std::vector<uint64_t> hi, lo;
    for (auto& v : idx) for (uint64_t i : v) (i < tree.stored_limit() ? hi : lo).push_back(i);
    {
        size_t nh = 0, nl = 0;
        for (auto& v : idx) for (uint64_t i : v) order.push_back(i < tree.stored_limit() ? nh++ : hi.size() + nl++);
    }
    flat = hi;
    flat.insert(flat.end(), lo.begin(), lo.end());
    std::vector<Digest> raw(flat.size()), got(flat.size());
    if (!flat.empty()) {
        if (!lo.empty() && tree.src_kind == 0) fail("batch opening: tree has unstored levels but no leaf source", ST_INTERNAL);
        DevBuf<uint64_t> d_idx(ctx, flat.size());
        DevBuf<Digest> d_out(ctx, flat.size());
        AERO_HIP(hipMemcpyAsync(d_idx.get(), flat.data(), flat.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        if (!hi.empty()) launch_gather_digests(ctx, tree.nodes.get(), d_idx.get(), (int)hi.size(), d_out.get());
        if (!lo.empty()) {
            if (tree.src_kind == 1) ctx->merkle_recompute(tree.row_src, tree.n, d_idx.get() + hi.size(), (int)lo.size(), d_out.get() + hi.size());
            else ctx->merkle_recompute(tree.fri_src, tree.n, d_idx.get() + hi.size(), (int)lo.size(), d_out.get() + hi.size());
        }
        AERO_HIP(hipMemcpyAsync(raw.data(), d_out.get(), flat.size() * sizeof(Digest), hipMemcpyDeviceToHost, ctx->stream));
        ctx->sync();
        for (size_t k = 0; k < order.size(); k++) got[k] = raw[order[k]];
    }
    Bytes out;
    if (idx.size() > 255) fail("batch opening: too many paths", ST_UNSUPPORTED);
    out.push_back((uint8_t)idx.size());
    size_t k = 0;
    for (auto& v : idx) {
        if (v.size() > 255) fail("batch opening: too many nodes", ST_UNSUPPORTED);
        out.push_back((uint8_t)v.size());
        for (size_t i = 0; i < v.size(); i++) wdigest(out, got[k++]);
    }
    return out;
}

// ================================================================================================
// stage-level API

Matrix Prover::interpolate_columns(const uint64_t* trace_dev, uint32_t width, int log_n) {
    size_t n = (size_t)1 << log_n;
    Matrix polys(ctx_, (int)width, n);
    // coefficient i scaled by 7^i so the LDE needs no coset-shift pass (see ntt.hip); the first pass reads the caller's matrix, nothing is copied
    ctx_->ntt_inverse(polys.data.get(), n, (int)width, log_n, 1, gl::GEN, 1, 0, nullptr, trace_dev, n);
    return polys;
}
Matrix Prover::evaluate_columns_over(const Matrix& polys, int log_blowup) {
    int log_n = ilog2(polys.rows);
    size_t N = polys.rows << log_blowup;
    Matrix lde(ctx_, polys.cols, N);
    ctx_->ntt_forward(polys.data.get(), polys.rows, lde.data.get(), N, polys.cols, log_n + log_blowup, log_blowup);
    return lde;
}
MerkleTree Prover::commit_to_rows(const Matrix& lde, bool keep_low_levels) {
    const RowSrc src{lde.data.get(), lde.rows, lde.cols};
    // narrow rows: fused leaf hashing + 3 levels per thread; wide rows: coalesced row-hash pass, then the tree
    const bool fused = lde.cols <= 4 && lde.rows >= ((size_t)1 << 18);
    const int skip = (fused && !keep_low_levels && low_level_skip) ? 3 : 0;   // never stores 7/8 of the digests
    MerkleTree t(ctx_, lde.rows, skip);
    if (skip) { t.src_kind = 1; t.row_src = src; }
    // the launch that produces the root stores it a second time, in mapped pinned memory: the host waits for the stream and reads it there
    // (no copy kernel between the tree and the transcript); a tree so small that no upper-level launch runs is fetched the old way
    Digest* h_root = (Digest*)ctx_->stage_alloc(sizeof(Digest));
    CoinStep cs;
    cs.root_out = ctx_->stage_dev(h_root);
    cs.flag_seq = ctx_->next_flag(); cs.flag_out = ctx_->flag_dev;
    const bool mapped = (fused ? lde.rows / 8 : lde.rows) > 1;
    if (fused) {
        ctx_->merkle_commit(src, lde.rows, t.nodes.get(), skip, mapped ? &cs : nullptr);
    } else {
        ctx_->hash_rows(lde.data.get(), lde.rows, lde.cols, lde.rows, t.leaves());
        ctx_->merkle_build(t.nodes.get(), t.n, mapped ? &cs : nullptr);
    }
    if (mapped) { ctx_->wait_flag(cs.flag_seq); t.root_host = *h_root; }
    else ctx_->fetch(&t.root_host, t.nodes.get() + 1, sizeof(Digest));
    return t;
}
MerkleTree Prover::commit_fri_layer(const FriSrc& src, bool keep_low_levels) {
    const size_t rows = src.rows;
    if (rows < 2) {   // a single leaf is its own root
        MerkleTree t(ctx_, 1);
        ctx_->hash_fri_rows(src, t.nodes.get() + 1);
        AERO_HIP(hipMemcpyAsync(&t.root_host, t.nodes.get() + 1, sizeof(Digest), hipMemcpyDeviceToHost, ctx_->stream));
        ctx_->sync();
        return t;
    }
    const bool fused = false;   // FRI layers are at most N/8 rows of 4+ compressions: one row per lane keeps the chip busy
    const int skip = 0;
    (void)keep_low_levels;
    MerkleTree t(ctx_, rows, skip);
    if (skip) { t.src_kind = 2; t.fri_src = src; }
    Digest* h_root = (Digest*)ctx_->stage_alloc(sizeof(Digest));
    CoinStep cs;
    cs.root_out = ctx_->stage_dev(h_root);
    if (fused) {
        ctx_->merkle_commit(src, rows, t.nodes.get(), skip, &cs);
    } else {
        ctx_->hash_fri_rows(src, t.leaves());
        ctx_->merkle_build(t.nodes.get(), rows, &cs);        // rows >= 2: the launch that writes node 1 exists
    }
    ctx_->sync();
    t.root_host = *h_root;
    return t;
}

MerkleTree Prover::commit_fri_layer_async(const FriSrc& src, const CoinStep* coin) {
    const size_t rows = src.rows;
    if (rows < 2) {   // a single leaf is its own root
        MerkleTree t(ctx_, 1);
        ctx_->hash_fri_rows(src, t.nodes.get() + 1);
        if (coin) {
            if (coin->deg == 1) launch_fri_coin_step<FB>(ctx_, coin->seed_io, t.nodes.get() + 1, coin->alpha_out);
            else launch_fri_coin_step<FQ>(ctx_, coin->seed_io, t.nodes.get() + 1, reinterpret_cast<gl::E2*>(coin->alpha_out));
            // root_out / seed_out may be mapped pinned memory (the block the host reads after the last layer): hipMemcpyDefault
            if (coin->root_out) AERO_HIP(hipMemcpyAsync(coin->root_out, t.nodes.get() + 1, sizeof(Digest), hipMemcpyDefault, ctx_->stream));
            if (coin->seed_out) AERO_HIP(hipMemcpyAsync(coin->seed_out, coin->seed_io, sizeof(Digest), hipMemcpyDefault, ctx_->stream));
        }
        return t;
    }
    MerkleTree t(ctx_, rows, 0);
    ctx_->hash_fri_rows(src, t.leaves());
    ctx_->merkle_build(t.nodes.get(), rows, coin);
    return t;
}

// ================================================================================================
template <class F> static void flatten(const typename F::T* v, size_t n, std::vector<uint64_t>& out) {
    for (size_t i = 0; i < n; i++) for (int d = 0; d < F::DEG; d++) out.push_back(F::comp(v[i], d));
}
template <class F> static Digest hash_e(const typename F::T* v, size_t n) {
    std::vector<uint64_t> f;
    flatten<F>(v, n, f);
    return b2s::hash_elements(f.data(), (uint32_t)f.size());
}
struct StageClock {
    Context* ctx; bool on; std::chrono::steady_clock::time_point t0;
    StageClock(Context* c, bool enable) : ctx(c), on(enable) { if (on) { ctx->sync(); } t0 = std::chrono::steady_clock::now(); }
    double lap() {
        if (on) ctx->sync();
        auto t1 = std::chrono::steady_clock::now();
        double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        t0 = t1;
        return ms;
    }
};

// AERO_HOST_GAPS=1 (diagnosis): host wall-clock between the points of one proof where the device waits for the host or the host for the device,
// one line per proof on stderr. "x->y" = time from mark x to mark y.
struct GapLog {
    bool on;
    std::vector<std::pair<const char*, std::chrono::steady_clock::time_point>> marks;
    GapLog() { static const bool e = getenv("AERO_HOST_GAPS") != nullptr; on = e; }
    void mark(const char* what) { if (on) marks.emplace_back(what, std::chrono::steady_clock::now()); }
    ~GapLog() {
        if (!on || marks.size() < 2) return;
        std::string line = "host gaps (us):";
        char buf[96];
        for (size_t i = 1; i < marks.size(); i++) {
            snprintf(buf, sizeof buf, " %s->%s %.1f", marks[i - 1].first, marks[i].first, std::chrono::duration<double, std::micro>(marks[i].second - marks[i - 1].second).count());
            line += buf;
        }
        fprintf(stderr, "%s\n", line.c_str());
    }
};

// ---- exchange helpers (sharded proof only) -----------------------------------------------------------
void Prover::comm_all_to_all(const void* send, void* recv, size_t bytes) {
    if (!comm_.stream_ordered) ctx_->sync();
    if (!comm_.all_to_all || comm_.all_to_all(comm_.user, send, recv, bytes) != 0) fail("sharded prove: all_to_all exchange failed", ST_COMM);
}
void Prover::comm_all_gather(const void* send, void* recv, size_t bytes) {
    if (!comm_.stream_ordered) ctx_->sync();
    if (!comm_.all_gather || comm_.all_gather(comm_.user, send, recv, bytes) != 0) fail("sharded prove: all_gather exchange failed", ST_COMM);
}
void Prover::comm_all_reduce(uint64_t* buf, size_t count) {
    if (!comm_.stream_ordered) ctx_->sync();
    if (!comm_.all_reduce_sum_u64 || comm_.all_reduce_sum_u64(comm_.user, buf, count) != 0) fail("sharded prove: all_reduce exchange failed", ST_COMM);
}
void Prover::comm_send_recv(const void* send, int to, void* recv, int from, size_t bytes) {
    if (!comm_.stream_ordered) ctx_->sync();
    if (!comm_.send_recv || comm_.send_recv(comm_.user, send, to, recv, from, bytes) != 0) fail("sharded prove: send_recv exchange failed", ST_COMM);
}
// Leaf digests of this rank's coset (local leaf t = global leaf t*G + rank) -> every rank ends up with the digests of the
// contiguous global range [rank*L, (rank+1)*L), builds that subtree, and the G subtree roots are all-gathered; the top
// log2 G levels are hashed on the host by every rank.
Commitment Prover::commit_exchange(DevBuf<Digest>& local, size_t L) {
    Context* ctx = ctx_;
    const int G = comm_.world;
    if (L < (size_t)G || L % G) fail("sharded prove: commitment too small for this many ranks");
    Commitment c;
    c.sharded = true;
    c.n_global = L * G;
    c.tree = MerkleTree(ctx, L);
    if (L >= 8) {
        // the pieces land straight in the leaf level (arrival order); the tree is built through the leaf permutation
        comm_all_to_all(local.get(), c.tree.leaves(), (L / G) * sizeof(Digest));   // chunk r = my leaves t in [r*L/G, (r+1)*L/G)
        c.leaf_parts_log = ilog2(G);
        ctx->merkle_build_parts(c.tree.nodes.get(), L, c.leaf_parts_log);
    } else {
        DevBuf<Digest> recv(ctx, L);
        comm_all_to_all(local.get(), recv.get(), (L / G) * sizeof(Digest));
        if (L >= 2) {
            launch_interleave_digests(ctx, recv.get(), L / G, c.tree.leaves(), G, L / G);
            ctx->merkle_build(c.tree.nodes.get(), L);
        } else {
            AERO_HIP(hipMemcpyAsync(c.tree.nodes.get() + 1, recv.get(), sizeof(Digest), hipMemcpyDeviceToDevice, ctx->stream));
        }
    }
    finish_exchange(c);
    return c;
}
// The same commitment when a row is SHORTER than its digest (8 * columns < 32 bytes: the 2-column trace and composition matrices
// of the Fibonacci workloads): the ranks exchange the rows and each hashes the leaves of its own contiguous range - half the bytes
// on the links, the same number of compressions per rank.
Commitment Prover::commit_exchange_rows(const Matrix& lde) {
    Context* ctx = ctx_;
    const int G = comm_.world;
    const size_t L = lde.rows, per = L / G, W = (size_t)lde.cols;
    Commitment c;
    c.sharded = true;
    c.n_global = L * G;
    c.tree = MerkleTree(ctx, L);
    DevBuf<uint64_t> send(ctx, W * L), recv(ctx, W * L);
    // chunk p = my rows [p * per, (p + 1) * per) of every column, column-major inside the chunk
    for (int p = 0; p < G; p++)
        AERO_HIP(hipMemcpy2DAsync(send.get() + (size_t)p * W * per, per * 8, lde.data.get() + (size_t)p * per, L * 8, per * 8, W, hipMemcpyDeviceToDevice, ctx->stream));
    comm_all_to_all(send.get(), recv.get(), W * per * 8);
    // chunk q holds rows of rank q's coset: global leaf u G + q of my range -> leaf slot q per + u (arrival order, as for digests)
    for (int q = 0; q < G; q++) ctx->hash_rows(recv.get() + (size_t)q * W * per, per, (int)W, per, c.tree.leaves() + (size_t)q * per);
    c.leaf_parts_log = ilog2(G);
    ctx->merkle_build_parts(c.tree.nodes.get(), L, c.leaf_parts_log);
    finish_exchange(c);
    return c;
}
// Chunked form of the two commitments above ("all-to-all issued per chunk behind the row hashing of the next chunk"). Every
// peer's piece of `per` leaves is cut into K sub-pieces; sub-piece k of ALL peers travels in one all-to-all of per / K leaves (or rows) per
// peer. The exchanges stay on the proving stream S, where the communicator enqueues them (RCCL and the in-process group are stream-ordered
// on the context's stream); the row hashing runs on the context's second stream H, tied to S by events:
//   digests:  H  hash(0) e0 | hash(1) e1 | hash(2) ...          rows:  S  pack(0) a2a(0) f0 | pack(1) a2a(1) f1 | ...
//             S  wait e0, a2a(0), place(0) | wait e1, a2a(1) ...        H  wait f0, hash(0) | wait f1, hash(1) | ...
// so piece k + 1 is hashed while piece k is on the links (digest path), or piece k is hashed while piece k + 1 is on the links (row path).
// The leaves land in the same arrival-order slots as in the unchunked forms: the tree, the openings and the proof bytes do not change. On
// one GPU (ranks sharing the device) this can only be checked for parity; whether it pays is a question for a node with links.
Commitment Prover::commit_exchange_chunked(const Matrix& lde, int K, bool rows_path) {
    Context* ctx = ctx_;
    const int G = comm_.world;
    const size_t L = lde.rows, per = L / G, W = (size_t)lde.cols, sub = per / (size_t)K;
    if (L % G || per % (size_t)K || sub == 0) fail("sharded prove: commitment does not divide into this many exchange chunks", ST_INTERNAL);
    Commitment c;
    c.sharded = true;
    c.n_global = L * G;
    c.tree = MerkleTree(ctx, L);
    hipStream_t S = ctx->stream, H = ctx->get_copy_stream();
    struct OnStream { Context* c; hipStream_t old; OnStream(Context* c_, hipStream_t s) : c(c_), old(c_->stream) { c->stream = s; } ~OnStream() { c->stream = old; } };
    const size_t ev0 = 40;      // events of this function (the pipelined hand-over and the pool prefetch use lower ones)
    AERO_HIP(hipEventRecord(ctx->sync_event(ev0), S));                 // H starts behind everything S carries (the LDE)
    AERO_HIP(hipStreamWaitEvent(H, ctx->sync_event(ev0), 0));
    if (!rows_path) {
        // send block k: [peer r][sub digests] contiguous, as the all-to-all wants it; receive block k likewise, then placed with one strided copy
        DevBuf<Digest> send(ctx, (size_t)K * G * sub), recv(ctx, (size_t)K * G * sub);
        for (int k = 0; k < K; k++) {
            {
                OnStream h(ctx, H);
                for (int r = 0; r < G; r++)
                    ctx->hash_rows(lde.data.get() + (size_t)r * per + (size_t)k * sub, L, (int)W, sub, send.get() + ((size_t)k * G + r) * sub);
            }
            AERO_HIP(hipEventRecord(ctx->sync_event(ev0 + 1 + (k & 1)), H));
            AERO_HIP(hipStreamWaitEvent(S, ctx->sync_event(ev0 + 1 + (k & 1)), 0));
            comm_all_to_all(send.get() + (size_t)k * G * sub, recv.get() + (size_t)k * G * sub, sub * sizeof(Digest));
            // piece q of block k -> leaf slots [q per + k sub, q per + (k + 1) sub)
            AERO_HIP(hipMemcpy2DAsync(c.tree.leaves() + (size_t)k * sub, per * sizeof(Digest), recv.get() + (size_t)k * G * sub, sub * sizeof(Digest),
                                      sub * sizeof(Digest), G, hipMemcpyDeviceToDevice, S));
        }
    } else {
        DevBuf<uint64_t> send(ctx, W * L), recv(ctx, W * L);
        for (int k = 0; k < K; k++) {
            uint64_t* sk = send.get() + (size_t)k * G * W * sub;
            uint64_t* rk = recv.get() + (size_t)k * G * W * sub;
            // block k, chunk p = my rows [p per + k sub, p per + (k + 1) sub) of every column, column-major inside the chunk
            for (int p = 0; p < G; p++)
                AERO_HIP(hipMemcpy2DAsync(sk + (size_t)p * W * sub, sub * 8, lde.data.get() + (size_t)p * per + (size_t)k * sub, L * 8, sub * 8, W, hipMemcpyDeviceToDevice, S));
            comm_all_to_all(sk, rk, W * sub * 8);
            AERO_HIP(hipEventRecord(ctx->sync_event(ev0 + 1 + (k & 1)), S));
            AERO_HIP(hipStreamWaitEvent(H, ctx->sync_event(ev0 + 1 + (k & 1)), 0));
            OnStream h(ctx, H);
            for (int q = 0; q < G; q++) ctx->hash_rows(rk + (size_t)q * W * sub, sub, (int)W, sub, c.tree.leaves() + (size_t)q * per + (size_t)k * sub);
        }
        // `send` / `recv` go back to the allocator when this scope ends: everything H still reads of them must be behind S by then (below)
        AERO_HIP(hipEventRecord(ctx->sync_event(ev0 + 3), H));
        AERO_HIP(hipStreamWaitEvent(S, ctx->sync_event(ev0 + 3), 0));
        c.leaf_parts_log = ilog2(G);
        ctx->merkle_build_parts(c.tree.nodes.get(), L, c.leaf_parts_log);
        finish_exchange(c);
        return c;
    }
    AERO_HIP(hipEventRecord(ctx->sync_event(ev0 + 3), H));
    AERO_HIP(hipStreamWaitEvent(S, ctx->sync_event(ev0 + 3), 0));
    c.leaf_parts_log = ilog2(G);
    ctx->merkle_build_parts(c.tree.nodes.get(), L, c.leaf_parts_log);
    finish_exchange(c);
    return c;
}
// all-gather of the G subtree roots (the only collective on the transcript's critical path); the top log2 G levels on the host
void Prover::finish_exchange(Commitment& c) {
    Context* ctx = ctx_;
    const int G = comm_.world;
    DevBuf<Digest> roots(ctx, G);
    comm_all_gather(c.tree.nodes.get() + 1, roots.get(), sizeof(Digest));
    c.top.assign(2 * (size_t)G, Digest{});
    AERO_HIP(hipMemcpyAsync(c.top.data() + G, roots.get(), G * sizeof(Digest), hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    for (int i = G - 1; i >= 1; i--) c.top[i] = b2s::merge(c.top[2 * i], c.top[2 * i + 1]);
    c.root = c.top[1];
    c.tree.root_host = c.top[G + comm_.rank];
}

// The constraint evaluations this rank needs from its peers: its constraint domain h_k<w_ce> is the union of `q` cosets, owned by
// the ranks (rank + t * step) mod G, t < q. With a pairwise exchange only those q - 1 pieces travel (2^24 x 2 over 8 GPUs: one
// piece of 128 MiB instead of seven); a communicator without one all-gathers. all[owner * bytes ...] holds the pieces.
void Prover::gather_h_cosets(const uint64_t* mine, uint64_t* all, size_t bytes, int q, int step) {
    const int G = comm_.world, rank = comm_.rank;
    if (!comm_.send_recv || q >= G) { comm_all_gather(mine, all, bytes); return; }
    uint8_t* base = reinterpret_cast<uint8_t*>(all);
    AERO_HIP(hipMemcpyAsync(base + (size_t)rank * bytes, mine, bytes, hipMemcpyDeviceToDevice, ctx_->stream));
    for (int t = 1; t < q; t++) {
        const int from = (rank + t * step) % G, to = ((rank - t * step) % G + G) % G;
        comm_send_recv(mine, to, base + (size_t)from * bytes, from, bytes);
    }
}

// ---- stage functions shared by prove_impl and the C ABI ---------------------------------------------------
void Prover::composition_from_evaluations(uint64_t* hbuf, int deg, int log_ce, int log_c, uint64_t h) {
    // interpolate over the coset: coefficient I gets h^-I (coset) * h^(I >> log C) (pre-scaling of column coefficient
    // i = I >> log C for the column LDE). In bit-reversed order the C column polynomials are the C contiguous chunks of
    // the buffer: no split pass (H(x) = sum_c x^c H_c(x^C)).
    ctx_->ntt_inverse(hbuf, (size_t)1 << log_ce, deg, log_ce, 1, gl::inv(h), h, log_c);
}

template <class F>
DevBuf<uint64_t> Prover::deep_compose(const uint64_t* tlde, const uint64_t* clde, const uint64_t* alde, uint32_t W, uint32_t A, uint32_t C,
                                      int log_n, int log_bl, uint64_t h, const DeepInputs<F>& in, const DeepCompact* compact) {
    typedef typename F::T T;
    Context* ctx = ctx_;
    const size_t n = (size_t)1 << log_n, M = n << log_bl;
    const int log_M = log_n + log_bl;
    if (in.ood_cur.size() != W + A || in.ood_next.size() != W + A || in.ood_h.size() != C || in.da.size() != W + A ||
        in.db.size() != W + A || in.dg.size() != W + A || in.dc.size() != C)
        fail("deep_compose: coefficient / OOD vector sizes do not match the trace shape");
    // deg(DEEP) < n: evaluate it on the n-point coset h<w_n> (every (M/n)-th LDE row), interpolate (the plain inverse
    // transform of values on h<w_n> yields exactly the h^i-prescaled coefficients), extend like any other column.
    DevBuf<uint64_t> out(ctx, (size_t)F::DEG * M);
    NttTables* tM = ctx->ntt_tables(log_M);
    DevBuf<uint64_t> dsm(ctx, (size_t)F::DEG * n);
    DeepArgs<F> a{};
    a.tlde = tlde; a.clde = clde; a.alde = alde; a.A = A; a.N = M; a.count = n; a.row_step = (uint32_t)1 << log_bl; a.W = W; a.C = C;
    a.t_stride = a.c_stride = a.a_stride = M;
    a.t_step = a.c_step = a.a_step = a.row_step;
    if (compact) {
        // a compact copy holds every 2^k-th row (k <= log_bl): stride M >> k, step row_step >> k
        // (a column stride given explicitly = the copy is de-interleaved and its part 0 is what is read)
        if (compact->t && compact->t_log <= log_bl) { a.tlde = compact->t; a.t_stride = compact->t_stride ? compact->t_stride : M >> compact->t_log; a.t_step = a.row_step >> compact->t_log; }
        if (compact->c && compact->c_log <= log_bl) { a.clde = compact->c; a.c_stride = M >> compact->c_log; a.c_step = a.row_step >> compact->c_log; }
        if (A && compact->a && compact->a_log <= log_bl) { a.alde = compact->a; a.a_stride = compact->a_stride ? compact->a_stride : M >> compact->a_log; a.a_step = a.row_step >> compact->a_log; }
    }
    a.tw_lo = tM->lo_fwd; a.tw_hi = tM->hi_fwd; a.tw_h = tM->h; a.offset = h;
    a.z = in.z; a.z_next = F::mulb(in.z, gl::root_of_unity(log_n)); a.z_c = gl::fpow<F>(in.z, C); a.z_conj = F::conj(in.z);
    a.lambda = in.lambda; a.mu = in.mu;
    ParamPack pp(ctx);
    const size_t i0 = pp.add(in.ood_cur), i1 = pp.add(in.ood_next), i2 = pp.add(in.ood_h), i3 = pp.add(in.da), i4 = pp.add(in.db), i5 = pp.add(in.dg), i6 = pp.add(in.dc);
    pp.commit();
    a.ood_cur = pp.ptr<T>(i0); a.ood_next = pp.ptr<T>(i1); a.ood_h = pp.ptr<T>(i2);
    a.da = pp.ptr<T>(i3); a.db = pp.ptr<T>(i4); a.dg = pp.ptr<T>(i5); a.dc = pp.ptr<T>(i6);
    if (log_bl == 0) {
        // a shard of exactly n rows (world = blowup): the n-point coset IS the local domain - no interpolation, no extension
        for (int d = 0; d < F::DEG; d++) a.out[d] = out.get() + (size_t)d * n;
        launch_deep<F>(ctx, a);
        return out;
    }
    for (int d = 0; d < F::DEG; d++) a.out[d] = dsm.get() + (size_t)d * n;
    launch_deep<F>(ctx, a);
    ctx->ntt_inverse(dsm.get(), n, F::DEG, log_n, 1, 1, 1, 0);
    ctx->ntt_forward(dsm.get(), n, out.get(), M, F::DEG, log_M, log_bl);
    return out;
}
template DevBuf<uint64_t> Prover::deep_compose<FB>(const uint64_t*, const uint64_t*, const uint64_t*, uint32_t, uint32_t, uint32_t, int, int, uint64_t, const DeepInputs<FB>&, const Prover::DeepCompact*);
template DevBuf<uint64_t> Prover::deep_compose<FQ>(const uint64_t*, const uint64_t*, const uint64_t*, uint32_t, uint32_t, uint32_t, int, int, uint64_t, const DeepInputs<FQ>&, const Prover::DeepCompact*);

template <class F> FriLayers Prover::fri_build_layers(DevBuf<uint64_t>&& evals, uint64_t N, HostCoin& coin, Bytes* roots) {
    typedef typename F::T T;
    Context* ctx = ctx_;
    const size_t Fd = opt_.fri_folding_factor;
    FriLayers fl;
    fl.lde_size = N; fl.deg = F::DEG; fl.fold = (int)Fd;
    fl.layers = num_fri_layers(N, Fd, 1ull << opt_.fri_log_max_remainder);
    {
        uint64_t rem = N;
        for (int l = 0; l < fl.layers; l++) rem /= Fd;
        if (rem < Fd) fail("FRI: remainder smaller than the folding factor");
    }
    fl.vals.push_back(std::move(evals));
    const uint64_t gen_inv = gl::inv(gl::GEN);
    // The transcript steps of the commit phase (reseed with the root, draw alpha) run on the device, so every layer is enqueued
    // without a host round trip; the roots come back with ONE copy at the end and the host replays the coin to stay in step.
    // the roots (and, after every step, the coin's seed) are ALSO stored in mapped pinned memory by the launches that produce them: after the
    // last layer the host waits for the stream and finds them - no copy. The seed the device steps stays in device memory.
    Digest* h_block = (Digest*)ctx->stage_alloc(sizeof(Digest) * (fl.layers + 2));      // roots of all layers, then the final seed
    Digest* d_roots = ctx->stage_dev(h_block);
    Digest* d_seed = (Digest*)ctx->scratch_alloc(sizeof(Digest));
    T* d_alpha = (T*)ctx->scratch_alloc(sizeof(T) * (fl.layers + 1));
    Digest* h_seed = (Digest*)ctx->stage_alloc(sizeof(Digest));
    *h_seed = coin.seed;
    AERO_HIP(hipMemcpyAsync(d_seed, h_seed, sizeof(Digest), hipMemcpyHostToDevice, ctx->stream));
    std::vector<Digest> h_roots(fl.layers + 2);
    uint64_t dom = N;
    // layers of at most FRI_TAIL_MAX_DOM points are pure latency: they all go into ONE single-workgroup launch (Context::fri_tail)
    int tail0 = fl.layers + 1;
    uint32_t fri_flag = 0;
    if (fri_tail && (Fd == 2 || Fd == 4 || Fd == 8)) {
        uint64_t d = N;
        for (int l = 0; l <= fl.layers; l++, d /= Fd)
            if (d <= (uint64_t)FRI_TAIL_MAX_DOM && d / Fd <= (uint64_t)FRI_TAIL_MAX_ROWS && fl.layers + 1 - l <= FRI_TAIL_MAX_LAYERS) { tail0 = l; break; }
    }
    for (int l = 0; l < tail0; l++) {
        const size_t rows = dom / Fd;
        const FriSrc fsrc{fl.vals[l].get(), fl.vals[l].get() + (F::DEG > 1 ? dom : 0), F::DEG, rows, (int)Fd};
        Commitment c;
        const CoinStep cs{d_seed, reinterpret_cast<uint64_t*>(d_alpha + l), F::DEG, d_roots + l, d_roots + fl.layers + 1};
        c.tree = commit_fri_layer_async(fsrc, &cs);
        c.n_global = rows;
        fl.coms.push_back(std::move(c));
        if (l == fl.layers) break;   // alpha drawn after the remainder commitment is unused
        fl.vals.emplace_back(ctx, (size_t)F::DEG * rows);
        NttTables* td = ctx->ntt_tables(ilog2(dom));
        FoldArgs<F> a{};
        for (int d = 0; d < F::DEG; d++) { a.in[d] = fl.vals[l].get() + (size_t)d * dom; a.out[d] = fl.vals[l + 1].get() + (size_t)d * rows; }
        if (F::DEG == 1) { a.in[1] = a.in[0]; a.out[1] = a.out[0]; }
        a.rows = rows; a.fold = (int)Fd; a.alpha = F::zero(); a.alpha_dev = d_alpha + l;
        a.twi_lo = td->lo_inv; a.twi_hi = td->hi_inv; a.tw_h = td->h;
        a.gen_inv = gen_inv; a.fold_inv = gl::inv(Fd);
        uint64_t wFi = gl::inv(gl::root_of_unity(ilog2(Fd)));
        for (size_t m = 0; m < Fd; m++) a.dft[m] = gl::pow(wFi, m);
        launch_fri_fold<F>(ctx, a);
        dom = rows;
    }
    if (tail0 <= fl.layers) {
        FriTailArgs t{};
        t.deg = F::DEG; t.n_layers = fl.layers + 1 - tail0; t.dom0 = (uint32_t)dom; t.vals0 = fl.vals[tail0].get();
        uint64_t dd = dom;
        for (int i = 0; i < t.n_layers; i++) {
            const size_t rows = dd / Fd;
            Commitment c;
            c.tree = MerkleTree(ctx, rows < 2 ? 1 : rows, 0);
            c.n_global = rows;
            t.nodes[i] = c.tree.nodes.get();
            fl.coms.push_back(std::move(c));
            if (i + 1 < t.n_layers) {
                fl.vals.emplace_back(ctx, (size_t)F::DEG * rows);
                t.vals_out[i] = fl.vals.back().get();
            }
            dd = rows;
        }
        t.flag_seq = fri_flag = ctx->next_flag(); t.flag_out = ctx->flag_dev;
        t.roots_out = d_roots + tail0; t.seed_io = d_seed; t.seed_out = d_roots + fl.layers + 1; t.alphas_out = reinterpret_cast<uint64_t*>(d_alpha + tail0);
        t.gen_inv = gen_inv; t.fold_inv = gl::inv(Fd); t.w_inv0 = gl::inv(gl::root_of_unity(ilog2(dom)));
        ctx->fri_tail(t, (int)Fd);
    }
    if (fri_flag) ctx->wait_flag(fri_flag);      // the tail launch is the last one and a single workgroup: it signals through the completion word
    else ctx->sync();
    memcpy(h_roots.data(), h_block, sizeof(Digest) * (fl.layers + 2));
    *h_seed = h_roots[fl.layers + 1];
    for (int l = 0; l <= fl.layers; l++) {
        Commitment& c = fl.coms[l];
        c.root = h_roots[l];
        c.tree.root_host = c.root;
        if (roots) wdigest(*roots, c.root);
        coin.reseed(c.root);
        (void)coin.draw<F>();
    }
    if (memcmp(coin.seed.w, h_seed->w, sizeof(Digest)) != 0) fail("FRI: device and host transcripts diverged", ST_INTERNAL);
    return fl;
}
template FriLayers Prover::fri_build_layers<FB>(DevBuf<uint64_t>&&, uint64_t, HostCoin&, Bytes*);
template FriLayers Prover::fri_build_layers<FQ>(DevBuf<uint64_t>&&, uint64_t, HostCoin&, Bytes*);

template <class F> Bytes Prover::fri_open(const FriLayers& fl, const std::vector<uint64_t>& positions) {
    Context* ctx = ctx_;
    const size_t Fd = fl.fold;
    Bytes out;
    w8(out, (uint8_t)fl.layers);
    std::vector<uint64_t> fp = positions;
    uint64_t dom = fl.lde_size;
    for (int l = 0; l < fl.layers; l++) {
        fp = fold_positions(fp, dom, Fd);
        const size_t rows = dom / Fd, cnt = fp.size() * Fd * F::DEG;
        DevBuf<uint64_t> d_pos(ctx, fp.size()), d_val(ctx, cnt);
        std::vector<uint64_t> vals(cnt);
        AERO_HIP(hipMemcpyAsync(d_pos.get(), fp.data(), fp.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        launch_gather_fri_rows(ctx, fl.vals[l].get(), fl.vals[l].get() + (F::DEG > 1 ? dom : 0), F::DEG, rows, (int)Fd, d_pos.get(), (int)fp.size(), d_val.get());
        AERO_HIP(hipMemcpyAsync(vals.data(), d_val.get(), cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
        ctx->sync();
        Bytes vb;
        for (uint64_t v : vals) w64(vb, v);
        const Bytes paths = open_batch(ctx, fl.coms[l].tree, fp);
        w32(out, vb.size()); wb(out, vb);
        w32(out, paths.size()); wb(out, paths);
        dom = rows;
    }
    std::vector<uint64_t> rem((size_t)F::DEG * dom);
    AERO_HIP(hipMemcpyAsync(rem.data(), fl.vals[fl.layers].get(), rem.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    Bytes rb;
    for (size_t i = 0; i < dom; i++) for (int d = 0; d < F::DEG; d++) w64(rb, rem[(size_t)d * dom + i]);
    w16(out, rb.size()); wb(out, rb);
    w8(out, 0);
    return out;
}
template Bytes Prover::fri_open<FB>(const FriLayers&, const std::vector<uint64_t>&);
template Bytes Prover::fri_open<FQ>(const FriLayers&, const std::vector<uint64_t>&);

template <class F>
Bytes Prover::prove_impl(const uint64_t* trace_dev, uint32_t W, int log_n, std::vector<uint64_t>* pub_out) {
    typedef typename F::T T;
    Context* ctx = ctx_;
    AERO_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << log_n, B = opt_.blowup_factor, Fd = opt_.fri_folding_factor;
    const int log_B = ilog2(B);
    // the AIR: a program (air_program.hpp) or the built-in FibAir with its optional auxiliary segment
    const air::Program* const prog = program_;
    air::Instance pinst;
    if (prog) {
        if (W != prog->W) fail("prove: the trace does not have the program's main width");
        if (program_pub_.size() != prog->num_pub) fail("prove: wrong number of public inputs for this program");
        for (uint64_t v : program_pub_) if (v >= gl::P) fail("prove: non-canonical public input");
        pinst = air::instantiate(*prog, log_n);
    }
    const uint32_t A = prog ? prog->A : aux_width_, R = prog ? prog->R : aux_rands_, D = prog ? 2 : aux_degree_;
    if (!prog && A && (D < 2 || D > 8)) fail("prove: auxiliary constraint degree must be in [2, 8]");
    FibAir air;
    air.width = W; air.log_n = log_n; air.aux_width = A; air.aux_rands = R; air.aux_degree = D;
    const size_t N = n * B, C = prog ? prog->ce_blowup : air.ce_blowup_factor(), ceN = C * n;
    const int log_N = log_n + log_B, log_ce = ilog2(ceN);
    const size_t n_trans = prog ? prog->num_transition() : air.num_transition_constraints();
    const size_t n_assert = prog ? prog->num_assertions() : air.num_assertions();
    if (!prog && (W < 2 || (W & 1) || W > 254)) fail("prove: FibAir needs an even column count in [2, 254]");
    if (A > 255 - W || (A && (R == 0 || R > 255))) fail("prove: auxiliary segment needs 1..255 random elements and main + aux width <= 255");
    const uint32_t TW = W + A;
    if (B < C) fail("prove: blowup factor smaller than the constraint evaluation blowup");
    if (log_N > gl::TWO_ADICITY) fail("prove: LDE domain exceeds the field's two-adicity (2^32)");
    if (log_n < 3) fail("prove: trace must have at least 8 rows");
    const int layers = num_fri_layers(N, Fd, 1ull << opt_.fri_log_max_remainder);
    {
        uint64_t rem = N;
        for (int l = 0; l < layers; l++) rem /= Fd;
        if (rem < Fd) fail("prove: FRI remainder smaller than the folding factor");
        if (rem * 8 * F::DEG > 0xffff) fail("prove: FRI remainder does not fit the proof's u16 length prefix", ST_UNSUPPORTED);
    }
    // ---- shard geometry: this rank owns LDE rows j = rank (mod G), the coset h <w_M>, h = 7 w_N^rank, M = N / G.
    // Every stage below is the single-GPU algorithm on that coset (offset h, local blowup B / G); G = 1 is the whole domain.
    const int G = comm_.world, rank = comm_.rank;
    if (G < 1 || (G & (G - 1)) || rank < 0 || rank >= G) fail("sharded prove: world must be a power of two and 0 <= rank < world");
    if ((size_t)G > B) fail("sharded prove: more ranks than the blowup factor (each rank owns whole cosets of the trace domain)");
    const int log_G = ilog2(G);
    const size_t M = N / G;
    const int log_M = log_N - log_G, log_Bl = log_B - log_G;
    if (G > 1 && M < (size_t)G) fail("sharded prove: LDE domain too small for this many ranks");
    const uint64_t h = gl::mul(gl::GEN, gl::pow(gl::root_of_unity(log_N), (uint64_t)rank));
    const uint64_t h_inv = gl::inv(h);
    const uint32_t min_peer = comm_.min_peer_digests ? comm_.min_peer_digests : 2048;

    StageMs ms;
    ctx->sync();
    ctx->stage_reset();
    StageClock clk(ctx, collect_stage_times);
    GapLog gaps;
    gaps.mark("start");
    auto t_start = std::chrono::steady_clock::now();

    // 0. AIR, public inputs, channel [proving_worker.rs:248-268]. The public inputs (results[k] = trace[2k+1][n-1]) seed the coin, but
    //    nothing before the first commitment depends on the coin: their read-back is only enqueued here and consumed after the
    //    stream synchronisation the trace commitment performs anyway (one host round trip less per proof).
    air.results.resize(W / 2);
    const bool landed = landed_;                        // host trace already on its way into `trace_dev` (pool prefetch): resident path + event + check
    if (landed) {
        if (G != 1 || !trace_dev || !host_trace_) fail("prove: a landed trace needs one GPU, the landing buffer and the host trace", ST_INTERNAL);
        AERO_HIP(hipStreamWaitEvent(ctx->stream, landed_ready_, 0));
    }
    const uint64_t* const host_trace = landed ? nullptr : host_trace_;     // trace still in HOST memory (pipelined hand-over below), or nullptr
    uint64_t* h_last_row = (uint64_t*)ctx->stage_alloc((size_t)W * 8);
    if (host_trace_) {
        for (uint32_t c = 0; c < W; c++) h_last_row[c] = host_trace_[(size_t)c * n + (n - 1)];
    } else {
        // position and row both live in mapped pinned memory: the gather reads its one position from the host and stores the row where
        // the host reads it after the first commitment's synchronisation (two copy launches less at the start of every proof)
        uint64_t* h_pos = (uint64_t*)ctx->stage_alloc(8);
        *h_pos = n - 1;
        launch_gather_rows(ctx, trace_dev, n, (int)W, ctx->stage_dev(h_pos), 1, ctx->stage_dev(h_last_row));
    }
    StarkProof proof;
    proof.main_width = (uint8_t)W; proof.aux_width = (uint8_t)A; proof.aux_rands = (uint8_t)R; proof.log_n = (uint8_t)log_n; proof.options = opt_;
    const uint64_t g = gl::root_of_unity(log_n);
    const uint64_t gen_inv = gl::inv(gl::GEN);

    // commitment to the rows of a (local) LDE matrix
    auto commit_matrix = [&](const Matrix& lde) {
        Commitment c;
        if (G == 1) {
            c.tree = commit_to_rows(lde, false);
            c.n_global = lde.rows;
            c.root = c.tree.root();
            return c;
        }
        const bool rows_path = exchange_rows && (size_t)lde.cols * 8 < sizeof(Digest) && lde.rows >= 8 * (size_t)G && lde.rows % G == 0;
        if (exchange_chunks > 1 && lde.rows % G == 0 && lde.rows / G >= 8) {
            // as many chunks as asked for, as long as a chunk keeps at least 8 leaves per peer (K and the piece are powers of two)
            int K = 1;
            while (K * 2 <= exchange_chunks && (lde.rows / G) / (size_t)(K * 2) >= 8) K *= 2;
            if (K > 1) return commit_exchange_chunked(lde, K, rows_path);
        }
        if (rows_path) return commit_exchange_rows(lde);
        DevBuf<Digest> local(ctx, lde.rows);
        ctx->hash_rows(lde.data.get(), lde.rows, lde.cols, lde.rows, local.get());
        return commit_exchange(local, lde.rows);
    };

    // 1. interpolate_columns [a3]: coefficient i scaled by h^i, so the coset LDE below needs no shift pass
    // 2. evaluate_columns_over [a4] (this rank's coset: M rows)
    // Compact copies for the per-row kernels that walk an LDE with a stride (one GPU only): the trace / aux LDE's last pass also
    // writes every ce_step-th row densely (what constraint evaluation reads; DEEP reads every (B / ce_step)-th row of that), or
    // every B-th row when the constraint domain is the whole LDE domain; the composition LDE every B-th row (DEEP only).
    const int log_ce_step = log_B - ilog2(C);
    const int tc_log = (G == 1 && compact_rows) ? (log_ce_step > 0 ? log_ce_step : log_B) : 0;
    // every (B / ce_step)-th compact row is a DEEP row: de-interleave so that DEEP reads part 0 contiguously
    const int tc_split = log_ce_step > 0 ? log_B - log_ce_step : 0;
    const int cc_log = (G == 1 && compact_rows) ? log_B : 0;
    Matrix polys(ctx, (int)W, n);
    Matrix tlde(ctx, (int)W, M), tlde_c, alde_c, clde_c;
    bool have_tc = false, have_ac = false, have_cc = false;
    CompactOut tco;
    if (tc_log > 0 && log_M >= 14) { tlde_c = Matrix(ctx, (int)W, M >> tc_log); tco.ptr = tlde_c.data.get(); tco.col_stride = M >> tc_log; tco.log_step = tc_log; tco.log_split = tc_split; }
    // columns [c0, c0 + nc): coefficients -> this rank's coset of the LDE (+ the compact copy)
    auto extend_columns = [&](uint32_t c0, uint32_t nc) {
        CompactOut co = tco;
        if (co.ptr) co.ptr += (size_t)c0 * co.col_stride;
        return ctx->ntt_forward(polys.data.get() + (size_t)c0 * n, n, tlde.data.get() + (size_t)c0 * M, M, (int)nc, log_M, log_Bl, &co);
    };
    DevBuf<uint64_t> trace_keep;              // device copy of a HOST trace, kept when the AIR reads the main segment again (aux builders)
    bool aux_columns_pending = false;         // some of its columns are still travelling on the copy stream (event 1)
    const uint64_t* trace_src = trace_dev;
    if (host_trace) {
        // Direct hand-over: the columns are copied straight into the buffer the interpolation works in; the canonical-form check
        // rides on the first inverse pass (the pass that reads the values anyway: no separate read of the trace).
        // Wide traces travel in column groups on a second stream: while group g + 1 is on the PCIe link, group g is interpolated
        // AND extended on the proving stream, so the copy hides behind the two transforms (2^20 x 72: 604 MB, 11 ms of link time
        // against 8 ms of transforms). For narrow traces the groups are not worth their events: with 8 proofs of 2^20 x 2 in
        // flight a second stream per proof cost 9 % of the throughput (round 2), and other proofs' kernels already overlap the copy.
        unsigned int* d_bad = (unsigned int*)ctx->scratch_alloc(8);
        AERO_HIP(hipMemsetAsync(d_bad, 0, 8, ctx->stream));
        if (A) trace_keep = DevBuf<uint64_t>(ctx, (size_t)W * n);
        uint64_t* const land = A ? trace_keep.get() : polys.data.get();
        const size_t col_bytes = n * 8;
        uint32_t gw = W;
        if (h2d_pipeline && G == 1 && (W >= 16 || (W >= 2 && col_bytes >= ((size_t)4 << 20)))) {     // wide traces, or few but long columns (2^24 x 2: one column per group)
            gw = (W + 15) / 16;                                                   // at most 16 groups ...
            const uint32_t min_cols = (uint32_t)(((size_t)(W >= 16 ? 32 : 8) << 20) / col_bytes);   // ... of at least 32 MiB (8 MiB for narrow traces)
            if (gw < min_cols) gw = min_cols;
            if (gw > W) gw = W;
        }
        if (G > 1 && W >= (uint32_t)G && W % (uint32_t)G == 0) {
            // One proof over G GPUs, trace in host memory: this rank copies and interpolates only ITS W / G columns (1 / G of the
            // PCIe traffic and of the interpolation), the coefficients are all-gathered (in place) and every rank applies the
            // pre-scaling h^i of its own coset afterwards. An AIR with an auxiliary segment reads main columns again when it builds
            // it: the evaluations are all-gathered as well (over the GPU links instead of G whole-trace copies over PCIe).
            const uint32_t cpr = W / (uint32_t)G, c0 = (uint32_t)rank * cpr;
            uint64_t* const mine = polys.data.get() + (size_t)c0 * n;
            AERO_HIP(hipMemcpyAsync(A ? land + (size_t)c0 * n : mine, host_trace + (size_t)c0 * n, (size_t)cpr * col_bytes, hipMemcpyHostToDevice, ctx->stream));
            if (A) {
                AERO_HIP(hipMemcpyAsync(mine, land + (size_t)c0 * n, (size_t)cpr * col_bytes, hipMemcpyDeviceToDevice, ctx->stream));
                // the main columns the auxiliary builders read: few of them (9 of 72 for the stand-in) come straight from the host,
                // otherwise the evaluations are all-gathered over the GPU links
                std::vector<uint8_t> need(W, 0);
                if (prog) { for (uint32_t d : prog->aux_desc) if (!(d & 0x40000000u)) need[d & 0xffff] = 1; for (uint32_t c : prog->general_main_cols) need[c] = 1; }
                else for (uint32_t c = 0; c < A; c++) need[c % W] = 1;
                // The choice must be the SAME on every rank (the all-gather is a collective): it is made from the largest number of
                // foreign needed columns any rank would have to fetch, computed here for all ranks alike - not from this rank's own count
                // (W = 8, G = 4, A = 5: ranks 0-2 need 3 or 4 foreign columns, rank 3 needs 5).
                uint32_t total_need = 0, min_own = W;
                for (uint32_t c = 0; c < W; c++) total_need += need[c];
                for (uint32_t r = 0; r < (uint32_t)G; r++) {
                    uint32_t own = 0;
                    for (uint32_t c = r * cpr; c < (r + 1) * cpr; c++) own += need[c];
                    min_own = std::min(min_own, own);
                }
                const uint32_t extra_max = total_need - min_own;
                if (extra_max * 2 <= W) {
                    // on the copy stream: they are not needed before the main segment is committed
                    hipStream_t cs = ctx->get_copy_stream();
                    AERO_HIP(hipEventRecord(ctx->sync_event(0), ctx->stream));
                    AERO_HIP(hipStreamWaitEvent(cs, ctx->sync_event(0), 0));
                    for (uint32_t c = 0; c < W; c++)
                        if (need[c] && (c < c0 || c >= c0 + cpr))
                            AERO_HIP(hipMemcpyAsync(land + (size_t)c * n, host_trace + (size_t)c * n, col_bytes, hipMemcpyHostToDevice, cs));
                    AERO_HIP(hipEventRecord(ctx->sync_event(1), cs));
                    aux_columns_pending = true;
                } else comm_all_gather(land + (size_t)c0 * n, land, (size_t)cpr * col_bytes);
            }
            ctx->ntt_inverse(mine, n, (int)cpr, log_n, 1, 1, 1, 0, d_bad);          // plain coefficients: no coset yet
            comm_all_gather(mine, polys.data.get(), (size_t)cpr * col_bytes);
            {
                const int lo_bits = (log_n + 1) / 2;
                std::vector<uint64_t> lo((size_t)1 << lo_bits), hi((size_t)1 << (log_n - lo_bits));
                uint64_t v = 1;
                for (auto& x : lo) { x = v; v = gl::mul(v, h); }
                const uint64_t step = v;                                              // h^(2^lo_bits)
                v = 1;
                for (auto& x : hi) { x = v; v = gl::mul(v, step); }
                ParamPack pp(ctx);
                const size_t i_lo = pp.add(lo), i_hi = pp.add(hi);
                pp.commit();
                launch_scale_pow_bitrev(ctx, polys.data.get(), n, (int)W, log_n, pp.ptr<uint64_t>(i_lo), pp.ptr<uint64_t>(i_hi), lo_bits);
            }
            ms.interpolate = clk.lap();
            have_tc = extend_columns(0, W);
        } else if (gw == W) {
            if (ctx->copy_gate) {
                // one copy at a time per pool (Context::CopyGate): behind the previous context's copy, in front of this proof's kernels
                std::lock_guard<std::mutex> lk(ctx->copy_gate->mu);
                if (ctx->copy_gate->last) AERO_HIP(hipStreamWaitEvent(ctx->stream, ctx->copy_gate->last, 0));
                AERO_HIP(hipMemcpyAsync(land, host_trace, (size_t)W * col_bytes, hipMemcpyHostToDevice, ctx->stream));
                if (!ctx->gate_event) AERO_HIP(hipEventCreateWithFlags(&ctx->gate_event, hipEventDisableTiming));
                AERO_HIP(hipEventRecord(ctx->gate_event, ctx->stream));
                ctx->copy_gate->last = ctx->gate_event;
            } else
            AERO_HIP(hipMemcpyAsync(land, host_trace, (size_t)W * col_bytes, hipMemcpyHostToDevice, ctx->stream));
            if (A) AERO_HIP(hipMemcpyAsync(polys.data.get(), land, (size_t)W * col_bytes, hipMemcpyDeviceToDevice, ctx->stream));
            ctx->ntt_inverse(polys.data.get(), n, (int)W, log_n, 1, h, 1, 0, d_bad);
            ms.interpolate = clk.lap();
            have_tc = extend_columns(0, W);
        } else {
            hipStream_t cs = ctx->get_copy_stream();
            const uint32_t groups = (W + gw - 1) / gw;
            AERO_HIP(hipEventRecord(ctx->sync_event(0), ctx->stream));           // the buffers are free once the proving stream gets here
            AERO_HIP(hipStreamWaitEvent(cs, ctx->sync_event(0), 0));
            for (uint32_t g = 0; g < groups; g++) {
                const uint32_t c0 = g * gw, nc = std::min(gw, W - c0);
                AERO_HIP(hipMemcpyAsync(land + (size_t)c0 * n, host_trace + (size_t)c0 * n, (size_t)nc * col_bytes, hipMemcpyHostToDevice, cs));
                AERO_HIP(hipEventRecord(ctx->sync_event(1 + g), cs));
            }
            have_tc = true;
            for (uint32_t g = 0; g < groups; g++) {
                const uint32_t c0 = g * gw, nc = std::min(gw, W - c0);
                AERO_HIP(hipStreamWaitEvent(ctx->stream, ctx->sync_event(1 + g), 0));
                if (A) AERO_HIP(hipMemcpyAsync(polys.data.get() + (size_t)c0 * n, land + (size_t)c0 * n, (size_t)nc * col_bytes, hipMemcpyDeviceToDevice, ctx->stream));
                ctx->ntt_inverse(polys.data.get() + (size_t)c0 * n, n, (int)nc, log_n, 1, h, 1, 0, d_bad);
                if (!extend_columns(c0, nc)) have_tc = false;
            }
            ms.interpolate = 0;
        }
        if (G > 1) {
            // every rank checked its own columns: one verdict for all, and known BEFORE the ranks go on exchanging data derived
            // from values the field arithmetic is not defined on (their transcripts could part ways mid-proof)
            comm_all_reduce(reinterpret_cast<uint64_t*>(d_bad), 1);
            uint64_t bad = 0;
            ctx->fetch(&bad, d_bad, 8);
            if (bad) { *host_verdict_ = 1; fail("prove: the trace holds a non-canonical field element (>= p)"); }
        }
        AERO_HIP(hipMemcpyAsync(host_verdict_, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
        trace_src = A ? trace_keep.get() : nullptr;
    } else {
        unsigned int* d_bad = nullptr;
        if (landed) {       // nobody has looked at these values yet: the pass that reads them checks the canonical form
            d_bad = (unsigned int*)ctx->scratch_alloc(8);
            AERO_HIP(hipMemsetAsync(d_bad, 0, 8, ctx->stream));
        }
        ctx->ntt_inverse(polys.data.get(), n, (int)W, log_n, 1, h, 1, 0, d_bad, trace_dev, n);      // read from the caller's matrix: no copy
        if (landed) AERO_HIP(hipMemcpyAsync(host_verdict_, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
        ms.interpolate = clk.lap();
        have_tc = extend_columns(0, W);
    }
    ms.lde = clk.lap();
    // 3. row hashes, Merkle tree, commit [a5, a6, a8]
    gaps.mark("lde_enqueued");
    Commitment tcom = commit_matrix(tlde);          // synchronises the stream: the last trace row has arrived as well
    gaps.mark("root1");
    if (trace_commit_only) {
        Bytes out;
        wdigest(out, tcom.root);
        if (G > 1) for (int r = 0; r < G; r++) wdigest(out, tcom.top[(size_t)G + r]);
        else wdigest(out, tcom.root);
        if (pub_out) pub_out->clear();
        ctx->sync();
        ctx->scratch_reset();
        return out;
    }
    if (prog) air.results = program_pub_;      // a program's public inputs are the caller's (FibAir reads its own off the trace)
    else for (uint32_t k = 0; k < W / 2; k++) air.results[k] = h_last_row[2 * k + 1];
    if (pub_out) *pub_out = air.results;
    HostCoin coin = HostCoin::from_elements(air.results.data(), (uint32_t)air.results.size());
    wdigest(proof.commitments, tcom.root);
    coin.reseed(tcom.root);
    // 3b. auxiliary segment [a8; stark_verifier.cairo:266-294]: draw the random elements, build the columns (prefix products
    //     over the rows), then interpolate / extend / commit them like the main segment
    std::vector<T> rands;
    Matrix apolys, alde;
    Commitment acom;
    const T* d_rands = nullptr;
    if (A) {
        for (uint32_t i = 0; i < R; i++) rands.push_back(coin.draw<F>());
        ParamPack pp(ctx);
        const size_t ir = pp.add(rands);
        pp.commit();
        d_rands = pp.ptr<T>(ir);
        apolys = Matrix(ctx, (int)(A * F::DEG), n);
        if (aux_columns_pending) AERO_HIP(hipStreamWaitEvent(ctx->stream, ctx->sync_event(1), 0));
        if (prog) air_build_aux<F>(ctx, *prog, trace_src, log_n, air.results.data(), rands.data(), apolys.data.get());
        else launch_aux_columns<F>(ctx, trace_src, n, W, A, R, D, d_rands, apolys.data.get());
        trace_keep.release();
        ctx->ntt_inverse(apolys.data.get(), n, (int)(A * F::DEG), log_n, 1, h, 1, 0);
        alde = Matrix(ctx, (int)(A * F::DEG), M);
        {
            CompactOut co;
            if (have_tc) { alde_c = Matrix(ctx, (int)(A * F::DEG), M >> tc_log); co.ptr = alde_c.data.get(); co.col_stride = M >> tc_log; co.log_step = tc_log; co.log_split = tc_split; }
            have_ac = ctx->ntt_forward(apolys.data.get(), n, alde.data.get(), M, (int)(A * F::DEG), log_M, log_Bl, &co);
        }
        acom = commit_matrix(alde);
        wdigest(proof.commitments, acom.root);
        coin.reseed(acom.root);
    }
    ms.trace_commit = clk.lap();

    // 4. constraint composition coefficients + evaluation + division (fused) [a9, a10, a11]
    std::vector<T> ta, tb, ba, bb;
    for (size_t i = 0; i < n_trans; i++) { ta.push_back(coin.draw<F>()); tb.push_back(coin.draw<F>()); }
    for (size_t i = 0; i < n_assert; i++) { ba.push_back(coin.draw<F>()); bb.push_back(coin.draw<F>()); }
    DevBuf<uint64_t> hbuf(ctx, (size_t)F::DEG * ceN);   // H evaluations over h<w_ce>, then coefficients: [DEG][ceN]
    {
        // H (degree < C*n) is interpolated from its values on the coset h<w_ce> of the constraint domain:
        //  * one GPU, or a shard at least as large as the constraint domain: every (M/ce_n)-th row of this rank's LDE;
        //  * a shard smaller than the constraint domain (ce_n = q M): h<w_ce> is the union of the cosets of q ranks - the point
        //    h w_ce^t is the global LDE row rank + t N / ce_n, owned by rank (row mod G). Every rank evaluates the constraints on
        //    the rows of its OWN coset (no extra extension of the trace, 1/q of the evaluations), the H values are all-gathered
        //    (8 M bytes per component to each peer) and the q cosets that make up h<w_ce> are picked out of the gathered block.
        //    (Before: a dedicated extension of every trace column onto h<w_ce> and the evaluation of all ce_n rows on every rank.)
        const bool gather_h = G > 1 && M < ceN;
        const uint64_t* frame_src = tlde.data.get();
        const uint64_t* aux_src = A ? alde.data.get() : nullptr;
        size_t frame_rows = M;
        // rows evaluated by this launch: the ce_n points of h<w_ce>, or (gather_h) the M rows of the coset h<w_M>
        const size_t rows_eval = gather_h ? M : ceN;
        const size_t xcount = gather_h ? M / n : C;             // distinct values of x^n over those rows
        NttTables* tce = ctx->ntt_tables(ilog2(rows_eval));
        int cons_split = 0;
        // one GPU: the rows the constraint domain consists of were also written densely by the LDE (every ce_step-th row)
        if (have_tc && (!A || have_ac) && log_ce_step > 0 && frame_src == tlde.data.get()) {
            frame_src = tlde_c.data.get();
            if (A) aux_src = alde_c.data.get();
            frame_rows = M >> tc_log;
            cons_split = tc_split;
        }
        if (prog) {
            // program AIR: the device interpreter over the same rows (air_kernels.hip)
            AirGeometry ge;
            ge.lde = frame_src; ge.aux = aux_src; ge.frame_rows = frame_rows; ge.split_log = (uint32_t)cons_split;
            ge.rows = rows_eval; ge.first = 0; ge.count = rows_eval; ge.offset = h;
            AirCoeffs<F> cc{ta, tb, ba, bb};
            if (!gather_h) {
                uint64_t* oh[2] = {hbuf.get(), hbuf.get() + (F::DEG > 1 ? ceN : 0)};
                air_eval_constraints<F>(ctx, *prog, pinst, ge, cc, air.results.data(), rands.data(), 1, nullptr, oh);
            } else {
                DevBuf<uint64_t> hloc(ctx, (size_t)F::DEG * M), hall(ctx, (size_t)F::DEG * N);
                uint64_t* oh[2] = {hloc.get(), hloc.get() + (F::DEG > 1 ? M : 0)};
                air_eval_constraints<F>(ctx, *prog, pinst, ge, cc, air.results.data(), rands.data(), 1, nullptr, oh);
                gather_h_cosets(hloc.get(), hall.get(), (size_t)F::DEG * M * 8, (int)(ceN / M), (int)((N / ceN) % (size_t)G));          // [rank][component][t]
                for (int d = 0; d < F::DEG; d++)
                    launch_select_coset_u64(ctx, hall.get() + (size_t)d * M, (size_t)F::DEG * M, hbuf.get() + (size_t)d * ceN, ceN, (uint32_t)rank,
                                            N / ceN, N, (uint32_t)G);
            }
        } else {
            FibConsArgs<F> a{};
            a.split_log = (uint32_t)cons_split;
            a.lde = frame_src; a.N = frame_rows; a.W = W; a.C = (uint32_t)C; a.blowup = (uint32_t)(frame_rows / n); a.ce_step = (uint32_t)(frame_rows / rows_eval);
            a.xmask = (uint32_t)xcount - 1;
            a.first = 0; a.count = rows_eval;
            ParamPack pp(ctx);
            const size_t i_ta = pp.add(ta), i_tb = pp.add(tb), i_ba = pp.add(ba), i_bb = pp.add(bb), i_res = pp.add(air.results);
            a.tw_lo = tce->lo_fwd; a.tw_hi = tce->hi_fwd; a.twi_lo = tce->lo_inv; a.twi_hi = tce->hi_inv; a.tw_h = tce->h;
            a.offset = h; a.gen_inv = h_inv; a.k7 = gl::pow(h, ceN);       // x^ce_n is constant on every coset of <w_ce>
            std::vector<uint64_t> xn(xcount), zn(xcount), xnp(xcount);
            uint64_t hn = gl::pow(h, n), wX = gl::root_of_unity(ilog2(xcount));
            for (size_t k = 0; k < xcount; k++) {
                uint64_t xnk = gl::mul(hn, gl::pow(wX, k));
                // aux degree adjustment x^((E + 1 - D) n + (D - 2)): the x^n part is constant on each coset of <w_n>
                xnp[k] = gl::pow(xnk, C + 1 - D);
                xn[k] = gl::inv(xnk);
                zn[k] = gl::inv(gl::sub(xnk, 1));
            }
            const size_t i_xn = pp.add(xn), i_zn = pp.add(zn), i_xnp = pp.add(xnp);
            pp.commit();
            a.aux = aux_src; a.A = A; a.R = R; a.D = D; a.rands = d_rands; a.xn = pp.ptr<uint64_t>(i_xnp);
            a.ta = pp.ptr<T>(i_ta); a.tb = pp.ptr<T>(i_tb); a.ba = pp.ptr<T>(i_ba); a.bb = pp.ptr<T>(i_bb);
            a.results = pp.ptr<uint64_t>(i_res); a.xn_inv = pp.ptr<uint64_t>(i_xn); a.zn_inv = pp.ptr<uint64_t>(i_zn);
            a.w_last = gl::pow(g, n - 1);
            a.out_cols = nullptr;
            if (!gather_h) {
                for (int d = 0; d < F::DEG; d++) a.out_h[d] = hbuf.get() + (size_t)d * ceN;
                // up to 2^25 constraint-domain rows (2^24-row traces: 1.3 GB, one table per device and shape for all contexts; round 5 stopped at 2^23
                // because every context held its own copy - config 4's constraint kernel then inverted per row, 1.17 ms of a 30 ms proof)
                if (ctx->cons_inv_table && rows_eval <= ((size_t)1 << 25) && rows_eval % 4 == 0) {
                    // the divisor inverses and the degree-adjustment powers of a constraint-domain point do not depend on the proof: one table per
                    // shape (5 words per row), built by the first proof of the shape
                    const std::vector<uint64_t> key{(uint64_t)rows_eval, h, a.w_last, (uint64_t)xcount, (uint64_t)n};
                    const uint64_t* tab = ctx->cons_inv_table_for(key, 5 * rows_eval * 8, [&](uint64_t* out) { launch_fib_inverse_table<F>(ctx, out, a); });
                    if (tab) { a.inv_tab = tab; a.inv_tab_n = rows_eval; }
                }
                launch_fib_constraints<F>(ctx, a, 1);
            } else {
                DevBuf<uint64_t> hloc(ctx, (size_t)F::DEG * M), hall(ctx, (size_t)F::DEG * N);
                for (int d = 0; d < F::DEG; d++) a.out_h[d] = hloc.get() + (size_t)d * M;
                launch_fib_constraints<F>(ctx, a, 1);
                gather_h_cosets(hloc.get(), hall.get(), (size_t)F::DEG * M * 8, (int)(ceN / M), (int)((N / ceN) % (size_t)G));          // [rank][component][t]
                for (int d = 0; d < F::DEG; d++)
                    launch_select_coset_u64(ctx, hall.get() + (size_t)d * M, (size_t)F::DEG * M, hbuf.get() + (size_t)d * ceN, ceN, (uint32_t)rank,
                                            N / ceN, N, (uint32_t)G);
            }
        }
    }
    gaps.mark("cons_enqueued");
    ms.constraints = clk.lap();
    // 5. composition polynomial: interpolate over the coset; coefficient I gets h^-I (coset) * h^(I >> log C)
    //    (pre-scaling of column coefficient i = I >> log C for the column LDE). In bit-reversed order the C column
    //    polynomials are the C contiguous chunks of the buffer (chunk q = column bitrev(q)): no split pass
    //    (H(x) = sum_c x^c H_c(x^C)).
    composition_from_evaluations(hbuf.get(), F::DEG, log_ce, ilog2(C), h);
    ms.composition = clk.lap();
    // 6. composition commitment [a12]: column c*DEG + d <- chunk c of component d
    Matrix clde(ctx, (int)(C * F::DEG), M);
    if (cc_log > 0 && log_M >= 14) { clde_c = Matrix(ctx, (int)(C * F::DEG), M >> cc_log); have_cc = true; }
    // In bit-reversed coefficient order the low log2(C) bits of the coefficient index (= the column) are the HIGH bits of the
    // position: column c is chunk bitrev(c). For C = 2 that is the identity and all columns of a component go in one launch.
    const int log_C = ilog2(C);
    for (int d = 0; d < F::DEG; d++) {
        if (C == 2) {
            CompactOut co;
            if (have_cc) { co.ptr = clde_c.data.get() + (size_t)d * (M >> cc_log); co.col_stride = (size_t)F::DEG * (M >> cc_log); co.log_step = cc_log; }
            if (!ctx->ntt_forward(hbuf.get() + (size_t)d * ceN, n, clde.data.get() + (size_t)d * M, (size_t)F::DEG * M, (int)C, log_M, log_Bl, &co)) have_cc = false;
        } else {
            for (size_t c = 0; c < C; c++)
            {
                CompactOut co;
                if (have_cc) { co.ptr = clde_c.data.get() + ((size_t)c * F::DEG + d) * (M >> cc_log); co.col_stride = M >> cc_log; co.log_step = cc_log; }
                if (!ctx->ntt_forward(hbuf.get() + (size_t)d * ceN + (size_t)gl::bitrev((uint32_t)c, log_C) * n, n,
                                      clde.data.get() + ((size_t)c * F::DEG + d) * M, M, 1, log_M, log_Bl, &co)) have_cc = false;
            }
        }
    }
    gaps.mark("clde_enqueued");
    Commitment ccom = commit_matrix(clde);
    gaps.mark("root2");
    wdigest(proof.commitments, ccom.root);
    coin.reseed(ccom.root);
    ms.comp_commit = clk.lap();

    // 7. OOD frame [a13]. Coefficients are pre-scaled by h^i, so evaluate at point / h.
    const T z = coin.draw<F>();
    const T z_next = F::mulb(z, g), z_c = gl::fpow<F>(z, C);
    std::vector<T> ood(2 * W + C + 2 * A);
    {
        // one GPU: the reductions store the frame in mapped pinned memory (the host waits for the stream, no copy); sharded: a device buffer,
        // the all-gather writes into it
        T* h_ood = (T*)ctx->stage_alloc(ood.size() * sizeof(T));
        DevBuf<T> d_ood_buf;
        if (G > 1) d_ood_buf = DevBuf<T>(ctx, 2 * W + C + 2 * A);
        struct { T* p; T* get() const { return p; } } d_out{G > 1 ? d_ood_buf.get() : ctx->stage_dev(h_ood)};
        const bool split_ood = G > 1 && W >= (uint32_t)G && W % (uint32_t)G == 0;
        if (A && split_ood) launch_eval_bitrev<F>(ctx, apolys.data.get(), (size_t)F::DEG * n, n, (int)A, F::DEG, log_n, F::mulb(z, h_inv), F::mulb(z_next, h_inv), 2, d_out.get() + 2 * W + C);
        if (split_ood) {
            // the value of a column polynomial at z does not depend on the coset a rank holds its coefficients for: every rank
            // evaluates W / G columns, one small all-gather (2 W / G elements per rank) completes the frame everywhere
            const uint32_t cpr = W / (uint32_t)G;
            DevBuf<T> part(ctx, 2 * cpr);
            launch_eval_bitrev<F>(ctx, polys.data.get() + (size_t)rank * cpr * n, n, 0, (int)cpr, 1, log_n, F::mulb(z, h_inv), F::mulb(z_next, h_inv), 2, part.get());
            comm_all_gather(part.get(), d_out.get(), 2 * cpr * sizeof(T));
            launch_eval_bitrev<F>(ctx, hbuf.get(), n, ceN, (int)C, F::DEG, log_n, F::mulb(z_c, h_inv), F::zero(), 1, d_out.get() + 2 * W);
        } else {
            // one GPU: the whole frame in three launches (tables, block sums, reduction) - trace polynomials at (z, z g), composition columns at
            // z^C and, below, nothing more; a sharded proof keeps the separate evaluations (its trace part is all-gathered)
            EvalJob<F> jobs[EVAL_MAX_JOBS];
            int nj = 0;
            jobs[nj].coeffs = polys.data.get(); jobs[nj].col_stride = n; jobs[nj].comp_stride = 0; jobs[nj].ncols = (int)W; jobs[nj].comps = 1; jobs[nj].npts = 2;
            jobs[nj].y0 = F::mulb(z, h_inv); jobs[nj].y1 = F::mulb(z_next, h_inv); jobs[nj].out_off = 0; nj++;
            jobs[nj].coeffs = hbuf.get(); jobs[nj].col_stride = n; jobs[nj].comp_stride = ceN; jobs[nj].ncols = (int)C; jobs[nj].comps = F::DEG; jobs[nj].npts = 1;
            jobs[nj].y0 = F::mulb(z_c, h_inv); jobs[nj].y1 = F::zero(); jobs[nj].out_off = 2 * W; nj++;
            if (A) {
                jobs[nj].coeffs = apolys.data.get(); jobs[nj].col_stride = (size_t)F::DEG * n; jobs[nj].comp_stride = n; jobs[nj].ncols = (int)A; jobs[nj].comps = F::DEG; jobs[nj].npts = 2;
                jobs[nj].y0 = F::mulb(z, h_inv); jobs[nj].y1 = F::mulb(z_next, h_inv); jobs[nj].out_off = 2 * W + (uint32_t)C; nj++;
            }
            launch_eval_multi<F>(ctx, jobs, nj, log_n, d_out.get());
        }
        if (G > 1) ctx->fetch(ood.data(), d_out.get(), ood.size() * sizeof(T));
        else { gaps.mark("ood_enqueued"); ctx->sync(); gaps.mark("ood_back"); memcpy(ood.data(), h_ood, ood.size() * sizeof(T)); }
    }
    std::vector<T> ood_cur(TW), ood_next(TW), ood_h(C);
    for (uint32_t c = 0; c < W; c++) { ood_cur[c] = ood[2 * c]; ood_next[c] = ood[2 * c + 1]; }
    for (uint32_t c = 0; c < A; c++) { ood_cur[W + c] = ood[2 * W + C + 2 * c]; ood_next[W + c] = ood[2 * W + C + 2 * c + 1]; }
    for (size_t c = 0; c < C; c++) ood_h[c] = ood[2 * W + gl::bitrev((uint32_t)c, log_C)];   // evaluated in chunk order
    {
        std::vector<uint64_t> f;
        flatten<F>(ood_cur.data(), TW, f); flatten<F>(ood_next.data(), TW, f);
        for (uint64_t v : f) w64(proof.ood_trace_states, v);
        f.clear();
        flatten<F>(ood_h.data(), C, f);
        for (uint64_t v : f) w64(proof.ood_evaluations, v);
    }
    coin.reseed(hash_e<F>(ood_cur.data(), TW));
    coin.reseed(hash_e<F>(ood_next.data(), TW));
    coin.reseed(hash_e<F>(ood_h.data(), C));
    ms.ood = clk.lap();

    // 8. DEEP composition [a14]
    std::vector<T> da(TW), db(TW), dg(TW), dc(C);
    for (uint32_t i = 0; i < TW; i++) { da[i] = coin.draw<F>(); db[i] = coin.draw<F>(); dg[i] = coin.draw<F>(); }
    for (size_t i = 0; i < C; i++) dc[i] = coin.draw<F>();
    const T lambda = coin.draw<F>(), mu = coin.draw<F>();
    // FRI evaluations per layer: [DEG][dom] component arrays, natural order (dom = this rank's share while the layer is sharded)
    std::vector<DevBuf<uint64_t>> fri_vals;
    {
        DeepInputs<F> in;
        in.z = z; in.ood_cur = ood_cur; in.ood_next = ood_next; in.ood_h = ood_h;
        in.da = da; in.db = db; in.dg = dg; in.dc = dc; in.lambda = lambda; in.mu = mu;
        DeepCompact dc_src;
        // with the de-interleaved layout the DEEP rows (every B-th LDE row) are part 0 of the compact copy: contiguous, i.e. the
        // copy looks like an every-B-th-row copy to the DEEP kernel
        if (have_tc && (!A || have_ac)) { dc_src.t = tlde_c.data.get(); dc_src.t_log = tc_log + tc_split; dc_src.t_stride = M >> tc_log;
                                          if (A) { dc_src.a = alde_c.data.get(); dc_src.a_log = tc_log + tc_split; dc_src.a_stride = M >> tc_log; } }
        if (have_cc) { dc_src.c = clde_c.data.get(); dc_src.c_log = cc_log; }
        bool coeff_form = false;
        if constexpr (F::DEG == 1) {
            // base field, one GPU, narrow traces: the quotients as synthetic divisions of the column polynomials (stark.hip: launch_deep_coeff) - no
            // inversions, no interpolation of the result (2^20 x 2: deep kernel + inverse transform 93 us -> 85 us in five short launches, one proof
            // alone 2.155 -> 2.13 ms, eight in flight +0.8 %). From 8 columns on the evaluation form's single fused pass over the rows wins
            // (2^20 x 72: 0.46 ms against 0.53 - 0.60), profiles/r5_deep_coeff.md
            if (ctx->deep_coeff && G == 1 && log_Bl > 0 && log_n >= 3 && W + A < 8) {
                coeff_form = true;
                std::vector<uint64_t> dc_chunk(C);
                for (size_t q = 0; q < C; q++) dc_chunk[q] = dc[gl::bitrev((uint32_t)q, log_C)];       // chunk q of hbuf is column bitrev(q)
                ParamPack pp(ctx);
                const size_t i_a = pp.add(da), i_b = pp.add(db), i_c = pp.add(dc_chunk);
                pp.commit();
                DeepCoeffArgs ca{};
                ca.tpolys = polys.data.get(); ca.t_stride = n; ca.W = W;
                ca.apolys = A ? apolys.data.get() : nullptr; ca.a_stride = n; ca.A = A;
                ca.hpolys = hbuf.get(); ca.h_stride = n; ca.C = (uint32_t)C;
                ca.da = pp.ptr<uint64_t>(i_a); ca.db = pp.ptr<uint64_t>(i_b); ca.dc = pp.ptr<uint64_t>(i_c);
                ca.log_n = log_n;
                ca.y[0] = gl::mul(z, h_inv); ca.y[1] = gl::mul(z_next, h_inv); ca.y[2] = gl::mul(z_c, h_inv);
                ca.lam = gl::mul(lambda, h_inv); ca.mu = mu;
                DevBuf<uint64_t> blocks(ctx, deep_coeff_scratch_words(log_n)), dsm(ctx, n), out(ctx, M);
                ca.blocks = blocks.get(); ca.out = dsm.get();
                launch_deep_coeff(ctx, ca);
                ctx->ntt_forward(dsm.get(), n, out.get(), M, 1, log_M, log_Bl);
                fri_vals.push_back(std::move(out));
            }
        }
        if (!coeff_form)
        fri_vals.push_back(deep_compose<F>(tlde.data.get(), clde.data.get(), A ? alde.data.get() : nullptr, W, A, (uint32_t)C, log_n, log_Bl, h, in, &dc_src));
    }
    gaps.mark("deep_enqueued");
    ms.deep = clk.lap();

    // 9. FRI commit phase [a15]: layers + 1 rounds (the last commits the remainder). A sharded layer of global domain Dom
    //    is this rank's coset 7 w_Dom^rank <w_(Dom/G)> (the fold groups {i + j*Dom/F} stay inside one coset); once the
    //    per-peer digest exchange would drop below min_peer digests the layer is all-gathered and the rest runs unsharded.
    std::vector<Commitment> fri_coms;
    std::vector<char> fri_sharded;
    if (G == 1) {
        FriLayers fl = fri_build_layers<F>(std::move(fri_vals[0]), N, coin, &proof.commitments);
        fri_vals = std::move(fl.vals);
        fri_coms = std::move(fl.coms);
        fri_sharded.assign(layers + 1, 0);
    } else {
        uint64_t Dom = N;
        bool sharded = true;
        for (int l = 0; l <= layers; l++) {
            if (sharded && (l == layers || Dom / Fd / G / G < min_peer)) {
                const size_t Lc = Dom / G;
                DevBuf<uint64_t> all(ctx, (size_t)F::DEG * Dom), full(ctx, (size_t)F::DEG * Dom);
                comm_all_gather(fri_vals[l].get(), all.get(), (size_t)F::DEG * Lc * 8);     // [rank][component][t]
                for (int d = 0; d < F::DEG; d++)
                    launch_interleave_u64(ctx, all.get() + (size_t)d * Lc, (size_t)F::DEG * Lc, full.get() + (size_t)d * Dom, G, Lc);
                fri_vals[l] = std::move(full);
                sharded = false;
            }
            const int parts = sharded ? G : 1;
            const size_t dom = Dom / parts, rows = dom / Fd;          // local domain / local leaf count
            const FriSrc fsrc{fri_vals[l].get(), fri_vals[l].get() + (F::DEG > 1 ? dom : 0), F::DEG, rows, (int)Fd};
            fri_sharded.push_back(sharded);
            if (sharded) {
                DevBuf<Digest> local(ctx, rows);
                ctx->hash_fri_rows(fsrc, local.get());
                fri_coms.push_back(commit_exchange(local, rows));
            } else {
                Commitment c;
                c.tree = commit_fri_layer(fsrc);
                c.n_global = rows;
                c.root = c.tree.root();
                fri_coms.push_back(std::move(c));
            }
            const Digest root = fri_coms.back().root;
            wdigest(proof.commitments, root);
            coin.reseed(root);
            const T alpha = coin.draw<F>();
            if (l == layers) break;   // alpha drawn after the remainder commitment is unused
            fri_vals.emplace_back(ctx, (size_t)F::DEG * rows);
            NttTables* td = ctx->ntt_tables(ilog2(dom));
            FoldArgs<F> a{};
            for (int d = 0; d < F::DEG; d++) { a.in[d] = fri_vals[l].get() + (size_t)d * dom; a.out[d] = fri_vals[l + 1].get() + (size_t)d * rows; }
            if (F::DEG == 1) { a.in[1] = a.in[0]; a.out[1] = a.out[0]; }
            a.rows = rows; a.fold = (int)Fd; a.alpha = alpha;
            a.twi_lo = td->lo_inv; a.twi_hi = td->hi_inv; a.tw_h = td->h;
            // row t of a sharded layer sits at x = 7 w_Dom^(t*G + rank) = (7 w_Dom^rank) * w_dom^t
            a.gen_inv = sharded ? gl::inv(gl::mul(gl::GEN, gl::pow(gl::root_of_unity(ilog2(Dom)), (uint64_t)rank))) : gen_inv;
            a.fold_inv = gl::inv(Fd);
            uint64_t wFi = gl::inv(gl::root_of_unity(ilog2(Fd)));
            for (size_t m = 0; m < Fd; m++) a.dft[m] = gl::pow(wFi, m);
            launch_fri_fold<F>(ctx, a);
            Dom /= Fd;
        }
    }
    gaps.mark("fri_back");
    ms.fri = clk.lap();

    // 10. grinding [a16]
    {
        const uint64_t nonce = run_grind(ctx, coin.seed, opt_.grinding_factor);
        proof.pow_nonce = nonce;
        coin.reseed_with_int(nonce);
    }
    gaps.mark("grind_back");
    ms.grind = clk.lap();

    // 11. queries [a17]: every position / node index is known up front, so all gathers are issued behind ONE upload and
    //     read back with ONE download (one stream synchronisation for the whole opening phase). Sharded: every rank lays
    //     out the same value block, fills the items it owns (zeros elsewhere) and ONE all-reduce completes it everywhere.
    static const bool q_timing = getenv("AERO_QUERY_TIMING") != nullptr;
    auto q_t0 = std::chrono::steady_clock::now();
    auto q_lap = [&](const char* what) {
        if (!q_timing) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "  queries/%s: %.1f us\n", what, std::chrono::duration<double, std::micro>(t1 - q_t0).count());
        q_t0 = t1;
    };
    std::vector<uint64_t> pos = coin.draw_integers(opt_.num_queries, N);
    q_lap("draw_positions");
    {
        const size_t Q = pos.size(), tw = W, aw = (size_t)A * F::DEG, cw = C * F::DEG;
        std::vector<std::vector<uint64_t>> fpos(layers);
        {
            std::vector<uint64_t> fp = pos;
            uint64_t dom = N;
            for (int l = 0; l < layers; l++) { fp = fold_positions(fp, dom, Fd); fpos[l] = fp; dom /= Fd; }
        }
        const uint64_t rem_dom = N / [&] { uint64_t d = 1; for (int l = 0; l < layers; l++) d *= Fd; return d; }();
        // node index plans per tree (global heap indices): trace, composition, FRI layers
        std::vector<const Commitment*> coms{&tcom, &ccom};
        std::vector<BatchPlan> plan_store;
        plan_store.reserve(layers + 1);
        plan_store.push_back(batch_proof_plan(N, pos));
        std::vector<const BatchPlan*> plans{&plan_store[0], &plan_store[0]};   // same positions, same tree shape: trace and composition share the plan
        const size_t fri_tree0 = A ? 3 : 2;          // tree order: trace, composition, [aux], FRI layers
        if (A) { coms.push_back(&acom); plans.push_back(&plan_store[0]); }
        for (int l = 0; l < layers; l++) { coms.push_back(&fri_coms[l]); plan_store.push_back(batch_proof_plan(fri_coms[l].n_global, fpos[l])); plans.push_back(&plan_store.back()); }
        // position of a row / node in this rank's arrays, GATHER_SKIP when another rank owns it
        auto local_row = [&](uint64_t p, bool sharded) -> uint64_t {
            if (!sharded) return (G == 1 || rank == 0) ? p : GATHER_SKIP;
            return (p & (uint64_t)(G - 1)) == (uint64_t)rank ? p >> log_G : GATHER_SKIP;
        };
        auto local_node = [&](uint64_t gidx, const Commitment& c) -> uint64_t {
            if (!c.sharded) return (G == 1 || rank == 0) ? gidx : GATHER_SKIP;
            if (gidx < 2 * (uint64_t)G) return GATHER_SKIP;              // top levels: filled in from the host copy
            int d = 63 - __builtin_clzll(gidx);
            const uint64_t o = gidx - (1ull << d), sub_bits = d - log_G;
            if ((o >> sub_bits) != (uint64_t)rank) return GATHER_SKIP;
            const uint64_t u = o & ((1ull << sub_bits) - 1), L = 1ull << sub_bits;
            if (L == c.tree.n && c.leaf_parts_log)             // leaf level stored as pieces in arrival order
                return L + (u & ((1ull << c.leaf_parts_log) - 1)) * (L >> c.leaf_parts_log) + (u >> c.leaf_parts_log);
            return L + u;
        };
        q_lap("plans");
        // Openings: every source address is known on the host, so ONE address list goes up, ONE kernel gathers every value and
        // digest, and the (few) unstored low tree nodes are recomputed behind it. Value block (u64):
        //   [trace rows | comp rows | aux rows | fri rows per layer | remainder | digests (4 u64 each) per tree]
        size_t voff = 0;
        const size_t off_trows = voff; voff += Q * tw;
        const size_t off_crows = voff; voff += Q * cw;
        const size_t off_arows = voff; voff += Q * aw;
        std::vector<size_t> off_frows(layers);
        for (int l = 0; l < layers; l++) { off_frows[l] = voff; voff += fpos[l].size() * Fd * F::DEG; }
        const size_t off_rem = voff; voff += (size_t)F::DEG * rem_dom;
        const size_t n_u64 = voff;
        auto A64 = [](const void* p) { return (uint64_t)(uintptr_t)p; };
        std::vector<uint64_t> addr(n_u64, 0);
        for (size_t q = 0; q < Q; q++) {
            const uint64_t r = local_row(pos[q], G > 1);
            if (r == GATHER_SKIP) continue;
            for (size_t c = 0; c < tw; c++) addr[off_trows + q * tw + c] = A64(tlde.data.get() + c * M + r);
            for (size_t c = 0; c < cw; c++) addr[off_crows + q * cw + c] = A64(clde.data.get() + c * M + r);
            for (size_t c = 0; c < aw; c++) addr[off_arows + q * aw + c] = A64(alde.data.get() + c * M + r);
        }
        {
            uint64_t Dom = N;
            for (int l = 0; l < layers; l++) {
                const size_t dom = Dom / (fri_sharded[l] ? G : 1), rows = dom / Fd;
                const uint64_t* comp[2] = {fri_vals[l].get(), fri_vals[l].get() + (F::DEG > 1 ? dom : 0)};
                for (size_t q = 0; q < fpos[l].size(); q++) {
                    const uint64_t r = local_row(fpos[l][q], fri_sharded[l]);
                    if (r == GATHER_SKIP) continue;
                    for (size_t j = 0; j < Fd; j++) for (int d = 0; d < F::DEG; d++)
                        addr[off_frows[l] + (q * Fd + j) * F::DEG + d] = A64(comp[d] + r + j * rows);
                }
                Dom /= Fd;
            }
        }
        if (G == 1 || rank == 0) for (size_t i = 0; i < (size_t)F::DEG * rem_dom; i++) addr[off_rem + i] = A64(fri_vals[layers].get() + i);
        // digests: per tree the plan's items in plan order; unstored low nodes (address 0 here) are recomputed into their slots
        std::vector<size_t> off_dig(coms.size());
        std::vector<std::vector<uint64_t>> low_idx(coms.size());       // heap indices of the unstored nodes a plan needs
        std::vector<std::vector<uint32_t>> low_slot(coms.size());      // their item numbers within the tree's digest block
        size_t n_dig = 0;
        for (size_t t = 0; t < coms.size(); t++) {
            off_dig[t] = n_u64 + 4 * n_dig;
            const MerkleTree& tree = coms[t]->tree;
            if (G > 1 && tree.skip) fail("sharded prove: tree with unstored levels", ST_INTERNAL);
            uint32_t k = 0;
            plans[t]->for_each([&](uint64_t i) {
                uint64_t a = 0;
                if (G > 1) { const uint64_t li = local_node(i, *coms[t]); if (li != GATHER_SKIP) a = A64(tree.nodes.get() + li); }
                else if (i < tree.stored_limit()) a = A64(tree.nodes.get() + i);
                else { low_idx[t].push_back(i); low_slot[t].push_back(k); }
                addr.push_back(a);
                k++;
            });
            if (!low_idx[t].empty() && tree.src_kind == 0) fail("batch opening: tree has unstored levels but no leaf source", ST_INTERNAL);
            n_dig += k;
        }
        voff = n_u64 + 4 * n_dig;
        size_t n_low = 0;
        for (auto& v : low_idx) n_low += v.size();
        q_lap("addresses");
        uint64_t* h_addr = (uint64_t*)ctx->stage_alloc((addr.size() + n_low) * 8);
        memcpy(h_addr, addr.data(), addr.size() * 8);
        {
            size_t o = addr.size();
            for (auto& v : low_idx) { memcpy(h_addr + o, v.data(), v.size() * 8); o += v.size(); }
        }
        uint64_t* h_val = (uint64_t*)ctx->stage_alloc((voff + 4 * n_low) * 8);
        // One GPU, row-matrix trees only (every proof of this library but a sharded one): ONE launch gathers the values and recomputes the
        // unstored low nodes, reading the lists from and writing the block to mapped pinned memory (hash.hip: openings_kernel)
        bool fused_open = G == 1;
        {
            int jobs = 0;
            for (size_t t = 0; t < coms.size(); t++) if (!low_idx[t].empty()) { jobs++; fused_open = fused_open && coms[t]->tree.src_kind == 1; }
            fused_open = fused_open && jobs <= OPEN_MAX_JOBS;
        }
        if (fused_open) {
            OpeningArgs oa;
            oa.addr = ctx->stage_dev(h_addr); oa.n_u64 = (uint32_t)n_u64; oa.n_dig = (uint32_t)n_dig; oa.gather_blocks = (uint32_t)((n_u64 + n_dig + 255) / 256);
            oa.out = ctx->stage_dev(h_val);
            size_t o = 0;
            for (size_t t = 0; t < coms.size(); t++) {
                const int nl = (int)low_idx[t].size();
                if (!nl) continue;
                OpeningArgs::Job& J = oa.jobs[oa.n_jobs++];
                J.src = coms[t]->tree.row_src; J.n = coms[t]->tree.n; J.idx = ctx->stage_dev(h_addr) + addr.size() + o; J.count = nl;
                J.blocks = (uint32_t)((nl * 8 + 255) / 256); J.out = reinterpret_cast<Digest*>(ctx->stage_dev(h_val) + voff) + o;
                o += nl;
            }
            ctx->openings(oa);
            q_lap("enqueue");
            ctx->sync();
            q_lap("device_round_trip");
        } else {
        DevBuf<uint64_t> d_addr(ctx, addr.size() + n_low), d_val(ctx, voff + 4 * n_low);
        AERO_HIP(hipMemcpyAsync(d_addr.get(), h_addr, (addr.size() + n_low) * 8, hipMemcpyHostToDevice, ctx->stream));
        launch_gather_addr(ctx, d_addr.get(), (uint32_t)n_u64, (uint32_t)n_dig, d_val.get());
        {
            // recomputed low nodes land behind the value block (one compact run per tree) and are patched in on the host
            size_t o = 0;
            for (size_t t = 0; t < coms.size(); t++) {
                const int nl = (int)low_idx[t].size();
                if (!nl) continue;
                const MerkleTree& tree = coms[t]->tree;
                Digest* dout = reinterpret_cast<Digest*>(d_val.get() + voff) + o;
                if (tree.src_kind == 1) ctx->merkle_recompute(tree.row_src, tree.n, d_addr.get() + addr.size() + o, nl, dout);
                else ctx->merkle_recompute(tree.fri_src, tree.n, d_addr.get() + addr.size() + o, nl, dout);
                o += nl;
            }
        }
        q_lap("enqueue");
        if (G > 1) comm_all_reduce(d_val.get(), voff);
        {   // straight into h_val (fetch() would stage the block a second time)
            AERO_HIP(hipMemcpyAsync(h_val, d_val.get(), (voff + 4 * n_low) * 8, hipMemcpyDeviceToHost, ctx->stream));
            ctx->sync();
        }
        q_lap("device_round_trip");
        }
        {
            size_t o = 0;
            for (size_t t = 0; t < coms.size(); t++)
                for (size_t k = 0; k < low_slot[t].size(); k++, o++) memcpy(h_val + off_dig[t] + 4 * (size_t)low_slot[t][k], h_val + voff + 4 * o, 32);
        }
        auto paths = [&](size_t t) {
            const Digest* raw = reinterpret_cast<const Digest*>(h_val + off_dig[t]);
            if (!coms[t]->sharded) return serialize_plan(*plans[t], raw);       // items are already in plan order
            std::vector<Digest> ordered(plans[t]->total());
            size_t k = 0;
            plans[t]->for_each([&](uint64_t i) {
                ordered[k] = i < 2 * (uint64_t)G ? coms[t]->top[i] : raw[k];
                k++;
            });
            return serialize_plan(*plans[t], ordered.data());
        };
        auto put = [&](Bytes& b, size_t off, size_t count) {      // `count` little-endian u64 from the value block
            const uint8_t* p8 = reinterpret_cast<const uint8_t*>(h_val + off);
            b.insert(b.end(), p8, p8 + 8 * count);
        };
        QueriesBytes tq;
        put(tq.values, off_trows, Q * tw);
        tq.paths = paths(0);
        proof.trace_queries.push_back(tq);
        if (A) {
            QueriesBytes aq;
            put(aq.values, off_arows, Q * aw);
            aq.paths = paths(2);
            proof.trace_queries.push_back(aq);
        }
        put(proof.constraint_queries.values, off_crows, Q * cw);
        proof.constraint_queries.paths = paths(1);
        for (int l = 0; l < layers; l++) {
            QueriesBytes q;
            put(q.values, off_frows[l], fpos[l].size() * Fd * F::DEG);
            q.paths = paths(fri_tree0 + l);
            proof.fri_layers.push_back(q);
        }
        // remainder = last layer's evaluations in natural order
        for (size_t i = 0; i < rem_dom; i++) for (int d = 0; d < F::DEG; d++) w64(proof.fri_remainder, h_val[off_rem + (size_t)d * rem_dom + i]);
    }
    q_lap("serialise");
    gaps.mark("done");
    ms.queries = clk.lap();
    ctx->scratch_reset();
    ms.total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    last_stage_ms = ms;
    return proof.to_bytes();
}

Bytes Prover::prove(const uint64_t* trace_dev, uint32_t width, int log_n, std::vector<uint64_t>* pub_out) {
    try {
        if (opt_.field_extension == EXT_NONE) return prove_impl<FB>(trace_dev, width, log_n, pub_out);
        return prove_impl<FQ>(trace_dev, width, log_n, pub_out);
    } catch (...) {
        if (ctx_->copy_stream) (void)hipStreamSynchronize(ctx_->copy_stream);     // a column-group copy may still read the caller's buffer
        (void)hipStreamSynchronize(ctx_->stream);     // nothing enqueued by the failed proof may outlive its scratch blocks
        (void)hipGetLastError();
        ctx_->scratch_reset();
        throw;
    }
}

}  // namespace aero
