// In-process communicator for a sharded proof (include/aero_stark.h: aero_local_group_*): `world` ranks = `world` contexts of THIS
// process, one host thread each, on one GPU or on several (peer access) - no RCCL, no second process, no Python.
//
// Purpose: (i) the sharded prover with its STREAM-ORDERED exchange contract (AERO_COMM_STREAM_ORDERED: the prover neither
// synchronises before an exchange nor waits after it) runs with world > 1 inside `pytest -m gpu` on a one-GPU box; (ii) a host
// that drives several GPUs from one process (the shape SURVEY 8b sketched: a context per device of a device list) can shard a
// proof without RCCL. The reference has no multi-device prover; what these exchanges replace is the fan-in of its worker pool
// (aero-sdk/miden-wasm/src/proving_worker.rs:302-310,428-437).
//
// Every exchange is enqueued on the calling rank's own stream:
//   1. the rank publishes its send buffer and records a `ready` event behind everything it has enqueued so far;
//   2. host rendezvous (all ranks have published);
//   3. the rank makes its stream wait for each peer's `ready` event and enqueues device-to-device copies out of the peers'
//      send buffers (all_to_all: peer p's chunk for this rank; all_gather: peer p's piece; all_reduce: a sum kernel over the
//      peers' staged copies), then records `done`;
//   4. host rendezvous, then the rank's stream waits for every peer's `done` (nobody overwrites a buffer a peer still reads).
// The host threads only meet at the rendezvous; no stream is ever synchronised.
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "capi_internal.hpp"

using namespace aero;

struct aero_local_group {
    uint32_t world = 1;
    std::mutex mu;
    std::condition_variable cv;
    uint32_t arrived = 0;
    uint64_t generation = 0;
    bool aborted = false;
    struct Slot {
        aero_local_group* g = nullptr;
        int rank = 0, device = 0;
        hipStream_t stream = nullptr;
        const void* send = nullptr;
        uint64_t bytes = 0;
        hipEvent_t ready = nullptr, done = nullptr;
        uint64_t* stage = nullptr;          // all_reduce: this rank's contribution, staged so that the sum can be formed in place
        size_t stage_cap = 0;
        const uint64_t** srcs = nullptr;    // device array of the peers' stage pointers
        uint64_t calls[3] = {0, 0, 0};
        uint64_t bytes_sent = 0;
        bool bound = false;
        std::string err;
    };
    std::vector<Slot> slots;

    // false when the group was aborted (a rank failed outside an exchange: its peers must not wait for it forever)
    bool rendezvous() {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const uint64_t gen = generation;
        if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return generation != gen || aborted; });
        return generation != gen;
    }
};

namespace {

__global__ __launch_bounds__(256) void local_sum_u64_kernel(uint64_t* __restrict__ dst, const uint64_t* const* __restrict__ srcs, uint32_t world, size_t count) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < count; i += stride) {
        uint64_t s = 0;
        for (uint32_t p = 0; p < world; p++) s += srcs[p][i];
        dst[i] = s;
    }
}

#define LOCAL_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { s->err = std::string(#expr) + ": " + hipGetErrorString(e_); return 1; } } while (0)

int32_t exchange(aero_local_group::Slot* s, const void* send, void* recv, uint64_t bytes, int kind) {
    aero_local_group* g = s->g;
    if (hipSetDevice(s->device) != hipSuccess) { s->err = "hipSetDevice failed"; return 1; }
    s->send = send; s->bytes = bytes;
    LOCAL_HIP(hipEventRecord(s->ready, s->stream));
    if (!g->rendezvous()) { s->err = "the group was aborted (a peer failed)"; return 1; }
    uint8_t* d = static_cast<uint8_t*>(recv);
    for (uint32_t p = 0; p < g->world; p++) {
        const aero_local_group::Slot& peer = g->slots[p];
        if (peer.bytes != bytes) { s->err = "peers disagree on the exchange size"; g->rendezvous(); return 1; }
        if ((int)p != s->rank) LOCAL_HIP(hipStreamWaitEvent(s->stream, peer.ready, 0));
        const uint8_t* src = static_cast<const uint8_t*>(peer.send) + (kind == 0 ? (size_t)s->rank * bytes : 0);
        // an in-place all-gather names the rank's own piece as source AND destination: copying it onto itself would be a write to bytes
        // the peers are reading at that very moment (the same values, but a race all the same: ThreadSanitizer over tools/hipstub)
        if (src == d + (size_t)p * bytes) continue;
        LOCAL_HIP(hipMemcpyAsync(d + (size_t)p * bytes, src, bytes, hipMemcpyDeviceToDevice, s->stream));
    }
    LOCAL_HIP(hipEventRecord(s->done, s->stream));
    if (!g->rendezvous()) { s->err = "the group was aborted (a peer failed)"; return 1; }
    for (uint32_t p = 0; p < g->world; p++)
        if ((int)p != s->rank) LOCAL_HIP(hipStreamWaitEvent(s->stream, g->slots[p].done, 0));
    s->calls[kind] += 1;
    s->bytes_sent += bytes * (g->world - 1);
    return 0;
}
int32_t local_all_to_all(void* user, const void* send, void* recv, uint64_t bytes) { return exchange(static_cast<aero_local_group::Slot*>(user), send, recv, bytes, 0); }
int32_t local_all_gather(void* user, const void* send, void* recv, uint64_t bytes) { return exchange(static_cast<aero_local_group::Slot*>(user), send, recv, bytes, 1); }
// one chunk to `to`, one chunk from `from`: the peer that reads this rank's buffer is `to`
int32_t local_send_recv(void* user, const void* send, int32_t to, void* recv, int32_t from, uint64_t bytes) {
    aero_local_group::Slot* s = static_cast<aero_local_group::Slot*>(user);
    aero_local_group* g = s->g;
    if (hipSetDevice(s->device) != hipSuccess) { s->err = "hipSetDevice failed"; return 1; }
    if (to < 0 || from < 0 || (uint32_t)to >= g->world || (uint32_t)from >= g->world) { s->err = "send_recv: bad peer"; return 1; }
    s->send = send; s->bytes = bytes;
    LOCAL_HIP(hipEventRecord(s->ready, s->stream));
    if (!g->rendezvous()) { s->err = "the group was aborted (a peer failed)"; return 1; }
    const aero_local_group::Slot& peer = g->slots[from];
    if (peer.bytes != bytes) { s->err = "peers disagree on the send_recv size"; g->rendezvous(); return 1; }
    if (from != s->rank) LOCAL_HIP(hipStreamWaitEvent(s->stream, peer.ready, 0));
    LOCAL_HIP(hipMemcpyAsync(recv, peer.send, bytes, hipMemcpyDeviceToDevice, s->stream));
    LOCAL_HIP(hipEventRecord(s->done, s->stream));
    if (!g->rendezvous()) { s->err = "the group was aborted (a peer failed)"; return 1; }
    if (to != s->rank) LOCAL_HIP(hipStreamWaitEvent(s->stream, g->slots[to].done, 0));
    s->calls[0] += 1;
    s->bytes_sent += bytes;
    return 0;
}
int32_t local_all_reduce(void* user, void* buf, uint64_t count) {
    aero_local_group::Slot* s = static_cast<aero_local_group::Slot*>(user);
    aero_local_group* g = s->g;
    if (hipSetDevice(s->device) != hipSuccess) { s->err = "hipSetDevice failed"; return 1; }
    if (count > s->stage_cap) {
        // growing the staging block: nothing of an earlier all_reduce can still be in flight on a peer (they waited for `done`)
        if (s->stage) LOCAL_HIP(hipFree(s->stage));
        s->stage = nullptr;
        LOCAL_HIP(hipMalloc((void**)&s->stage, count * 8));
        s->stage_cap = count;
    }
    LOCAL_HIP(hipMemcpyAsync(s->stage, buf, count * 8, hipMemcpyDeviceToDevice, s->stream));
    s->send = s->stage; s->bytes = count;
    LOCAL_HIP(hipEventRecord(s->ready, s->stream));
    if (!g->rendezvous()) { s->err = "the group was aborted (a peer failed)"; return 1; }
    std::vector<const uint64_t*> ptrs(g->world);
    for (uint32_t p = 0; p < g->world; p++) {
        if (g->slots[p].bytes != count) { s->err = "peers disagree on the all_reduce size"; g->rendezvous(); return 1; }
        ptrs[p] = static_cast<const uint64_t*>(g->slots[p].send);
        if ((int)p != s->rank) LOCAL_HIP(hipStreamWaitEvent(s->stream, g->slots[p].ready, 0));
    }
    LOCAL_HIP(hipMemcpyAsync(s->srcs, ptrs.data(), g->world * sizeof(uint64_t*), hipMemcpyHostToDevice, s->stream));
    LOCAL_HIP(hipStreamSynchronize(s->stream));          // `ptrs` is a stack array: the copy must have left it (the one wait of this file)
    size_t blocks = (count + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(local_sum_u64_kernel, dim3((unsigned)blocks), dim3(256), 0, s->stream, static_cast<uint64_t*>(buf), s->srcs, g->world, (size_t)count);
    LOCAL_HIP(hipGetLastError());
    LOCAL_HIP(hipEventRecord(s->done, s->stream));
    if (!g->rendezvous()) { s->err = "the group was aborted (a peer failed)"; return 1; }
    for (uint32_t p = 0; p < g->world; p++)
        if ((int)p != s->rank) LOCAL_HIP(hipStreamWaitEvent(s->stream, g->slots[p].done, 0));
    s->calls[2] += 1;
    s->bytes_sent += count * 8;
    return 0;
}

}  // namespace

extern "C" {

int32_t aero_local_group_create(uint32_t world, aero_local_group** out) {
    if (!out || world < 1 || world > 128 || (world & (world - 1))) return AERO_E_BAD_ARG;
    aero_local_group* g = new (std::nothrow) aero_local_group();
    if (!g) return AERO_E_OOM;
    g->world = world;
    g->slots.resize(world);
    for (uint32_t r = 0; r < world; r++) { g->slots[r].g = g; g->slots[r].rank = (int)r; }
    *out = g;
    return AERO_OK;
}
int32_t aero_local_group_comm(aero_local_group* g, aero_ctx* ctx, int32_t rank, uint32_t min_peer_digests, aero_comm* out) {
    if (!g || !ctx || !ctx->c || !out || rank < 0 || (uint32_t)rank >= g->world) return AERO_E_BAD_ARG;
    aero_local_group::Slot& s = g->slots[rank];
    // ranks bind from their own threads, possibly at the same moment, and each looks at the slots bound so far (peer access): under the
    // group's mutex - two ranks that bound simultaneously could otherwise each see the other as unbound and neither enable peer access
    // (found by ThreadSanitizer over the stand-in runtime, tools/hipstub)
    std::lock_guard<std::mutex> bind_lock(g->mu);
    if (!s.bound) {
        if (hipSetDevice(ctx->c->device) != hipSuccess) return AERO_E_HIP;
        s.device = ctx->c->device; s.stream = ctx->c->stream;
        if (hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess ||
            hipMalloc((void**)&s.srcs, g->world * sizeof(uint64_t*)) != hipSuccess) { (void)hipGetLastError(); return AERO_E_HIP; }
        // ranks on different GPUs read each other's buffers directly (xGMI / PCIe peer access)
        for (uint32_t p = 0; p < g->world; p++)
            if (g->slots[p].bound && g->slots[p].device != s.device) {
                (void)hipDeviceEnablePeerAccess(g->slots[p].device, 0); (void)hipGetLastError();
                (void)hipSetDevice(g->slots[p].device); (void)hipDeviceEnablePeerAccess(s.device, 0); (void)hipGetLastError();
                (void)hipSetDevice(s.device);
            }
        s.bound = true;
    }
    out->rank = rank; out->world = (int32_t)g->world; out->user = &s;
    out->all_to_all = local_all_to_all; out->all_gather = local_all_gather; out->all_reduce_sum_u64 = local_all_reduce;
    out->min_peer_digests = min_peer_digests; out->flags = AERO_COMM_STREAM_ORDERED; out->send_recv = local_send_recv;
    return AERO_OK;
}
int32_t aero_local_group_stats(const aero_local_group* g, int32_t rank, uint64_t out[4]) {
    if (!g || !out || rank < 0 || (uint32_t)rank >= g->world) return AERO_E_BAD_ARG;
    const aero_local_group::Slot& s = g->slots[rank];
    out[0] = s.calls[0]; out[1] = s.calls[1]; out[2] = s.calls[2]; out[3] = s.bytes_sent;
    return AERO_OK;
}
const char* aero_local_group_last_error(const aero_local_group* g, int32_t rank) {
    if (!g || rank < 0 || (uint32_t)rank >= g->world) return "";
    return g->slots[rank].err.c_str();
}
void aero_local_group_abort(aero_local_group* g) {
    if (!g) return;
    std::lock_guard<std::mutex> lk(g->mu);
    g->aborted = true;
    g->cv.notify_all();
}
void aero_local_group_destroy(aero_local_group* g) {
    if (!g) return;
    for (auto& s : g->slots) {
        if (!s.bound) continue;
        (void)hipSetDevice(s.device);
        if (s.ready) (void)hipEventDestroy(s.ready);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.stage) (void)hipFree(s.stage);
        if (s.srcs) (void)hipFree(s.srcs);
    }
    delete g;
}

// ONE proof by `world` ranks of this process: a context and a host thread per rank (device_ids[r]; the same id may repeat - ranks
// then share that GPU), the trace in HOST memory, exchanges through a local group. Every rank's proof comes back (they must be
// identical, and identical to the single-GPU proof); rank_ms[r] = that rank's wall-clock.
int32_t aero_prove_fib_sharded_local(const int32_t* device_ids, uint32_t world, const uint64_t* trace_col_major, uint32_t width, uint32_t log_n,
                                     const aero_fib_air* air, const aero_proof_options* options, uint32_t min_peer_digests, uint8_t** proofs,
                                     size_t* proof_lens, uint64_t* pub_out, double* rank_ms, uint64_t* bytes_sent, char* err, size_t err_cap) {
    auto put = [&](const std::string& s) { if (err && err_cap) { const size_t k = std::min(err_cap - 1, s.size()); memcpy(err, s.data(), k); err[k] = 0; } };
    if (!device_ids || !trace_col_major || !options || !proofs || !proof_lens || world < 1) { put("prove_fib_sharded_local: null argument"); return AERO_E_BAD_ARG; }
    aero_local_group* g = nullptr;
    int32_t rc = aero_local_group_create(world, &g);
    if (rc != AERO_OK) { put("prove_fib_sharded_local: world must be a power of two in [1, 128]"); return rc; }
    std::vector<aero_ctx*> ctxs(world, nullptr);
    for (uint32_t r = 0; r < world && rc == AERO_OK; r++) {
        rc = aero_ctx_create(device_ids[r], &ctxs[r]);
        if (rc != AERO_OK) put(aero_last_error(nullptr));
    }
    std::vector<int32_t> status(world, AERO_OK);
    std::vector<std::string> msgs(world);
    if (rc == AERO_OK) {
        std::vector<std::thread> th;
        for (uint32_t r = 0; r < world; r++) {
            proofs[r] = nullptr; proof_lens[r] = 0;
            th.emplace_back([&, r] {
                try {
                aero_comm comm{};
                int32_t st = aero_local_group_comm(g, ctxs[r], (int32_t)r, min_peer_digests, &comm);
                const auto t0 = std::chrono::steady_clock::now();
                std::vector<uint64_t> pub(width / 2 + 1);
                if (st == AERO_OK) st = aero_prove_fib_sharded_host(ctxs[r], &comm, trace_col_major, width, log_n, air, options, &proofs[r], &proof_lens[r], pub.data());
                if (rank_ms) rank_ms[r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                status[r] = st;
                if (st != AERO_OK) { msgs[r] = std::string(aero_last_error(ctxs[r])) + " " + aero_local_group_last_error(g, (int32_t)r); aero_local_group_abort(g); }
                else if (r == 0 && pub_out) memcpy(pub_out, pub.data(), (width / 2) * 8);
                } catch (...) {   // never out of a thread body (std::terminate would take the host process down); peers must not wait for this rank
                    status[r] = AERO_E_INTERNAL;
                    try { msgs[r] = "unexpected exception in the rank's thread"; } catch (...) {}
                    aero_local_group_abort(g);
                }
            });
        }
        for (auto& t : th) t.join();
        for (uint32_t r = 0; r < world; r++) {
            if (bytes_sent) { uint64_t st[4]; aero_local_group_stats(g, (int32_t)r, st); bytes_sent[r] = st[3]; }
            if (status[r] != AERO_OK && rc == AERO_OK) rc = status[r];
        }
        if (rc != AERO_OK) {
            std::string all;
            for (uint32_t r = 0; r < world; r++) if (status[r] != AERO_OK) all += "rank " + std::to_string(r) + " (" + std::to_string(status[r]) + "): " + msgs[r] + "; ";
            put(all);
        }
        if (rc != AERO_OK) for (uint32_t r = 0; r < world; r++) { free(proofs[r]); proofs[r] = nullptr; proof_lens[r] = 0; }
    }
    aero_local_group_destroy(g);
    for (aero_ctx* c : ctxs) if (c) aero_ctx_destroy(c);
    return rc;
}

}  // extern "C"
