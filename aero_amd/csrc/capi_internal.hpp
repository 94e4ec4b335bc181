// Handle types and the exception guard shared by the C-ABI translation units (capi.hip, capi_air.hip).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <memory>

#include "../../include/aero_stark.h"
#include "prover.hpp"

using namespace aero;

struct aero_ctx {
    std::shared_ptr<Context> keep;   // device objects created from this context share ownership, so destroying the
    Context* c = nullptr;            // context handle before its matrices / trees is safe
    std::string err;
    StageMs last_ms;
    bool stage_timing = false;
    // set by a pool while several of its slots prove at once: other proofs' kernels then overlap this proof's host-to-device copy,
    // and a second stream per proof only adds cross-stream waits (measured, 2^20 x 72, 8 in flight: 3.85 G cells/s against 3.3)
    bool concurrent_peers = false;
    int self_verify = AERO_SELF_VERIFY_AUTO;     // aero_ctx_set_self_verify
    double last_self_verify_ms = 0;
};
struct aero_matrix {
    std::shared_ptr<Context> keep;   // declared first: destroyed after `m`, whose buffers return to the context pool
    Matrix m;
    explicit aero_matrix(aero_ctx* ctx) : keep(ctx->keep) {}
};
struct aero_tree {
    std::shared_ptr<Context> keep;
    MerkleTree t;
    explicit aero_tree(aero_ctx* ctx) : keep(ctx->keep) {}
};

struct aero_fri {
    std::shared_ptr<Context> keep;
    FriLayers fl;
    ProofOptions opt{};
    explicit aero_fri(aero_ctx* ctx) : keep(ctx->keep) {}
};

extern thread_local std::string g_create_err;


// Work enqueued before a failure may still read the caller's (pinned) buffers and the scratch blocks: whatever went wrong, the
// streams drain before the scratch blocks return to the pool and the caller gets its memory back.
static inline void drain_after_failure(aero_ctx* ctx) {
    (void)hipGetLastError();   // leave no stale HIP error behind for the next call's launch checks
    if (!ctx || !ctx->c) return;
    if (ctx->c->copy_stream) (void)hipStreamSynchronize(ctx->c->copy_stream);
    (void)hipStreamSynchronize(ctx->c->stream);
    (void)hipGetLastError();
    ctx->c->scratch_reset();
}
template <class Fn> static int32_t guard(aero_ctx* ctx, Fn&& fn) {
    try {
        if (!ctx || !ctx->c) { g_create_err = "null context"; return AERO_E_BAD_ARG; }
        AERO_HIP(hipSetDevice(ctx->c->device));
        fn();
        return AERO_OK;
    } catch (const Error& e) {
        ctx->err = e.what();
        drain_after_failure(ctx);
        return e.code;
    } catch (const std::bad_alloc&) {
        ctx->err = "host allocation failed";
        drain_after_failure(ctx);
        return AERO_E_OOM;
    } catch (const std::exception& e) {
        ctx->err = e.what();
        drain_after_failure(ctx);
        return AERO_E_INTERNAL;
    } catch (...) {   // nothing may cross the C boundary
        ctx->err = "unknown exception";
        drain_after_failure(ctx);
        return AERO_E_INTERNAL;
    }
}
#define REQUIRE(cond, msg) do { if (!(cond)) fail(msg); } while (0)

// the aero_comm of a *_sharded call as the prover takes it
static inline ShardComm shard_comm_of(const aero_comm* comm, const char* who) {
    if (!(comm->world >= 1 && comm->rank >= 0 && comm->rank < comm->world)) fail(std::string(who) + ": bad rank / world");
    if (!(comm->world == 1 || (comm->all_to_all && comm->all_gather && comm->all_reduce_sum_u64))) fail(std::string(who) + ": missing exchange callback");
    ShardComm sc;
    sc.rank = comm->rank; sc.world = comm->world; sc.user = comm->user;
    sc.all_to_all = comm->all_to_all; sc.all_gather = comm->all_gather; sc.all_reduce_sum_u64 = comm->all_reduce_sum_u64; sc.send_recv = comm->send_recv;
    sc.min_peer_digests = comm->min_peer_digests ? comm->min_peer_digests : 2048;
    sc.stream_ordered = (comm->flags & AERO_COMM_STREAM_ORDERED) != 0;
    return sc;
}
// prove-then-verify at the boundary (include/aero_stark.h: aero_ctx_set_self_verify)
static inline bool self_verify_wanted(const aero_ctx* ctx, const aero_comm* comm) {
    if (ctx->self_verify == AERO_SELF_VERIFY_AUTO) return comm && comm->world > 1;
    return ctx->self_verify != AERO_SELF_VERIFY_OFF;
}
namespace aero {
// verify.hip: the library's verifier on the bytes a prove call is about to return, the proof's options and trace length pinned to the
// call's; throws Error(AERO_E_SELF_VERIFY) with the verifier's reason when it rejects. Exactly one of air / prog is non-null.
void self_verify_or_throw(const uint8_t* proof, size_t len, const std::vector<uint64_t>& pub, const aero_fib_air* air, const air::Program* prog,
                          uint32_t log_n, const aero_proof_options& opt);
}
static inline void run_self_verify(aero_ctx* ctx, const Bytes& b, const std::vector<uint64_t>& pub, const aero_fib_air* air, const air::Program* prog,
                                   uint32_t log_n, const aero_proof_options& opt) {
    const auto t0 = std::chrono::steady_clock::now();
    self_verify_or_throw(b.data(), b.size(), pub, air, prog, log_n, opt);
    ctx->last_self_verify_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

static inline int ilog2u(uint64_t x) { int r = 0; while ((1ull << r) < x) r++; return r; }



// numa.hip: placement of a rank's host side next to its GPU
namespace aero {
int numa_node_of_device(int device);        // -1 = unknown (or AERO_NUMA=0)
int bind_thread_to_node(int node);          // calling thread -> the node's CPUs the process may use; returns their number, 0 = left alone
}
