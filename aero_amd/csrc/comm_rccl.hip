// Native RCCL implementation of the exchange steps of a sharded proof (include/aero_stark.h: aero_comm / aero_rccl_*).
//
// One process per GPU; every rank creates ONE communicator next to its context (`ncclCommInitRank` with the 128-byte id
// rank 0 produced) and hands the resulting aero_comm to aero_prove_fib_sharded / aero_prove_fib_air. All three exchanges are
// enqueued on the CONTEXT'S OWN STREAM and are therefore stream-ordered with the kernels that produce / consume the buffers
// (flag AERO_COMM_STREAM_ORDERED: the prover neither synchronises before the exchange nor waits after it):
//   all_to_all .......... ncclGroupStart; world x (ncclSend + ncclRecv); ncclGroupEnd   — the leaf-digest exchange per commitment
//   all_gather .......... ncclAllGather                                                   — subtree roots, un-sharded FRI layer
//   all_reduce_sum_u64 .. ncclAllReduce(ncclUint64, ncclSum)                              — the opening block
// xGMI is point-to-point: the all-to-all uses all 7 links of a GPU at once, none of the exchanges is a ring all-reduce of
// bulk data (the one all-reduce carries a few hundred KiB).
//
// librccl is bound at run time (dlopen), not at link time: libaero_stark.so loads on a box without RCCL as long as no sharded
// proof is requested, and the copy that is bound is the one living next to the HIP runtime this library itself uses (see rccl_api).
#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// librccl is bound with dlopen; without its headers the few types of its C API this file names are declared here (NCCL 2.x ABI)
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#endif

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/aero_stark.h"
#include "aero_internal.hpp"

using namespace aero;

namespace {

struct RcclApi {
    void* handle = nullptr;
    std::string err;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    // optional (aero_rccl_info): what the communicator itself reports
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
};

RcclApi g_api;
std::mutex g_api_mu;
thread_local std::string g_rccl_err;

template <class T> bool bind(void* h, const char* name, T& fn) {
    fn = reinterpret_cast<T>(dlsym(h, name));
    return fn != nullptr;
}

// Returns nullptr (and g_rccl_err) when no usable RCCL is present.
RcclApi* rccl_api() {
    std::lock_guard<std::mutex> lk(g_api_mu);
    if (g_api.handle) return &g_api;
    if (!g_api.err.empty()) { g_rccl_err = g_api.err; return nullptr; }
    // The RCCL to bind is the one that runs on the HIP runtime THIS library runs on. A process can hold two runtimes (PyTorch
    // bundles libamdhip64 + librccl in torch/lib; imported after this library it brings a second, separate runtime): binding
    // "whatever librccl is loaded" could pick the copy of the other runtime, whose ncclCommInitRank then fails. So: look next
    // to our own libamdhip64 first (torch/lib when torch came first and we share its runtime, /opt/rocm/lib otherwise).
    std::vector<std::string> names;
    {
        Dl_info info;
        if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                names.push_back(dir + "librccl.so.1");
                names.push_back(dir + "librccl.so");
            }
        }
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
    }
    void* h = nullptr;
    for (const std::string& n : names) if ((h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) { g_api.err = std::string("RCCL is not available: ") + (dlerror() ? dlerror() : "librccl.so.1 not found"); g_rccl_err = g_api.err; return nullptr; }
    RcclApi a;
    const bool ok = bind(h, "ncclGetUniqueId", a.GetUniqueId) && bind(h, "ncclCommInitRank", a.CommInitRank) && bind(h, "ncclCommDestroy", a.CommDestroy) &&
                    bind(h, "ncclGroupStart", a.GroupStart) && bind(h, "ncclGroupEnd", a.GroupEnd) && bind(h, "ncclSend", a.Send) && bind(h, "ncclRecv", a.Recv) &&
                    bind(h, "ncclAllGather", a.AllGather) && bind(h, "ncclAllReduce", a.AllReduce) && bind(h, "ncclGetErrorString", a.GetErrorString);
    if (!ok) { g_api.err = "RCCL library lacks a required entry point"; g_rccl_err = g_api.err; return nullptr; }
    (void)bind(h, "ncclCommCount", a.CommCount);
    (void)bind(h, "ncclCommUserRank", a.CommUserRank);
    (void)bind(h, "ncclCommCuDevice", a.CommCuDevice);
    a.handle = h;
    g_api = a;
    return &g_api;
}

}  // namespace

struct aero_rccl {
    RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;   // the owning context's stream (the context must outlive this object)
    int rank = 0, world = 1;
    uint64_t calls[3] = {0, 0, 0};  // all_to_all, all_gather, all_reduce
    uint64_t bytes_sent = 0;
    std::string err;

    bool ok(ncclResult_t r, const char* what) {
        if (r == ncclSuccess) return true;
        err = std::string(what) + ": " + (api->GetErrorString ? api->GetErrorString(r) : "RCCL error");
        return false;
    }
};

static int32_t rccl_all_to_all(void* user, const void* send, void* recv, uint64_t bytes) {
    aero_rccl* r = static_cast<aero_rccl*>(user);
    if (hipSetDevice(r->device) != hipSuccess) { r->err = "all_to_all: hipSetDevice failed"; return 1; }
    const uint8_t* s = static_cast<const uint8_t*>(send);
    uint8_t* d = static_cast<uint8_t*>(recv);
    if (!r->ok(r->api->GroupStart(), "all_to_all: ncclGroupStart")) return 1;
    bool good = true;
    for (int p = 0; p < r->world && good; p++) {
        good = r->ok(r->api->Send(s + (size_t)p * bytes, bytes, ncclUint8, p, r->comm, r->stream), "all_to_all: ncclSend") &&
               r->ok(r->api->Recv(d + (size_t)p * bytes, bytes, ncclUint8, p, r->comm, r->stream), "all_to_all: ncclRecv");
    }
    const ncclResult_t e = r->api->GroupEnd();   // always close the group
    if (!good) return 1;
    if (!r->ok(e, "all_to_all: ncclGroupEnd")) return 1;
    r->calls[0]++;
    r->bytes_sent += bytes * (uint64_t)(r->world - 1);
    return 0;
}
static int32_t rccl_send_recv(void* user, const void* send, int32_t to, void* recv, int32_t from, uint64_t bytes) {
    aero_rccl* r = static_cast<aero_rccl*>(user);
    if (hipSetDevice(r->device) != hipSuccess) { r->err = "send_recv: hipSetDevice failed"; return 1; }
    if (!r->ok(r->api->GroupStart(), "send_recv: ncclGroupStart")) return 1;
    const bool good = r->ok(r->api->Send(send, bytes, ncclUint8, to, r->comm, r->stream), "send_recv: ncclSend") &&
                      r->ok(r->api->Recv(recv, bytes, ncclUint8, from, r->comm, r->stream), "send_recv: ncclRecv");
    const bool ended = r->ok(r->api->GroupEnd(), "send_recv: ncclGroupEnd");
    if (!good || !ended) return 1;
    r->calls[0] += 1;
    r->bytes_sent += bytes;
    return 0;
}
static int32_t rccl_all_gather(void* user, const void* send, void* recv, uint64_t bytes) {
    aero_rccl* r = static_cast<aero_rccl*>(user);
    if (hipSetDevice(r->device) != hipSuccess) { r->err = "all_gather: hipSetDevice failed"; return 1; }
    if (!r->ok(r->api->AllGather(send, recv, bytes, ncclUint8, r->comm, r->stream), "all_gather: ncclAllGather")) return 1;
    r->calls[1]++;
    r->bytes_sent += bytes * (uint64_t)(r->world - 1);
    return 0;
}
static int32_t rccl_all_reduce(void* user, void* buf, uint64_t count) {
    aero_rccl* r = static_cast<aero_rccl*>(user);
    if (hipSetDevice(r->device) != hipSuccess) { r->err = "all_reduce: hipSetDevice failed"; return 1; }
    if (!r->ok(r->api->AllReduce(buf, buf, count, ncclUint64, ncclSum, r->comm, r->stream), "all_reduce: ncclAllReduce")) return 1;
    r->calls[2]++;
    r->bytes_sent += count * 8;
    return 0;
}

// aero_ctx is defined in capi.hip; only its stream / device are needed here
namespace aero { Context* ctx_of(aero_ctx* c); }

extern "C" {

// Is a usable RCCL present in this process? Binds the library, creates nothing (a rank other than rank 0 must not ask RCCL for a
// unique id just to find out: that starts a bootstrap listener it will never use).
int32_t aero_rccl_available(void) { return rccl_api() ? AERO_OK : AERO_E_COMM; }

int32_t aero_rccl_unique_id(uint8_t id_out[AERO_RCCL_ID_BYTES]) {
    if (!id_out) return AERO_E_BAD_ARG;
    RcclApi* api = rccl_api();
    if (!api) return AERO_E_COMM;
    static_assert(sizeof(ncclUniqueId) == AERO_RCCL_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    const ncclResult_t e = api->GetUniqueId(&id);
    if (e != ncclSuccess) { g_rccl_err = std::string("ncclGetUniqueId: ") + api->GetErrorString(e); return AERO_E_COMM; }
    memcpy(id_out, &id, sizeof id);
    return AERO_OK;
}

int32_t aero_rccl_create(aero_ctx* ctx, int32_t rank, int32_t world, const uint8_t id[AERO_RCCL_ID_BYTES], aero_rccl** out) {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) { g_rccl_err = "rccl_create: bad argument"; return AERO_E_BAD_ARG; }
    *out = nullptr;
    RcclApi* api = rccl_api();
    if (!api) return AERO_E_COMM;
    Context* c = ctx_of(ctx);
    if (!c) { g_rccl_err = "rccl_create: null context"; return AERO_E_BAD_ARG; }
    if (hipSetDevice(c->device) != hipSuccess) { g_rccl_err = "rccl_create: hipSetDevice failed"; return AERO_E_HIP; }
    aero_rccl* r = new (std::nothrow) aero_rccl();
    if (!r) return AERO_E_OOM;
    r->api = api; r->device = c->device; r->stream = c->stream; r->rank = rank; r->world = world;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    const ncclResult_t e = api->CommInitRank(&r->comm, world, uid, rank);
    if (e != ncclSuccess) {
        g_rccl_err = std::string("ncclCommInitRank: ") + api->GetErrorString(e);
        delete r;
        return AERO_E_COMM;
    }
    *out = r;
    return AERO_OK;
}

int32_t aero_rccl_comm(aero_rccl* r, uint32_t min_peer_digests, aero_comm* out) {
    if (!r || !out) return AERO_E_BAD_ARG;
    memset(out, 0, sizeof *out);
    out->rank = r->rank; out->world = r->world; out->user = r;
    out->all_to_all = rccl_all_to_all; out->all_gather = rccl_all_gather; out->all_reduce_sum_u64 = rccl_all_reduce;
    out->min_peer_digests = min_peer_digests;
    out->flags = AERO_COMM_STREAM_ORDERED;
    out->send_recv = rccl_send_recv;
    return AERO_OK;
}

int32_t aero_rccl_stats(const aero_rccl* r, uint64_t out[4]) {
    if (!r || !out) return AERO_E_BAD_ARG;
    out[0] = r->calls[0]; out[1] = r->calls[1]; out[2] = r->calls[2]; out[3] = r->bytes_sent;
    return AERO_OK;
}

int32_t aero_rccl_info(const aero_rccl* r, int32_t out[4]) {
    if (!r || !out) return AERO_E_BAD_ARG;
    out[0] = out[1] = out[2] = -1;
    out[3] = r->world;
    if (r->comm && r->api) {
        int v = -1;
        if (r->api->CommCount && r->api->CommCount(r->comm, &v) == ncclSuccess) out[0] = v;
        if (r->api->CommUserRank && r->api->CommUserRank(r->comm, &v) == ncclSuccess) out[1] = v;
        if (r->api->CommCuDevice && r->api->CommCuDevice(r->comm, &v) == ncclSuccess) out[2] = v;
    }
    return AERO_OK;
}

const char* aero_rccl_last_error(const aero_rccl* r) { return r ? r->err.c_str() : g_rccl_err.c_str(); }

void aero_rccl_destroy(aero_rccl* r) {
    if (!r) return;
    if (r->comm) {
        (void)hipSetDevice(r->device);
        (void)hipStreamSynchronize(r->stream);
        (void)r->api->CommDestroy(r->comm);
    }
    delete r;
}

}  // extern "C"
