// A stand-in HIP runtime for running the HOST logic of libaero_stark under ThreadSanitizer / AddressSanitizer on a box without a GPU
// (handle lifetimes, pool allocator, staging allocator, thread pool, local-group rendezvous).
//
// Semantics kept: a stream is an ordered queue executed by its own worker thread; copies and memsets really move bytes; events
// order streams (record = marker in the recording stream, wait = the waiting stream blocks until that marker has been reached);
// a host-to-device copy reads its (pageable) source before the call returns, like the real runtime's staging does. Kernels do NOT
// run: a launch is an empty queue entry. Whatever the library orders through events is therefore ordered here the same way - and a
// copy the library forgot to order shows up as a data race between two stream workers under TSan.
// "Device memory" is calloc'ed host memory; HIPSTUB_DEVICES (default 1) devices exist. TEST INFRASTRUCTURE, never shipped.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

namespace {
struct Stream {
    std::mutex m;
    std::condition_variable cv, idle;
    std::deque<std::function<void()>> q;
    bool stop = false, busy = false;
    std::thread th;
    Stream() { th = std::thread([this] { run(); }); }
    ~Stream() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        th.join();
    }
    void run() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
                busy = true;
            }
            f();
            {
                std::lock_guard<std::mutex> lk(m);
                busy = false;
            }
            idle.notify_all();
        }
    }
    void push(std::function<void()> f) {
        { std::lock_guard<std::mutex> lk(m); q.push_back(std::move(f)); }
        cv.notify_one();
    }
    void sync() {
        std::unique_lock<std::mutex> lk(m);
        idle.wait(lk, [&] { return q.empty() && !busy; });
    }
};
struct EventState {
    std::mutex m;
    std::condition_variable cv;
    uint64_t recorded = 0, completed = 0;
};
struct Event { std::shared_ptr<EventState> st = std::make_shared<EventState>(); };    // queued records / waits keep the state alive past hipEventDestroy (the real runtime defers the release too)
std::mutex g_mu;
std::set<std::shared_ptr<Stream>> g_streams;       // shared: a device-wide synchronisation in one thread survives another thread destroying its stream
Stream* g_null_stream = nullptr;
thread_local int t_device = 0;
std::atomic<uint64_t> g_launches{0}, g_copies{0};

Stream* S(hipStream_t s) {
    if (s) return reinterpret_cast<Stream*>(s);
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_null_stream) { auto sp = std::make_shared<Stream>(); g_null_stream = sp.get(); g_streams.insert(sp); }
    return g_null_stream;
}
void sync_all() {
    std::vector<std::shared_ptr<Stream>> v;
    { std::lock_guard<std::mutex> lk(g_mu); v.assign(g_streams.begin(), g_streams.end()); }
    for (auto& s : v) s->sync();
}
int device_count() { const char* e = getenv("HIPSTUB_DEVICES"); return e ? atoi(e) : 1; }
}  // namespace

extern "C" {
uint64_t hipstub_launches() { return g_launches.load(); }
uint64_t hipstub_copies() { return g_copies.load(); }

hipError_t hipGetDeviceCount(int* n) { *n = device_count(); return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= device_count()) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hipstub error"; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipDeviceGetPCIBusId(char* buf, int len, int d) { snprintf(buf, (size_t)len, "0000:%02x:00.0", 0xc1 + d); return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { sync_all(); free(p); return hipSuccess; }      // hipFree synchronises the device
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { sync_all(); free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    auto sp = std::make_shared<Stream>();
    { std::lock_guard<std::mutex> lk(g_mu); g_streams.insert(sp); }
    *s = reinterpret_cast<hipStream_t>(sp.get());
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    Stream* st = reinterpret_cast<Stream*>(s);
    st->sync();
    std::shared_ptr<Stream> keep;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto it = g_streams.begin(); it != g_streams.end(); ++it) if (it->get() == st) { keep = *it; g_streams.erase(it); break; }
    }
    return hipSuccess;       // the worker thread ends with the last owner
}
hipError_t hipStreamSynchronize(hipStream_t s) { S(s)->sync(); return hipSuccess; }
hipError_t hipDeviceSynchronize() { sync_all(); return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) {
    Stream* st = &*S(s);
    std::lock_guard<std::mutex> lk(st->m);
    return (st->q.empty() && !st->busy) ? hipSuccess : hipErrorNotReady;
}

hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(new Event()); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<Event*>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    std::shared_ptr<EventState> ev = reinterpret_cast<Event*>(e)->st;
    uint64_t seq;
    { std::lock_guard<std::mutex> lk(ev->m); seq = ++ev->recorded; }
    S(s)->push([ev, seq] {
        { std::lock_guard<std::mutex> lk(ev->m); if (ev->completed < seq) ev->completed = seq; }
        ev->cv.notify_all();
    });
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    std::shared_ptr<EventState> ev = reinterpret_cast<Event*>(e)->st;
    uint64_t target;
    { std::lock_guard<std::mutex> lk(ev->m); target = ev->recorded; }
    S(s)->push([ev, target] {
        std::unique_lock<std::mutex> lk(ev->m);
        ev->cv.wait(lk, [&] { return ev->completed >= target; });
    });
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    std::shared_ptr<EventState> ev = reinterpret_cast<Event*>(e)->st;
    std::unique_lock<std::mutex> lk(ev->m);
    const uint64_t target = ev->recorded;
    ev->cv.wait(lk, [&] { return ev->completed >= target; });
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s) {
    g_copies++;
    if (kind == hipMemcpyHostToDevice) {
        // the runtime stages a pageable source before it returns: the caller may reuse the buffer at once
        std::shared_ptr<std::vector<char>> tmp = std::make_shared<std::vector<char>>((const char*)src, (const char*)src + n);
        S(s)->push([dst, tmp, n] { memcpy(dst, tmp->data(), n); });
    } else {
        S(s)->push([dst, src, n] { memcpy(dst, src, n); });
    }
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t s) {
    g_copies++;
    S(s)->push([=] { for (size_t r = 0; r < height; r++) memcpy((char*)dst + r * dpitch, (const char*)src + r * spitch, width); });
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t s) { S(s)->push([=] { memset(dst, v, n); }); return hipSuccess; }

// kernels: queue entries that do nothing
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t s) { g_launches++; S(s)->push([] {}); return hipSuccess; }
hipError_t hipModuleLaunchKernel(hipFunction_t, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t s, void**, void**) {
    g_launches++; S(s)->push([] {}); return hipSuccess;
}
hipError_t hipModuleLoadData(hipModule_t* m, const void*) { *m = reinterpret_cast<hipModule_t>(new int(0)); return hipSuccess; }
hipError_t hipModuleUnload(hipModule_t m) { delete reinterpret_cast<int*>(m); return hipSuccess; }
hipError_t hipModuleGetFunction(hipFunction_t* f, hipModule_t m, const char*) { *f = reinterpret_cast<hipFunction_t>(m); return hipSuccess; }

// the guard-page allocator's calls: not available here (AERO_POOL_GUARD is a GPU-box diagnosis)
hipError_t hipMemGetAllocationGranularity(size_t*, const hipMemAllocationProp*, hipMemAllocationGranularity_flags) { return hipErrorNotSupported; }
hipError_t hipMemAddressReserve(void**, size_t, size_t, void*, unsigned long long) { return hipErrorNotSupported; }
hipError_t hipMemAddressFree(void*, size_t) { return hipErrorNotSupported; }
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t*, size_t, const hipMemAllocationProp*, unsigned long long) { return hipErrorNotSupported; }
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t) { return hipErrorNotSupported; }
hipError_t hipMemMap(void*, size_t, size_t, hipMemGenericAllocationHandle_t, unsigned long long) { return hipErrorNotSupported; }
hipError_t hipMemUnmap(void*, size_t) { return hipErrorNotSupported; }
hipError_t hipMemSetAccess(void*, size_t, const hipMemAccessDesc*, size_t) { return hipErrorNotSupported; }

// what the compiler-generated registration code and the <<< >>> lowering call
struct CallCfg { dim3 g, b; size_t shmem; hipStream_t s; };
static thread_local CallCfg t_cfg;
void** __hipRegisterFatBinary(const void*) { static void* h = nullptr; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t shmem, hipStream_t s) { t_cfg = CallCfg{g, b, shmem, s}; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* shmem, hipStream_t* s) { *g = t_cfg.g; *b = t_cfg.b; *shmem = t_cfg.shmem; *s = t_cfg.s; return hipSuccess; }
}  // extern "C"
