// Host-side placement for a process that drives one GPU of a multi-socket node (include/aero_stark.h: aero_numa_*).
//
// The reference sizes its worker pool by `navigator.hardwareConcurrency` and spreads batches round-robin (aero-sdk/miden-wasm/src/pool.rs:28-45,
// 105-124); it has no notion of where a worker runs. On an 8-GPU MI355X node the host has two sockets: a rank whose worker threads and pinned
// trace buffers sit on the other socket pays the inter-socket hop on every launch and on every byte of the hand-over. So, INSIDE each rank
// process (no numactl, no re-exec - a process that has touched the GPU must not exec):
//   * the GPU's NUMA node is read from sysfs (/sys/bus/pci/devices/<bdf>/numa_node, bdf from hipDeviceGetPCIBusId),
//   * the pool's worker threads are bound (sched_setaffinity) to that node's CPUs, intersected with what the process is allowed,
//   * pinned trace buffers are allocated with the thread's memory policy preferring that node (set_mempolicy + hipHostMallocNumaUser).
// Everything degrades to "do nothing" when the node is unknown (-1: single-socket boxes, containers that hide the topology).
// The sysfs root is a parameter of the query so that the parsing is testable on a box without a GPU (tests/test_numa_cpu.py).
#include <dirent.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/aero_stark.h"
#include "aero_internal.hpp"

namespace aero {

// "0-15,32-47,64" -> cpu numbers; false on malformed text
bool parse_cpulist(const char* text, std::vector<int>& out) {
    out.clear();
    if (!text) return false;
    const char* p = text;
    while (*p == ' ' || *p == '\t') p++;
    if (*p == 0 || *p == '\n') return true;      // an empty list is a list
    for (;;) {
        char* end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p || a < 0 || a > 65535) return false;
        long b = a;
        p = end;
        if (*p == '-') {
            p++;
            b = strtol(p, &end, 10);
            if (end == p || b < a || b > 65535) return false;
            p = end;
        }
        for (long c = a; c <= b; c++) out.push_back((int)c);
        while (*p == ' ' || *p == '\t') p++;
        if (*p == ',') { p++; continue; }
        if (*p == 0 || *p == '\n') return true;
        return false;
    }
}
static bool read_text(const std::string& path, std::string& out) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    out = buf;
    return true;
}
// node of the PCI device `bdf` ("0000:c1:00.0", any case) under `root` ("/sys" on a real box); -1 = unknown
int numa_node_of_pci(const std::string& root, const std::string& bdf_in) {
    std::string bdf = bdf_in;
    for (auto& ch : bdf) if (ch >= 'A' && ch <= 'F') ch = (char)(ch - 'A' + 'a');
    std::string text;
    if (!read_text(root + "/bus/pci/devices/" + bdf + "/numa_node", text)) return -1;
    char* end = nullptr;
    const long v = strtol(text.c_str(), &end, 10);
    if (end == text.c_str() || v < 0 || v > 4095) return -1;
    return (int)v;
}
bool cpus_of_node(const std::string& root, int node, std::vector<int>& out) {
    std::string text;
    if (node < 0 || !read_text(root + "/devices/system/node/node" + std::to_string(node) + "/cpulist", text)) return false;
    return parse_cpulist(text.c_str(), out);
}
static const char* sysfs_root() { return "/sys"; }
int numa_node_of_device(int device) {
    static const bool off = getenv("AERO_NUMA") && getenv("AERO_NUMA")[0] == '0';
    if (off) return -1;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return numa_node_of_pci(sysfs_root(), bdf);
}
// binds the CALLING thread to the CPUs of `node` that the process may use; returns how many CPUs the thread ended up on (0 = left alone)
int bind_thread_to_node(int node) {
    std::vector<int> cpus;
    if (!cpus_of_node(sysfs_root(), node, cpus) || cpus.empty()) return 0;
    cpu_set_t allowed, want;
    CPU_ZERO(&allowed); CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return 0;
    int k = 0;
    for (int c : cpus) if (c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) { CPU_SET(c, &want); k++; }
    if (k == 0) return 0;                       // the node's CPUs are not ours (cgroup cpuset): stay where we are
    return sched_setaffinity(0, sizeof want, &want) == 0 ? k : 0;
}
// Memory policy of the calling thread for the length of one allocation: prefer `node` (MPOL_PREFERRED), then back to whatever the thread
// had before - a policy the host application or `numactl --membind / --interleave` set is the caller's, not ours to reset.
struct ScopedPreferNode {
    static constexpr unsigned long MAXNODE = 64 * 8 * sizeof(unsigned long);
    int saved_mode = 0;
    unsigned long saved_mask[64] = {0};
    bool saved = false, set = false;
    explicit ScopedPreferNode(int node) {
#if defined(SYS_set_mempolicy) && defined(SYS_get_mempolicy)
        if (node < 0 || node >= (int)MAXNODE) return;
        saved = syscall(SYS_get_mempolicy, &saved_mode, saved_mask, MAXNODE, nullptr, 0ul) == 0;
        unsigned long mask[64] = {0};
        mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
        set = syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, MAXNODE) == 0;
#else
        (void)node;
#endif
    }
    ~ScopedPreferNode() {
#if defined(SYS_set_mempolicy) && defined(SYS_get_mempolicy)
        if (!set) return;
        if (saved && saved_mode != 0) { if (syscall(SYS_set_mempolicy, saved_mode, saved_mask, MAXNODE) == 0) return; }
        (void)syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0);
#endif
    }
};

}  // namespace aero

using namespace aero;

extern "C" {

int32_t aero_numa_query(const char* sysfs_root_dir, const char* pci_bdf, int32_t* node_out, int32_t* cpus_out, uint32_t cpus_cap, uint32_t* n_cpus_out) {
    if (!sysfs_root_dir || !pci_bdf || !node_out) return AERO_E_BAD_ARG;
    const int node = numa_node_of_pci(sysfs_root_dir, pci_bdf);
    *node_out = node;
    if (n_cpus_out) *n_cpus_out = 0;
    if (node < 0) return AERO_OK;
    std::vector<int> cpus;
    if (!cpus_of_node(sysfs_root_dir, node, cpus)) return AERO_E_BAD_ARG;       // the node exists but its cpulist is unreadable / malformed
    if (n_cpus_out) *n_cpus_out = (uint32_t)cpus.size();
    if (cpus_out) for (size_t i = 0; i < cpus.size() && i < cpus_cap; i++) cpus_out[i] = cpus[i];
    return AERO_OK;
}
int32_t aero_numa_parse_cpulist(const char* text, int32_t* cpus_out, uint32_t cpus_cap, uint32_t* n_cpus_out) {
    if (!text || !n_cpus_out) return AERO_E_BAD_ARG;
    std::vector<int> cpus;
    if (!parse_cpulist(text, cpus)) return AERO_E_BAD_ARG;
    *n_cpus_out = (uint32_t)cpus.size();
    if (cpus_out) for (size_t i = 0; i < cpus.size() && i < cpus_cap; i++) cpus_out[i] = cpus[i];
    return AERO_OK;
}
int32_t aero_numa_device_node(int32_t device_id, int32_t* node_out) {
    if (!node_out || device_id < 0) return AERO_E_BAD_ARG;
    *node_out = numa_node_of_device(device_id);
    return AERO_OK;
}
int32_t aero_numa_bind_thread(int32_t device_id, uint32_t* n_cpus_out) {
    if (device_id < 0) return AERO_E_BAD_ARG;
    const int k = bind_thread_to_node(numa_node_of_device(device_id));
    if (n_cpus_out) *n_cpus_out = (uint32_t)k;
    return AERO_OK;
}
int32_t aero_host_alloc_near(size_t bytes, int32_t device_id, void** out) {
    if (!out || !bytes || device_id < 0) return AERO_E_BAD_ARG;
    *out = nullptr;
    const int node = numa_node_of_device(device_id);
    if (node < 0) return aero_host_alloc(bytes, out);
    hipError_t e;
    {
        // pinning faults the pages in: they are placed by THIS call, under the policy in force here (no pass over the buffer to touch them -
        // the caller is about to fill it anyway); the thread's own policy is back when the scope ends
        ScopedPreferNode prefer(node);
        e = hipHostMalloc(out, bytes, hipHostMallocNumaUser);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return aero_host_alloc(bytes, out); }
    return AERO_OK;
}

}  // extern "C"
