"""Probe: which HIP/HSA runtimes are mapped when torch and libaero_stark live in one process; speed of torch copies on
foreign (libaero-allocated) device pointers."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import aero_amd
from aero_amd.shard import _DevPtr
ctx = aero_amd.Context(0)
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if ("hip64" in l or "hsa-runtime" in l or "rccl" in l)})
print("\n".join(libs))
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so.7")
n = 256 << 20
p = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(p), n) == 0
q = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(q), n) == 0
t = torch.as_tensor(_DevPtr(p.value, n), device="cuda:0")
u = torch.as_tensor(_DevPtr(q.value, n), device="cuda:0")
own = torch.empty(n, dtype=torch.uint8, device="cuda:0")
for name, dst, src in (("foreign->foreign", u, t), ("foreign->torch", own, t), ("torch->foreign", t, own), ("torch->torch", own.clone(), own)):
    dst.copy_(src); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): dst.copy_(src)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {n / dt / 1e9:.1f} GB/s")
h = torch.empty(n, dtype=torch.uint8).pin_memory()
t0 = time.perf_counter(); h.copy_(t); torch.cuda.synchronize(); print(f"foreign->pinned host: {n / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
t0 = time.perf_counter(); h.copy_(own); torch.cuda.synchronize(); print(f"torch->pinned host: {n / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
