"""Probe: wrap a raw device pointer as a torch tensor via __cuda_array_interface__; gloo with device tensors."""
import ctypes, os, sys, torch
import torch.distributed as dist

class DevPtr:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3, "strides": None}

hip = ctypes.CDLL("libamdhip64.so")
torch.cuda.init()
x = torch.arange(64, dtype=torch.uint8, device="cuda:0")
p = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(p), 4096) == 0
t = torch.as_tensor(DevPtr(p.value, 64), device="cuda:0")
print("wrapped", t.dtype, t.shape, t.device, t.data_ptr() == p.value)
t.copy_(x)
torch.cuda.synchronize()
back = torch.empty(64, dtype=torch.uint8)
assert hip.hipMemcpy(ctypes.c_void_p(back.data_ptr()), p, 64, 2) == 0
print("roundtrip ok", bool((back == x.cpu()).all()))
t64 = t.view(torch.int64)
print("view int64", t64.shape)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("gloo", rank=0, world_size=1)
try:
    dist.all_reduce(t64); torch.cuda.synchronize(); print("gloo device all_reduce ok")
except Exception as e:
    print("gloo device all_reduce failed:", type(e).__name__, str(e)[:200])
try:
    out = torch.empty_like(t); dist.all_to_all_single(out, t); print("gloo device all_to_all ok")
except Exception as e:
    print("gloo all_to_all failed:", type(e).__name__, str(e)[:200])
dist.destroy_process_group()
os.environ["MASTER_PORT"] = "29534"
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
try:
    dist.all_reduce(t64); out = torch.empty_like(t); dist.all_to_all_single(out, t)
    g = torch.empty(64, dtype=torch.uint8, device="cuda:0"); dist.all_gather_into_tensor(g, t)
    torch.cuda.synchronize(); print("nccl world1 all_reduce/all_to_all/all_gather ok", bool((out == t).all()))
except Exception as e:
    print("nccl failed:", type(e).__name__, str(e)[:300])
dist.destroy_process_group()
