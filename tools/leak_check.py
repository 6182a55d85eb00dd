"""exploration: device memory must come back when a Context (and the buffers allocated through it) is closed"""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
from zktls_amd.device import Context
from zktls_amd._lib import Params
hip = C.CDLL("libamdhip64.so")
def free_mem():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value
base = None
for i in range(12):
    ctx = Context(0)
    tr = ctx.gen_trace(1, i, 14, 64)
    ctx.prove_shard(tr, 14, 64, [1], Params(1, 20, 8))
    ctx.prove_shard(tr, 14, 64, [1], Params(2, 20, 0, 0, 4, 6, 24))
    ctx.close()
    f = free_mem()
    if base is None: base = f
    print(i, (base - f) // 1024, "KiB below first")
