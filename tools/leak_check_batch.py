"""exploration: device memory in use across repeated zkhip_prove_shards calls (cached internal contexts) and after releasing the cache"""
import ctypes as C, sys
sys.path.insert(0, "/root/repo")
from zktls_amd.device import Context, prove_shards
from zktls_amd._lib import Params, load
hip = C.CDLL("libamdhip64.so")
def used():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return (t.value - f.value) >> 20
ctx = Context(0)
log_n, w = 14, 64
traces = [ctx.gen_trace(1, s, log_n, w) for s in range(6)]
pvs = [[s] for s in range(6)]
base = used()
marks = []
for it in range(40):
    prove_shards(traces, log_n, w, pvs, Params(1, 20, 8), in_flight=3)
    if it in (0, 1, 10, 39): marks.append(used())
load().zkhip_release_cached_contexts()
print("MiB in use: before %d, after calls 1/2/11/40: %s, after releasing the cache %d" % (base, marks, used()))
