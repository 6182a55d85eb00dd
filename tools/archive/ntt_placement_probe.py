"""exploration: is the strided pass's bimodal duration (0.49 vs 0.53 ms between processes) a matter of WHERE the buffers lie?
One process, several 1 GiB sources and destinations, every pair timed."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (tools/archive/ -> the repository root)
sys.path.insert(0, ROOT)
from zktls_amd.device import Context
hip = C.CDLL("libamdhip64.so")
reps = 200
ctx = Context(0)
log_n, w = 20, 256
srcs = [ctx.fill_uniform(1 + i, log_n, w) for i in range(3)]
dsts = [ctx.alloc(w << log_n) for _ in range(5)]
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
st = C.c_void_p(ctx.stream)
def run(which, s, d):
    for _ in range(5): ctx.ntt_pass(s, d, log_n, w, which)
    hip.hipEventRecord(e0, st)
    for _ in range(reps): ctx.ntt_pass(s, d, log_n, w, which)
    hip.hipEventRecord(e1, st); hip.hipEventSynchronize(e1)
    ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), e0, e1)
    return ms.value / reps
print("src ptrs", [hex(s.ptr) for s in srcs]); print("dst ptrs", [hex(d.ptr) for d in dsts])
for i, s in enumerate(srcs):
    print("src %d -> dst k, pass0:" % i, " ".join("%.4f" % run(0, s, d) for d in dsts), "| pass1:", " ".join("%.4f" % run(1, s, d) for d in dsts))
