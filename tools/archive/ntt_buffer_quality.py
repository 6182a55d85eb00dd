"""exploration: is the slow mode of the strided pass a property of single buffers?  One process; for every buffer: strided pass with the
stores dropped (reads only, buffer as source) and with the loads dropped (writes only, buffer as destination), then all pairs in full.
Run with ZKHIP_NTT_DEBUG=0 in the environment (the library then re-reads the variable per launch)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (tools/archive/ -> the repository root)
sys.path.insert(0, ROOT)
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import _ab  # noqa: F401  (A/B build of the library: the env knobs below exist only there)
from zktls_amd.device import Context
hip = C.CDLL("libamdhip64.so")
ctx = Context(0)
log_n, w = 20, 256
bufs = [ctx.fill_uniform(1 + i, log_n, w) for i in range(6)]
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
st = C.c_void_p(ctx.stream)
def run(s, d, dbg, reps=100):
    os.environ["ZKHIP_NTT_DEBUG"] = str(dbg)
    for _ in range(5): ctx.ntt_pass(s, d, log_n, w, 0)
    hip.hipEventRecord(e0, st)
    for _ in range(reps): ctx.ntt_pass(s, d, log_n, w, 0)
    hip.hipEventRecord(e1, st); hip.hipEventSynchronize(e1)
    ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), e0, e1)
    return ms.value / reps
for _ in range(300): ctx.ntt_pass(bufs[0], bufs[1], log_n, w, 0)
print("reads only  (buffer i as source):     ", " ".join("%.4f" % run(b, bufs[(i + 1) % 6], 2) for i, b in enumerate(bufs)))
print("writes only (buffer i as destination):", " ".join("%.4f" % run(bufs[(i + 1) % 6], b, 1) for i, b in enumerate(bufs)))
print("full pass, rows = source, columns = destination:")
for i, s in enumerate(bufs):
    print("  src %d:" % i, " ".join("  --  " if i == j else "%.4f" % run(s, d, 0) for j, d in enumerate(bufs)))
