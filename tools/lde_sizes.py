"""exploration: coset LDE (blowup 2) time per size, 256 columns: relative efficiency of the tile heights"""
import ctypes as C, sys
sys.path.insert(0, "/root/repo")
from zktls_amd.device import Context
hip = C.CDLL("libamdhip64.so")
ctx = Context(0)
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
st = C.c_void_p(ctx.stream)
w = 256
for log_n in (20, 19, 18, 17, 16, 15, 14, 12, 10):
    src = ctx.fill_uniform(1, log_n, w)
    out = ctx.alloc((w << log_n) * 2)
    for _ in range(3): ctx.coset_lde(src, log_n, w, out=out)
    reps = 20
    hip.hipEventRecord(e0, st)
    for _ in range(reps): ctx.coset_lde(src, log_n, w, out=out)
    hip.hipEventRecord(e1, st); hip.hipEventSynchronize(e1)
    ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), e0, e1)
    t = ms.value / reps
    cells = w << log_n
    print("log_n %2d: LDE %.3f ms  -> %.2f ns per 1000 input cells, %.2f TB/s at 48 B/cell" % (log_n, t, t * 1e6 / cells * 1000, 48.0 * cells / t / 1e9))
    src.free(); out.free()
