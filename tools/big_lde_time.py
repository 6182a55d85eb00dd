"""The trace LDE above 2^20 rows: 2^21 x 256 and 2^22 x 128, blowup 2, HIP events around back-to-back LDEs.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel averages (every launch of the run is a full-width one)."""
import ctypes as C, sys
sys.path.insert(0, "/root/repo")
from zktls_amd.device import Context
hip = C.CDLL("libamdhip64.so")
ctx = Context(0)
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
st = C.c_void_p(ctx.stream)
shapes = [(21, 256), (22, 128)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for log_n, w in shapes:
    src = ctx.fill_uniform(1, log_n, w)
    out = ctx.alloc((w << log_n) * 2)
    for _ in range(3): ctx.coset_lde(src, log_n, w, out=out)
    reps = 12
    hip.hipEventRecord(e0, st)
    for _ in range(reps): ctx.coset_lde(src, log_n, w, out=out)
    hip.hipEventRecord(e1, st); hip.hipEventSynchronize(e1)
    ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), e0, e1)
    t = ms.value / reps
    cells = w << log_n
    bpc = 60.0 if log_n > 20 else 36.0
    print("2^%d x %d: LDE %.3f ms = %.0f GB/s at %.0f B per trace cell = %.3f of 8 TB/s" % (log_n, w, t, bpc * cells / t / 1e6, bpc, bpc * cells / t / 1e6 / 8000.0), flush=True)
    src.free(); out.free()
