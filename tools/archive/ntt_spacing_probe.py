"""exploration: strided-pass time against the address distance between source and destination (one process; the destination is
carved out of one large allocation at several offsets)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (tools/archive/ -> the repository root)
sys.path.insert(0, ROOT)
from zktls_amd.device import Context, DeviceBuffer
hip = C.CDLL("libamdhip64.so")
reps = 200
ctx = Context(0)
log_n, w = 20, 256
src = ctx.fill_uniform(1, log_n, w)
big = ctx.alloc(3 * (w << log_n))          # 3 GiB
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
print("src %#x  big %#x" % (src.ptr, big.ptr))
MiB = 1 << 20
for off in (0, 2 * MiB, 4 * MiB, 8 * MiB, 16 * MiB, 32 * MiB, 64 * MiB, 128 * MiB, 256 * MiB, 512 * MiB, 1024 * MiB, 1026 * MiB, 1100 * MiB, 1536 * MiB, 2048 * MiB):
    d = DeviceBuffer(ctx, w << log_n, ptr=big.ptr + off)
    delta = src.ptr - d.ptr
    print("dst offset %5d MiB  (src - dst = %6.1f MiB): pass0 %.4f  pass1 %.4f" % (off // MiB, delta / MiB, run(0, src, d), run(1, src, d)))
