"""exploration: does the relative placement of source and destination matter for the strided NTT pass?"""
import ctypes as C, sys
sys.path.insert(0, "/root/repo")
from zktls_amd.device import Context, DeviceBuffer
hip = C.CDLL("libamdhip64.so")
ctx = Context(0)
log_n, w = 20, 256
src = ctx.fill_uniform(1, log_n, w)
big = ctx.alloc((w << log_n) + (64 << 20) // 4)
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
st = C.c_void_p(ctx.stream)
print("src %#x big %#x" % (src.ptr, big.ptr))
for off_bytes in (0, 256, 1024, 4096, 4096 + 256, 65536, 65536 + 1024, 1 << 20, (1 << 20) + 4096, 3 << 20, (16 << 20) + 8192):
    dst = DeviceBuffer(ctx, w << log_n, ptr=big.ptr + off_bytes)
    res = []
    for which in (0, 1):
        for _ in range(20): ctx.ntt_pass(src, dst, log_n, w, which)
        hip.hipEventRecord(e0, st)
        for _ in range(200): ctx.ntt_pass(src, dst, log_n, w, which)
        hip.hipEventRecord(e1, st); hip.hipEventSynchronize(e1)
        ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), e0, e1); res.append(ms.value / 200)
    print("dst offset %9d B: pass0 %.4f ms  pass1 %.4f ms" % (off_bytes, res[0], res[1]))
