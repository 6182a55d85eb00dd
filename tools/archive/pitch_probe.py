"""Does the row pitch of the caller's trace decide how fast the LDE's first (strided) pass reads it?  2^20 x 256 LDEs from sources of
pitch 256 (rows 1 KiB apart: tile rows exactly 1 MiB apart) and padded pitches; several buffers per pitch (placement).
python tools/pitch_probe.py [reps]"""
import sys
import ctypes as C
import time

sys.path.insert(0, ".")
from zktls_amd.device import Context

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = Context(0)
log_n, width = 20, 256
out = ctx.alloc(width << (log_n + 1))
for pitch in (256, 272, 264, 288, 320, 260):
    res = []
    for k in range(4):
        src = ctx.alloc(pitch << log_n)
        ctx.gen_trace(7, k, log_n, width, out=src) if pitch == width else ctx.lib.zkhip_gen_trace(ctx.handle, 7, k, log_n, width, C.c_void_p(src.ptr), pitch)
        for _ in range(3):
            ctx.coset_lde(src, log_n, width, out=out, in_ld=pitch)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.coset_lde(src, log_n, width, out=out, in_ld=pitch)
        ctx.sync()
        res.append((time.perf_counter() - t0) / reps * 1e3)
        src.free()
    print("pitch %d words: LDE %s ms" % (pitch, " ".join("%.3f" % r for r in res)), flush=True)
