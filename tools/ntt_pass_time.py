#!/usr/bin/env python3
"""Average duration of the two NTT pass launches (2^20 x 256) over many back-to-back launches,
HIP events on the context stream.  Exploration tool for A/B runs (env knobs ZKHIP_NTT_*)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import _ab  # noqa: F401  (A/B build of the library: the env knobs below exist only there)
from zktls_amd.device import Context  # noqa: E402

hip = C.CDLL("libamdhip64.so")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = Context(0)
log_n, w = 20, 256
src = ctx.fill_uniform(1, log_n, w)
dst = ctx.alloc(w << log_n)
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
st = C.c_void_p(ctx.stream)
out = []
for which in (0, 1):
    for inplace in (False, True):
        d = src if inplace else dst
        for _ in range(20):
            ctx.ntt_pass(src, d, log_n, w, which)
        hip.hipEventRecord(e0, st)
        for _ in range(reps):
            ctx.ntt_pass(src, d, log_n, w, which)
        hip.hipEventRecord(e1, st)
        hip.hipEventSynchronize(e1)
        ms = C.c_float()
        hip.hipEventElapsedTime(C.byref(ms), e0, e1)
        out.append("pass%d%s %.4f ms" % (which, "(in-place)" if inplace else "", ms.value / reps))
print(" | ".join(out))
ctx.close()
