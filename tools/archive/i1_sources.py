"""The LDE's first pass (I1: strided reads of the caller's trace -> one block per tile in the context's coefficient workspace) over several
source buffers and several contexts of ONE process: is the bimodal duration (0.46 / 0.52 ms) a property of the source, of the context's
workspace, or of the process?  python tools/i1_sources.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (tools/archive/ -> the repository root)
sys.path.insert(0, ROOT)
from zktls_amd.device import Context

hip = C.CDLL("libamdhip64.so")
log_n, w, reps = 20, 256, 100
ctxs = [Context(0) for _ in range(3)]
srcs = [ctxs[0].gen_trace(7, k, log_n, w) for k in range(8)]
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
for ci, ctx in enumerate(ctxs):
    ctx.coset_lde(srcs[0], log_n, w).free()              # the context's workspaces
    st = C.c_void_p(ctx.stream)
    row = []
    for s in srcs:
        for _ in range(5):
            ctx.ntt_pass(s, None, log_n, w, 6)
        hip.hipEventRecord(e0, st)
        for _ in range(reps):
            ctx.ntt_pass(s, None, log_n, w, 6)
        hip.hipEventRecord(e1, st); hip.hipEventSynchronize(e1)
        ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), e0, e1)
        row.append(ms.value / reps)
    print("context %d: I1 over 8 sources: %s ms" % (ci, " ".join("%.3f" % v for v in row)), flush=True)
