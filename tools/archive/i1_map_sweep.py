"""A/B build: the tile -> workgroup rotation (ZKHIP_NTT_MAP) of the LDE's first pass over several (trace, workspace) pairs.  python tools/i1_map_sweep.py"""
import ctypes as C
import os
import sys

os.environ.setdefault("ZKHIP_NTT_MAP", "255")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (tools/archive/ -> the repository root)
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import _ab  # noqa: F401,E402
from zktls_amd.device import Context  # noqa: E402

hip = C.CDLL("libamdhip64.so")
log_n, w, reps = 20, 256, 60
ctx = Context(0)
srcs = [ctx.gen_trace(7, k, log_n, w) for k in range(6)]
ctx.coset_lde(srcs[0], log_n, w).free()
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
st = C.c_void_p(ctx.stream)
libc = C.CDLL(None)
which = int(sys.argv[1]) if len(sys.argv) > 1 else 6      # 6 = I1 (strided in -> blocks), 2 = I1 unfused (strided -> strided), 4 = F1 (block in -> strided out)
for m in (255, 1, 2, 3, 4, 5, 6, 7, 8, 9, 0):
    libc.setenv(b"ZKHIP_NTT_MAP", str(m).encode(), 1)
    row = []
    for s in srcs:
        for _ in range(5):
            ctx.ntt_pass(s, None, log_n, w, which)
        hip.hipEventRecord(e0, st)
        for _ in range(reps):
            ctx.ntt_pass(s, None, log_n, w, which)
        hip.hipEventRecord(e1, st); hip.hipEventSynchronize(e1)
        ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), e0, e1)
        row.append(ms.value / reps)
    print("pass %d map %3d: %s  mean %.4f" % (which, m, " ".join("%.3f" % v for v in row), sum(row) / len(row)), flush=True)
