"""exploration: tile-order A/B of the strided pass on slow (same class) and fast buffer pairs, one process.
Run with ZKHIP_NTT_MAP=1 ZKHIP_NTT_DEBUG=0 in the environment."""
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
def run(s, d, which=0, reps=100):
    for _ in range(5): ctx.ntt_pass(s, d, log_n, w, which)
    hip.hipEventRecord(e0, st)
    for _ in range(reps): ctx.ntt_pass(s, d, log_n, w, which)
    hip.hipEventRecord(e1, st); hip.hipEventSynchronize(e1)
    ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), e0, e1)
    return ms.value / reps
for _ in range(300): ctx.ntt_pass(bufs[0], bufs[1], log_n, w, 0)
os.environ["ZKHIP_NTT_MAP"] = "1"
pairs = {}
for i, sb in enumerate(bufs):
    for j, db in enumerate(bufs):
        if i != j: pairs[(i, j)] = run(sb, db, 0, 50)
slow = max(pairs, key=pairs.get); fast = min(pairs, key=pairs.get)
print("map 1: slowest pair %s %.4f, fastest pair %s %.4f" % (slow, pairs[slow], fast, pairs[fast]))
for m in (1, 2, 3, 4, 5, 6, 7, 0, 1):
    os.environ["ZKHIP_NTT_MAP"] = str(m)
    print("map %d: slow pair %.4f  fast pair %.4f  in place %.4f | pass1: %.4f %.4f" % (m, run(bufs[slow[0]], bufs[slow[1]]), run(bufs[fast[0]], bufs[fast[1]]), run(bufs[0], bufs[0]), run(bufs[slow[0]], bufs[slow[1]], 1), run(bufs[fast[0]], bufs[fast[1]], 1)))
