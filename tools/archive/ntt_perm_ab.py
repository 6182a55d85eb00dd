"""exploration: tile-index bit permutations for the strided pass (which address bits are 'frozen' among the tiles resident at one time),
on a same-class and a cross-class buffer pair.  Run with ZKHIP_NTT_MAP=1 ZKHIP_NTT_PERM=0 ZKHIP_NTT_DEBUG=0 in the environment."""
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
def perm_hex(dst_of_bit):            # dst_of_bit[b] = destination position of source bit b
    v = 0
    for b, d in enumerate(dst_of_bit): v |= d << (4 * b)
    return "%x" % v
for _ in range(300): ctx.ntt_pass(bufs[0], bufs[1], log_n, w, 0)
os.environ["ZKHIP_NTT_MAP"] = "4"; os.environ["ZKHIP_NTT_PERM"] = "0"
pairs = {(i, j): run(bufs[i], bufs[j], 0, 40) for i in range(6) for j in range(6) if i != j}
slow = max(pairs, key=pairs.get); fast = min(pairs, key=pairs.get)
print("rotl 4: slowest pair %s %.4f, fastest pair %s %.4f" % (slow, pairs[slow], fast, pairs[fast]))
os.environ["ZKHIP_NTT_MAP"] = "1"
cands = {
    "identity": list(range(10)),
    "rotl 4": [(b + 4) % 10 for b in range(10)],
    "rotl 5": [(b + 5) % 10 for b in range(10)],
    "rotl 3": [(b + 3) % 10 for b in range(10)],
    "bit reversal": [9 - b for b in range(10)],
    "low3->high3, next3->low3": [7, 8, 9, 0, 1, 2, 3, 4, 5, 6],
    "even/odd interleave": [0, 2, 4, 6, 8, 1, 3, 5, 7, 9],
    "odd/even interleave": [1, 3, 5, 7, 9, 0, 2, 4, 6, 8],
    "3 low + 3 high": [0, 1, 2, 7, 8, 9, 3, 4, 5, 6],
    "2 low + 4 high": [0, 1, 6, 7, 8, 9, 2, 3, 4, 5],
    "4 low + 2 high": [0, 1, 2, 3, 8, 9, 4, 5, 6, 7],
    "1 low + 5 high": [0, 5, 6, 7, 8, 9, 1, 2, 3, 4],
    "rotl 4 then swap 4<->9": [9, 5, 6, 7, 8, 4, 0, 1, 2, 3],
}
for name, pm in cands.items():
    os.environ["ZKHIP_NTT_PERM"] = perm_hex(pm) if name != "identity" else "0"
    print("%-28s same-class %.4f  cross-class %.4f  in place %.4f" % (name, run(bufs[slow[0]], bufs[slow[1]]), run(bufs[fast[0]], bufs[fast[1]]), run(bufs[0], bufs[0])))
