"""exploration: cache-policy A/B of the NTT pass kernel inside ONE process (same buffers, same physical pages): needs the
-DNTT_POLICY_SWEEP build (tools/ntt_policy_sweep.sh builds it); ZKHIP_NTT_POL is re-read at every launch."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (tools/archive/ -> the repository root)
sys.path.insert(0, ROOT)
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import _ab  # noqa: F401  (A/B build of the library: the env knobs below exist only there)
from zktls_amd.device import Context
hip = C.CDLL("libamdhip64.so")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = Context(0)
log_n, w = 20, 256
src = ctx.fill_uniform(1, log_n, w)
dst = ctx.alloc(w << log_n)
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
st = C.c_void_p(ctx.stream)
def run(which, d):
    for _ in range(10): ctx.ntt_pass(src, d, log_n, w, which)
    hip.hipEventRecord(e0, st)
    for _ in range(reps): ctx.ntt_pass(src, d, log_n, w, which)
    hip.hipEventRecord(e1, st); hip.hipEventSynchronize(e1)
    ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), e0, e1)
    return ms.value / reps
pols = ["default", "0,2", "0,3", "2,2", "2,3", "3,0", "18,0", "16,0", "3,2", "18,2", "0,1", "2,1"]
for rnd in range(3):
    for p in pols:
        if p == "default": os.environ.pop("ZKHIP_NTT_POL", None)
        else: os.environ["ZKHIP_NTT_POL"] = p
        print("round %d  %-8s pass0 %.4f  pass0(in place) %.4f  pass1 %.4f  pass1(in place) %.4f" % (rnd, p, run(0, dst), run(0, src), run(1, dst), run(1, src)))
