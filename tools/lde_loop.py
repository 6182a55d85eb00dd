"""exploration: 40 back-to-back coset LDEs of 2^20 x 256 (run under rocprofv3 --kernel-trace to see each pass in steady state)"""
import sys
sys.path.insert(0, "/root/repo")
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import _ab  # noqa: F401  (A/B build of the library: the env knobs below exist only there)
from zktls_amd.device import Context
ctx = Context(0)
w, log_n = 256, 20
src = ctx.fill_uniform(1, log_n, w)
out = ctx.alloc((w << log_n) * 2)
for _ in range(40):
    ctx.coset_lde(src, log_n, w, out=out)
ctx.sync()
