"""exploration: a few back-to-back leaf hashings of a 2^21 x 256 matrix (run under rocprofv3 --pmc ... for the counters)"""
import sys
sys.path.insert(0, "/root/repo")
from zktls_amd.device import Context
ctx = Context(0)
w, log_n = 256, 21
m = ctx.fill_uniform(7, log_n, w)
dig = ctx.alloc(8 << log_n)
for _ in range(4):
    ctx.hash_rows([(m, w)], 1 << log_n, out=dig)
ctx.sync()
