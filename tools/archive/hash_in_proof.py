"""exploration: leaf-hash kernel duration inside a proof vs alone (run under rocprofv3 --kernel-trace)"""
import sys
sys.path.insert(0, "/root/repo")
from zktls_amd.device import Context
from zktls_amd._lib import Params
ctx = Context(0)
log_n, w = 20, 256
tr = ctx.gen_trace(1, 0, log_n, w)
for _ in range(3):
    ctx.prove_shard(tr, log_n, w, [1], Params(1, 100, 16))
lde = ctx.fill_uniform(2, log_n + 1, w)
dig = ctx.alloc(8 << (log_n + 1))
for _ in range(3):
    ctx.hash_rows([(lde, w)], 2 << log_n, out=dig)
ctx.sync()
