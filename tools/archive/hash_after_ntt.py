import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
from zktls_amd.device import Context
hip = C.CDLL("libamdhip64.so")
ctx = Context(0)
log_n, w = 20, 256
src = ctx.fill_uniform(1, log_n, w)
lde = ctx.fill_uniform(2, log_n + 1, w)
dig = ctx.alloc(8 << (log_n + 1))
ev = [C.c_void_p() for _ in range(8)]
for e in ev: hip.hipEventCreate(C.byref(e))
st = C.c_void_p(ctx.stream)
def t(a, b):
    ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), a, b); return ms.value
for trial in range(3):
    # hash alone x3
    res = []
    for k in range(3):
        hip.hipEventRecord(ev[0], st); ctx.hash_rows([(lde, w)], 2 << log_n, out=dig); hip.hipEventRecord(ev[1], st); hip.hipEventSynchronize(ev[1]); res.append(t(ev[0], ev[1]))
    # 6 ntt passes then hash
    res2 = []
    for k in range(3):
        for j in range(6): ctx.ntt_pass(src, src, log_n, w, j & 1)
        hip.hipEventRecord(ev[0], st); ctx.hash_rows([(lde, w)], 2 << log_n, out=dig); hip.hipEventRecord(ev[1], st); hip.hipEventSynchronize(ev[1]); res2.append(t(ev[0], ev[1]))
    # coset lde into lde then hash of it
    res3 = []
    for k in range(2):
        ctx.coset_lde(src, log_n, w, out=lde)
        hip.hipEventRecord(ev[0], st); ctx.hash_rows([(lde, w)], 2 << log_n, out=dig); hip.hipEventRecord(ev[1], st); hip.hipEventSynchronize(ev[1]); res3.append(t(ev[0], ev[1]))
    print("alone", ["%.2f" % x for x in res], "after ntt", ["%.2f" % x for x in res2], "after lde (its output)", ["%.2f" % x for x in res3])
    lde2 = ctx.fill_uniform(3, log_n + 1, w, out=lde)
