"""exploration: CU-masked streams (hipExtStreamCreateWithCUMask).  (1) how does the leaf-hash time scale with the mask (which
bits are which CUs)?  (2) with the chip split between a 'hash' stream and a 'memory' stream, do the LDE and the hashing overlap?"""
import ctypes as C, sys, time, threading
sys.path.insert(0, "/root/repo")
from zktls_amd.device import Context
hip = C.CDLL("libamdhip64.so")
hip.hipSetDevice(0)

def masked_stream(bits):
    """bits: iterable of CU indices to enable (0..255)"""
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= (1 << (b % 32))
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask rc=%d" % rc)
    return s.value

w, log_n = 256, 20
def hash_time(ctx, m, dig, n=3):
    ctx.hash_rows([(m, w)], 1 << (log_n + 1), out=dig); ctx.sync()
    t = time.time()
    for _ in range(n): ctx.hash_rows([(m, w)], 1 << (log_n + 1), out=dig)
    ctx.sync()
    return (time.time() - t) / n * 1e3
def lde_time(ctx, src, out, n=5):
    ctx.coset_lde(src, log_n, w, out=out); ctx.sync()
    t = time.time()
    for _ in range(n): ctx.coset_lde(src, log_n, w, out=out)
    ctx.sync()
    return (time.time() - t) / n * 1e3

masks = {
    "all 256": range(256),
    "bits 0..127": range(128),
    "bits 0..63": range(64),
    "even bits": range(0, 256, 2),
    "bits = 0..3 mod 8 (128)": [b for b in range(256) if b % 8 < 4],
    "bits 0..191": range(192),
    "bits 192..255": range(192, 256),
}
for name, bits in masks.items():
    try:
        st = masked_stream(list(bits))
    except Exception as e:
        print(name, "failed:", e); continue
    c = Context(0, stream=st)
    m = c.fill_uniform(7, log_n + 1, w); dig = c.alloc(8 << (log_n + 1))
    src = c.fill_uniform(1, log_n, w); out = c.alloc((w << log_n) * 2)
    print("%-28s hash %.2f ms   LDE %.2f ms" % (name, hash_time(c, m, dig), lde_time(c, src, out)))
    c.close()

# split: hash on 192 CUs, memory-bound on 64 (and other splits)
for nh in (192, 176, 160, 128):
    sh, sm = masked_stream(range(nh)), masked_stream(range(nh, 256))
    c1, c2 = Context(0, stream=sm), Context(0, stream=sh)
    src = c1.fill_uniform(1, log_n, w); out = c1.alloc((w << log_n) * 2)
    m = c2.fill_uniform(7, log_n + 1, w); dig = c2.alloc(8 << (log_n + 1))
    def lde(n):
        for _ in range(n): c1.coset_lde(src, log_n, w, out=out)
        c1.sync()
    def hsh(n):
        for _ in range(n): c2.hash_rows([(m, w)], 1 << (log_n + 1), out=dig)
        c2.sync()
    lde(2); hsh(2)
    N = 10
    t = time.time(); lde(N); t_l = time.time() - t
    t = time.time(); hsh(N); t_h = time.time() - t
    a, b = threading.Thread(target=lde, args=(N,)), threading.Thread(target=hsh, args=(N,))
    t = time.time(); a.start(); b.start(); a.join(); b.join(); t_both = time.time() - t
    print("split %d/%d: LDE alone %.1f ms | hash alone %.1f ms | both at once %.1f ms (x%d each)" % (nh, 256 - nh, t_l * 1e3, t_h * 1e3, t_both * 1e3, N))
    c1.close(); c2.close()
