"""exploration: as overlap_probe.py, but the two contexts sit on streams made with hipExtStreamCreateWithCUMask and a FULL mask
(guaranteed separate hardware queues, no partition).
The "reserved room" variant quoted in DESIGN.md section 7 additionally launched the leaf kernel with 1024 threads and 81 KiB of
dynamic LDS per workgroup through a temporary launch knob that was not kept."""
import ctypes as C, sys, time, threading
sys.path.insert(0, "/root/repo")
from zktls_amd.device import Context
hip = C.CDLL("libamdhip64.so")
hip.hipSetDevice(0)
def full_stream():
    words = (C.c_uint32 * 8)(*([0xFFFFFFFF] * 8))
    s = C.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words) == 0
    return s.value
w, log_n = 256, 20
c1, c2 = Context(0, stream=full_stream()), Context(0, stream=full_stream())
src = c1.fill_uniform(1, log_n, w); out = c1.alloc((w << log_n) * 2)
m = c2.fill_uniform(7, log_n + 1, w); dig = c2.alloc(8 << (log_n + 1))
def lde(n):
    for _ in range(n): c1.coset_lde(src, log_n, w, out=out)
    c1.sync()
def hsh(n):
    for _ in range(n): c2.hash_rows([(m, w)], 1 << (log_n + 1), out=dig)
    c2.sync()
lde(3); hsh(3)
N = 20
t = time.time(); lde(N); t_l = time.time() - t
t = time.time(); hsh(N); t_h = time.time() - t
a, b = threading.Thread(target=lde, args=(N,)), threading.Thread(target=hsh, args=(N,))
t = time.time(); a.start(); b.start(); a.join(); b.join(); t_both = time.time() - t
print("LDE x%d %.1f ms | hash x%d %.1f ms | sum %.1f ms | both streams at once %.1f ms" % (N, t_l * 1e3, N, t_h * 1e3, (t_l + t_h) * 1e3, t_both * 1e3))
