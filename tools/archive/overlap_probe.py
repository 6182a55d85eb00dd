"""exploration: do a VALU-bound kernel (Poseidon2 leaves) and a memory-bound one (coset LDE) overlap when issued from two streams?"""
import sys, time, threading
sys.path.insert(0, "/root/repo")
from zktls_amd.device import Context
w, log_n = 256, 20
c1, c2 = Context(0), Context(0)
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
