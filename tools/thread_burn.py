"""which threads of a proving process burn CPU: python tools/thread_burn.py [block=0]
proves 2^20 x 256 shards, four in flight, for ~3 s and prints per-thread CPU time (from /proc/self/task)"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from zktls_amd import _lib
from zktls_amd._lib import Params
from zktls_amd.device import Context

block = int(sys.argv[1]) if len(sys.argv) > 1 else 0
_lib.load().zkhip_set_wait_mode(block)
prm = Params(1, 100, 16)
ctxs = [Context(0) for _ in range(4)]
traces = [c.gen_trace(7, i, 20, 256) for i, c in enumerate(ctxs)]
for c, t in zip(ctxs, traces): c.prove_shard(t, 20, 256, [1], prm)
stop = time.time() + 3.0
count = [0] * 4
def work(i):
    while time.time() < stop:
        ctxs[i].prove_shard(traces[i], 20, 256, [i], prm); count[i] += 1
def snap():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % tid).read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[tid] = (comm, (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
        except OSError:
            pass
    return out
th = [threading.Thread(target=work, args=(i,), name="prover%d" % i) for i in range(4)]
[t.start() for t in th]
time.sleep(0.3)
s0 = snap(); t0 = time.time()
time.sleep(2.2)
dt = time.time() - t0; s1 = snap()
[t.join() for t in th]
rows = sorted(((s1[k][1] - s0.get(k, (None, 0))[1], s1[k][0], k) for k in s1), reverse=True)
print("wait mode %s: %d proofs in %.2f s = %.2f ms each; threads by CPU seconds:" % ("block" if block else "poll", sum(count), dt, dt * 1e3 / max(1, sum(count))))
for cpu, comm, tid in rows[:12]: print("  %-20s tid %-8s %.2f s = %.0f %% of a core" % (comm, tid, cpu, 100 * cpu / dt))
