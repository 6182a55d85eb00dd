"""BASELINE configs[2]: 64 transcripts of 13 KB as keyed SHA-256 machines in ONE call -- lock-step batches (csrc/batch.h) against
one context / one stream per worker.  usage: python tools/lockstep_time.py [n=64]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from zktls_amd._lib import Params
from zktls_amd.device import lockstep_stats, prove_transcripts, set_lockstep, verify_sha256_machine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prm = Params(1, 100, 16)
base = open(os.path.join(ROOT, "tests", "golden", "reference", "guest_input0.cbor"), "rb").read()
msgs = [base + i.to_bytes(4, "little") for i in range(n)]


def timed(label, verify=False, reps=3, **kw):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        vk, res = prove_transcripts(msgs, prm, devices=[0], verify=verify, **kw)
        best = min(best, time.perf_counter() - t0)
    print("%-58s %7.1f ms = %5.2f ms per transcript, %6.0f transcripts/s" % (label, best * 1e3, best * 1e3 / n, n / best), flush=True)
    return vk, res


set_lockstep(0)
prove_transcripts(msgs[:16], prm, devices=[0], in_flight=16)
vk0, ref = timed("one stream per worker, 16 in flight", in_flight=16)
timed("... proven and verified inside the call", verify=True, in_flight=16)
for batch, lanes in ((16, 4), (8, 8), (6, 11), (4, 16), (8, 4), (11, 6)):
    set_lockstep(batch, lanes)
    prove_transcripts(msgs, prm, devices=[0])               # contexts, keys, plans
    s0 = lockstep_stats()
    vk, res = timed("lock-step, batches of %d, %d in flight" % (batch, lanes))
    s1 = lockstep_stats()
    assert vk.tolist() == vk0.tolist() and all(a[1].tobytes() == b[1].tobytes() and a[0] == b[0] for a, b in zip(ref, res)), "bytes differ"
    print("    merged launches %d for %d member requests (%.1f per launch), %d mixed rendezvous" % (
        (s1[0] - s0[0]) // 3, (s1[1] - s0[1]) // 3, (s1[1] - s0[1]) / max(1, s1[0] - s0[0]), s1[2] - s0[2]))
    timed("... proven and verified inside the call", verify=True)
t0 = time.perf_counter()
assert all(verify_sha256_machine(p, d, vk, prm, len(m)) == (0, 0) and d == hashlib.sha256(m).digest() for m, (d, p) in zip(msgs, res))
print("all %d verified against the vk on the host in %.1f ms" % (n, (time.perf_counter() - t0) * 1e3))
