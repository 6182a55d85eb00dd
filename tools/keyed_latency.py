"""latency of ONE keyed SHA-256 machine proof of the recorded 13 KB transcript (2^14 x 640 chip + 2^16-row table), one context, nothing else in
flight: python tools/keyed_latency.py [reps=20]   (run under rocprofv3 --kernel-trace --stats for the launch count and the busy time)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from zktls_amd._lib import Params
from zktls_amd.device import Context, verify_sha256_machine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = Context(0)
prm = Params(1, 100, 16)
msg = open(os.path.join(ROOT, "tests", "golden", "reference", "guest_input0.cbor"), "rb").read()
key = ctx.sha256_setup(prm)
for _ in range(3):
    digest, proof = ctx.prove_sha256_machine(key, msg, prm)
t0 = time.perf_counter()
for _ in range(reps):
    digest, proof = ctx.prove_sha256_machine(key, msg, prm)
dt = (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
assert verify_sha256_machine(proof, digest, key.root, prm, len(msg)) == (0, 0)
tv = time.perf_counter() - t0
print("keyed SHA-256 machine, %d-byte transcript: %.2f ms per proof (one in flight), %d bytes; host verification %.2f ms" % (len(msg), dt * 1e3, proof.size, tv * 1e3))
