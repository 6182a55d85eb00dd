"""BASELINE configs[2] with a real statement per transcript: 64 SHA-256 chip proofs (13 KB inputs) in ONE call on one GPU.
usage: python tools/batch64_time.py [in_flight=4]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from zktls_amd._lib import Params
from zktls_amd.device import Context, prove_shards_air_multi, sha256_air, sha256_pad, verify_sha256

ctx = Context(0)
prm = Params(1, 100, 16)
base = open(os.path.join(ROOT, "tests", "golden", "reference", "guest_input0.cbor"), "rb").read()
inflight = int(sys.argv[1]) if len(sys.argv) > 1 else 4
traces, pubs, digests, lengths = [], [], [], []
t0 = time.perf_counter()
for i in range(64):
    msg = base + i.to_bytes(4, "little")
    d, limbs = ctx.sha256_gen_trace(sha256_pad(msg))
    traces.append(d); pubs.append(limbs.tolist()); digests.append(hashlib.sha256(msg).digest()); lengths.append(len(msg))
ctx.sync()
tg = time.perf_counter() - t0
prog = sha256_air()
prove_shards_air_multi(prog, traces[:8], 14, 640, pubs[:8], prm, devices=[0], in_flight=inflight)       # contexts, plans
for rep in range(3):
    t0 = time.perf_counter()
    proofs = prove_shards_air_multi(prog, traces, 14, 640, pubs, prm, devices=[0], in_flight=inflight)
    dt = time.perf_counter() - t0
    print("64 transcripts of %d bytes: traces generated in %.1f ms, proven in %.1f ms = %.2f ms per transcript (%d in flight)" % (len(base) + 4, tg * 1e3, dt * 1e3, dt * 1e3 / 64, inflight))
t0 = time.perf_counter()
assert all(verify_sha256(p, digests[i], prm, lengths[i]) == (0, 0) for i, p in enumerate(proofs))
print("all 64 verified on the host in %.1f ms" % ((time.perf_counter() - t0) * 1e3))

# the same batch as KEYED MACHINES through one call of zkhip_prove_transcripts: padding, trace generation, the range table's multiplicities
# and the proof per transcript all inside the library (host bytes in, proofs out), setup once per pooled context
from zktls_amd.device import prove_transcripts, verify_sha256_machine
msgs = [base + i.to_bytes(4, "little") for i in range(64)]
prove_transcripts(msgs[:8], prm, devices=[0], in_flight=inflight)
for rep in range(3):
    t0 = time.perf_counter()
    vk, res = prove_transcripts(msgs, prm, devices=[0], in_flight=inflight)
    dt = time.perf_counter() - t0
    print("64 transcripts as keyed SHA-256 machines (chip + range table), host bytes in: %.1f ms = %.2f ms per transcript (%d in flight)" % (dt * 1e3, dt * 1e3 / 64, inflight))
for rep in range(2):
    t0 = time.perf_counter()
    vk, res = prove_transcripts(msgs, prm, devices=[0], in_flight=inflight, verify=True)
    dt = time.perf_counter() - t0
    print("... proven AND verified inside the call (each worker checks its proof on the host while the GPU runs the others): %.1f ms = %.2f ms per transcript" % (dt * 1e3, dt * 1e3 / 64))
t0 = time.perf_counter()
assert all(verify_sha256_machine(p, d, vk, prm, len(m)) == (0, 0) and d == hashlib.sha256(m).digest() for m, (d, p) in zip(msgs, res))
print("all 64 verified against the vk on the host in %.1f ms" % ((time.perf_counter() - t0) * 1e3))
