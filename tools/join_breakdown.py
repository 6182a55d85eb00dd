#!/usr/bin/env python3
"""where the join's time goes (A/B build: ZKHIP_REC_TIMING prints the phases of zkhip_prove_shard_verifier to stderr)"""
import os
import sys
import time
os.environ["ZKHIP_REC_TIMING"] = "1"
os.environ["ZKHIP_CHIPS_TIMING"] = "1"          # the phases of the machine proof inside it
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _ab  # noqa: E402,F401
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import Context, verify_shard  # noqa: E402

ctx = Context(0)
if "--sha" in sys.argv:
    # air mode: n transcript-sized SHA-256 proofs (13 221 bytes: 2^14 x 640, 100 queries) -> one proof
    from zktls_amd.device import prove_transcripts, sha256_air, sha256_padding_publics
    sys.argv.remove("--sha")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    prm = Params(1, 100, 16)
    prog = sha256_air()
    msgs = [bytes((7 * i + 3 * p + 1) & 0xff for i in range(13221)) for p in range(n)]
    made = prove_transcripts(msgs, prm, devices=[0], keyed=False)[1]
    pv = []
    for (d, _), m in zip(made, msgs):
        limbs = []
        for i in range(8):
            w = int.from_bytes(d[4 * i:4 * i + 4], "big")
            limbs += [w & 0xffff, w >> 16]
        pv.append(limbs + sha256_padding_publics(len(m)).tolist())
    inner = [p for _, p in made]
    key = ctx.shard_verifier_setup(14, 640, 100, 16, 91, prm, n_proofs=n, program=prog)
    for rep in range(3):
        t0 = time.perf_counter()
        outer = ctx.prove_shard_verifier(key, inner, 14, 640, pv, prm, prm, program=prog)
        print("compress %d SHA-256 proofs: %.1f ms, %d bytes" % (n, (time.perf_counter() - t0) * 1e3, outer.size), flush=True)
    key.close()
    ctx.close()
    sys.exit(0)
if "--keyed" in sys.argv:
    # machine mode: n KEYED transcript proofs (zkhip_prove_transcripts) -> one proof
    from zktls_amd.device import prove_transcripts, sha256_inner_machine, sha256_padding_publics
    sys.argv.remove("--keyed")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    prm = Params(1, 100, 16)
    msgs = [bytes((7 * i + 3 * p + 1) & 0xff for i in range(13221)) for p in range(n)]
    vk, made = prove_transcripts(msgs, prm, devices=[0])
    pv = []
    for (d, _), m in zip(made, msgs):
        limbs = []
        for i in range(8):
            w = int.from_bytes(d[4 * i:4 * i + 4], "big")
            limbs += [w & 0xffff, w >> 16]
        pv.append(limbs + sha256_padding_publics(len(m)).tolist())
    im = sha256_inner_machine(len(msgs[0]), vk, prm)
    key = ctx.machine_verifier_setup(im, prm, n)
    for rep in range(3):
        t0 = time.perf_counter()
        outer = ctx.prove_machine_verifier(key, im, [p for _, p in made], pv, prm)
        print("compress %d keyed transcript proofs: %.1f ms, %d bytes" % (n, (time.perf_counter() - t0) * 1e3, outer.size), flush=True)
    key.close()
    ctx.close()
    sys.exit(0)
log_n, width, q, pb = 20, 256, 100, 16
iprm, prm = Params(1, q, pb), Params(1, 100, 16)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tr = ctx.gen_trace(1, 0, log_n, width)
pv = [[1, 2, 3, 4, 5, 6, 7, 8, s] for s in range(n)]
inner = [ctx.prove_shard(tr, log_n, width, pv[s], iprm) for s in range(n)]
t0 = time.perf_counter()
verify_shard(inner[0], log_n, width, pv[0], iprm)
print("host verify of one inner proof %.2f ms" % ((time.perf_counter() - t0) * 1e3))
key = ctx.shard_verifier_setup(log_n, width, q, pb, 9, prm, n_proofs=n)
for rep in range(3):
    t0 = time.perf_counter()
    outer = ctx.prove_shard_verifier(key, inner, log_n, width, pv, iprm, prm)
    print("join %d: %.1f ms" % (n, (time.perf_counter() - t0) * 1e3), flush=True)
key.close()
ctx.close()
