#!/usr/bin/env python3
"""64 transcript-sized SHA-256 proofs (13 221 bytes: 2^14 x 640, 100 queries, zkhip_prove_transcripts_air) -> ONE proof
(zkhip_prove_shard_verifier_air), three times, on the shipped library: the command behind profiles/r05_compress64_*"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import Context, prove_transcripts, sha256_air, sha256_padding_publics, verify_shard_recursive  # noqa: E402

ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prm = Params(1, 100, 16)
prog = sha256_air()
msgs = [bytes((7 * i + 3 * p + 1) & 0xff for i in range(13221)) for p in range(n)]
key = ctx.shard_verifier_setup(14, 640, 100, 16, 91, prm, n_proofs=n, program=prog)
for rep in range(3):
    t0 = time.perf_counter()
    made = prove_transcripts(msgs, prm, devices=[0], keyed=False)[1]
    t1 = time.perf_counter()
    pv = []
    for (d, _), m in zip(made, msgs):
        limbs = []
        for i in range(8):
            w = int.from_bytes(d[4 * i:4 * i + 4], "big")
            limbs += [w & 0xffff, w >> 16]
        pv.append(limbs + sha256_padding_publics(len(m)).tolist())
    t2 = time.perf_counter()
    outer = ctx.prove_shard_verifier(key, [p for _, p in made], 14, 640, pv, prm, prm, program=prog)
    t3 = time.perf_counter()
    print("%d transcripts: inner proofs %.1f ms, compress %.1f ms, %d -> %d bytes" % (n, (t1 - t0) * 1e3, (t3 - t2) * 1e3, sum(p.size for _, p in made), outer.size), flush=True)
assert verify_shard_recursive(outer, 14, 640, 100, 16, [v for p in pv for v in p], key.root, prm, n_proofs=n, program=prog) == (0, 0)
key.close()
ctx.close()
