"""Times the machine prover (GPU box): the SHA-256 chip with its unchecked limbs sent to a 2^16-row range table.
usage: python tools/machine_time.py [log_blocks ...]"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O          # program / table builders only (no oracle proving here)
import sha256_air as S
from zktls_amd._lib import Params
from zktls_amd.device import Context, sha256_air, sha256_pad, verify_machine

ctx = Context(0)
prm = Params(1, 100, 16)
V = O.air_var
sent = [S.OUT + 6, S.OUT + 7, S.OUT + 14, S.OUT + 15]
sha_tab = O.interaction_table([(O.SEND, None, 16, [c]) for c in sent])
table_prog = O.air_program(4, S.N_PUBLIC, [(O.SEL_FIRST, [(1, [V(0)])]), (O.SEL_TRANSITION, [(1, [V(0, True)]), (O.P - 1, [V(0)]), (O.P - 1, [])])])
table_tab = O.interaction_table([(O.RECEIVE, 1, 16, [0])])
prog = sha256_air()
for log_blocks in ([int(x) for x in sys.argv[1:]] or (4, 8, 12)):
    n = (64 << log_blocks) - 9
    msg = np.random.default_rng(log_blocks).integers(0, 256, n, dtype=np.uint8).tobytes()
    d_sha, limbs = ctx.sha256_gen_trace(sha256_pad(msg), 1 << log_blocks)
    d_table = ctx.range_table(d_sha, 640, 64 << log_blocks, sent, 16)     # values + multiplicities counted on the device
    lns, ws = [16, log_blocks + 6], [4, 640]
    chips, progs, tables = [(d_table, 16, 4), (d_sha, log_blocks + 6, 640)], [table_prog, prog], [table_tab, sha_tab]
    if log_blocks + 6 > 16:
        chips, progs, tables, lns, ws = chips[::-1], progs[::-1], tables[::-1], lns[::-1], ws[::-1]
    pub = limbs.tolist()
    ctx.prove_machine(chips, progs, tables, pub, prm)
    ctx.sync()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        proof = ctx.prove_machine(chips, progs, tables, pub, prm)
    dt = (time.perf_counter() - t0) / reps
    assert verify_machine(proof, lns, ws, progs, tables, pub, prm) == (0, 0)
    assert S.digest_bytes(pub) == hashlib.sha256(msg).digest()
    print("SHA-256 chip 2^%d x 640 + range table 2^16 x 4: %.1f ms per proof, %d bytes" % (log_blocks + 6, dt * 1e3, proof.size))
    d_sha.free(); d_table.free()
