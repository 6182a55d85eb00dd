"""Times zkhip_prove_sha256 on messages of 2^k blocks (GPU box): python tools/sha256_time.py"""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zktls_amd._lib import Params
from zktls_amd.device import Context, verify_sha256

ctx = Context(0)
prm = Params(1, 100, 16)
for log_blocks in ([int(x) for x in sys.argv[1:]] or (4, 8, 10, 12, 14)):
    n = (64 << log_blocks) - 9
    msg = np.random.default_rng(log_blocks).integers(0, 256, n, dtype=np.uint8).tobytes()
    ctx.prove_sha256(msg, prm)
    ctx.sync()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        digest, proof = ctx.prove_sha256(msg, prm)
    dt = (time.perf_counter() - t0) / reps
    from zktls_amd.device import sha256_air, sha256_pad
    blocks = sha256_pad(msg)
    buf, limbs = ctx.sha256_gen_trace(blocks, 1 << log_blocks)
    ctx.sync()
    t1 = time.perf_counter()
    ctx.sha256_gen_trace(blocks, 1 << log_blocks, out=buf)
    ctx.sync()
    tg = time.perf_counter() - t1
    prog = sha256_air()
    ctx.prove_shard_air(prog, buf, log_blocks + 6, 640, limbs.tolist(), prm)
    t2 = time.perf_counter()
    ctx.prove_shard_air(prog, buf, log_blocks + 6, 640, limbs.tolist(), prm)
    tp = time.perf_counter() - t2
    print("   prove_shard_air on the resident trace alone: %.1f ms" % (tp * 1e3))
    buf.free()
    assert digest == hashlib.sha256(msg).digest() and verify_sha256(proof, digest, prm, len(msg)) == (0, 0)
    print("2^%d blocks (%d bytes): rows 2^%d x 640, prove %.1f ms (trace gen alone %.1f ms), %.1f MB/s of message, %.2f G cells/s, proof %d bytes"
          % (log_blocks, n, log_blocks + 6, dt * 1e3, tg * 1e3, n / dt / 1e6, (640 << (log_blocks + 6)) / dt / 1e9, proof.size))
