"""SHA-256 of a large body as a chain of shard proofs (BASELINE configs[3]): python tools/sharded_time.py [MiB=4] [in_flight=2]"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from zktls_amd._lib import Params
from zktls_amd.device import prove_sha256_sharded, verify_sha256_sharded

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 4
inflight = int(sys.argv[2]) if len(sys.argv) > 2 else 2
msg = np.random.default_rng(1).integers(0, 256, (mib << 20) - 9, dtype=np.uint8).tobytes()      # the padding fits the last block: mib full shards
prm = Params(1, 100, 16)
prove_sha256_sharded(msg[: (1 << 20) - 9], 14, prm, devices=[0], in_flight=inflight)
for rep in range(3):
    t0 = time.perf_counter()
    res = prove_sha256_sharded(msg, 14, prm, devices=[0], in_flight=inflight)
    dt = time.perf_counter() - t0
    print("%d MiB body: %d shards of 2^20 rows x 640 proven in %.1f ms = %.1f ms per MiB (%.2f G cells/s, %d in flight, host padding + chaining values included)"
          % (mib, len(res.proofs), dt * 1e3, dt * 1e3 / mib, len(res.proofs) * (640 << 20) / dt / 1e9, inflight))
assert res.digest == hashlib.sha256(msg).digest()
t0 = time.perf_counter()
assert verify_sha256_sharded(res, params=prm) == (0, 0, 0)
print("chain of %d proofs verified on the host in %.1f ms" % (len(res.proofs), (time.perf_counter() - t0) * 1e3))
