"""Inside a lock-step batch: tree levels of up to 16384 / members nodes on the latency-optimised (16 lanes per permutation) kernels
(shipped) against the unbatched bound of 16384 (ZKHIP_COOP_KEEP=1, A/B build).  64 transcripts of 13 KB, one call.
usage: [ZKHIP_COOP_KEEP=1] python tools/coop_ab.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (tools/archive/ -> the repository root)
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import _ab  # noqa: F401,E402
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import prove_transcripts, set_lockstep  # noqa: E402

prm = Params(1, 100, 16)
base = open(os.path.join(ROOT, "tests", "golden", "reference", "guest_input0.cbor"), "rb").read()
msgs = [base + i.to_bytes(4, "little") for i in range(64)]
for batch, lanes in ((16, 6), (16, 6)):
    set_lockstep(batch, lanes)
    prove_transcripts(msgs, prm, devices=[0])
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        vk, res = prove_transcripts(msgs, prm, devices=[0])
        best = min(best, time.perf_counter() - t0)
    import hashlib
    print("ZKHIP_COOP_KEEP=%s batches of %d, %d lanes: %.1f ms  (proof bytes sha256 %s)" % (os.environ.get("ZKHIP_COOP_KEEP", "0"), batch, lanes, best * 1e3,
          hashlib.sha256(b"".join(r[1].tobytes() for r in res)).hexdigest()[:16]), flush=True)
