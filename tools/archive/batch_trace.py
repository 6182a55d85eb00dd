"""exploration: 64 keyed transcripts in one call under rocprofv3 --kernel-trace; prints how much the kernels of different proofs overlap"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (tools/archive/ -> the repository root)
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from zktls_amd._lib import Params
from zktls_amd.device import prove_transcripts
prm = Params(1, 100, 16)
base = open(os.path.join(ROOT, "tests", "golden", "reference", "guest_input0.cbor"), "rb").read()
msgs = [base + i.to_bytes(4, "little") for i in range(64)]
inflight = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prove_transcripts(msgs[:16], prm, devices=[0], in_flight=inflight)
import time
t0 = time.perf_counter()
prove_transcripts(msgs, prm, devices=[0], in_flight=inflight)
print("batch %.1f ms" % ((time.perf_counter() - t0) * 1e3))
