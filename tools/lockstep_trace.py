"""one lock-step batch under the profiler: usage python tools/lockstep_trace.py [batch=64] [lanes=1] [n=64]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
if os.environ.get("ZKHIP_AB_LIB"):        # the A/B build (tools/_ab.py): ZKHIP_HOST_MERGE=0 keeps the members' host work serial
    import _ab  # noqa: F401
from zktls_amd._lib import Params
from zktls_amd.device import lockstep_stats, prove_transcripts, set_lockstep
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
prm = Params(1, 100, 16)
base = open(os.path.join(ROOT, "tests", "golden", "reference", "guest_input0.cbor"), "rb").read()
msgs = [base + i.to_bytes(4, "little") for i in range(n)]
set_lockstep(batch, lanes)
print("host cores:", os.cpu_count(), len(os.sched_getaffinity(0)))
prove_transcripts(msgs, prm, devices=[0], in_flight=16)
prove_transcripts(msgs, prm, devices=[0], in_flight=16)
s0 = lockstep_stats()
t0 = time.perf_counter()
prove_transcripts(msgs, prm, devices=[0], in_flight=16)
dt = time.perf_counter() - t0
s1 = lockstep_stats()
d = [b - a for a, b in zip(s0, s1)]
print("batch %d lanes %d: %.1f ms; merged launches %d, requests %d, mixed %d; summed over the lanes: issuing launches %.1f ms, waiting for the stream %.1f ms, members' host code %.1f ms" % (
    batch, lanes, dt * 1e3, d[0], d[1], d[2], d[3] / 1e6, d[4] / 1e6, d[5] / 1e6))
