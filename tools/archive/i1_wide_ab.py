#!/usr/bin/env python3
"""A/B: the LDE's first pass (I1) with 64-column tiles (ZKHIP_I1_WIDE=1, A/B build) against the shipped 32-column form.
Run twice (with and without the variable): prints the pass times and a checksum of the whole LDE (must be equal)."""
import hashlib
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (tools/archive/ -> the repository root)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _ab  # noqa: E402,F401
from time_ops import Timer  # noqa: E402
from zktls_amd.device import Context  # noqa: E402

ctx = Context(0)
t = Timer(ctx.stream)
log_n, w = 20, 256
srcs = [ctx.gen_trace(1, s, log_n, w) for s in range(4)]
lde = ctx.coset_lde(srcs[0], log_n, w)
h = hashlib.sha256(lde.download().tobytes()).hexdigest()[:16]
for which, name in ((6, "I1"), (5, "F2"), (7, "fused")):
    for _ in range(200):
        ctx.ntt_pass(srcs[0], None, log_n, w, which)
    ms = []
    for s in srcs:
        best, avg = t.time(lambda: [ctx.ntt_pass(s, None, log_n, w, which) for _ in range(100)], reps=3, warm=1)
        ms.append(avg / 100)
    print("%s %s: %s ms (mean %.4f, %.3f of 8 TB/s at 8 B per element)" % ("WIDE" if os.environ.get("ZKHIP_I1_WIDE") or os.environ.get("ZKHIP_F2_WIDE") else "base", name,
          " ".join("%.4f" % x for x in ms), sum(ms) / len(ms), 8.0 * w * (1 << log_n) / (sum(ms) / len(ms)) / 1e6 / 8000))
best, avg = t.time(lambda: ctx.coset_lde(srcs[1], log_n, w, out=lde), reps=20, warm=3)
print("whole LDE %.4f ms  checksum %s" % (avg, h))
ctx.close()
