"""exploration (A/B library): the wide term kernel without its staging (ZKHIP_AIRQ_ABL=2) or without its terms (=4); run under
rocprofv3 --kernel-trace --stats: python tools/airq_ablate.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (tools/archive/ -> the repository root)
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import _ab  # noqa: F401
from zktls_amd.device import Context, sha256_air, p2chip_air
ctx = Context(0)
for name, width, prog, pub in (("sha", 608, sha256_air(), list(range(16))), ("p2", 360, p2chip_air(), list(range(9)))):
    tr = ctx.gen_trace(1, 0, 20, width)
    lde = ctx.coset_lde(tr, 20, width)
    out = ctx.quotient_values_air(prog, lde, 20, width, pub, [1, 2, 3, 4])
    for _ in range(3):
        ctx.quotient_values_air(prog, lde, 20, width, pub, [1, 2, 3, 4], out=out)
    ctx.sync()
    tr.free(); lde.free(); out.free()
