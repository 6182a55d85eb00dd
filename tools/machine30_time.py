"""A machine of 30 tables (an SP1 shard holds two to three dozen chips) at three scales (GPU box): python tools/machine30_time.py"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, oracle_lib as O, machines as M
from zktls_amd._lib import Params
from zktls_amd.device import Context, verify_machine
ctx = Context(0)
for scale in ([int(x) for x in sys.argv[1:]] or (0, 6, 10)):
    traces, progs, tables, pub = M.range_machine(6 + min(scale, 8), 8 + scale, seed=9)
    extra = [(10 + scale, 8)] * 3 + [(9 + scale, 4)] * 6 + [(8 + scale, 12)] * 7 + [(7 + scale, 4)] * 8 + [(5 + scale, 8)] * 3
    allt = [(t, p_, tb) for t, p_, tb in zip(traces, progs, tables)] + [(None, None, None, h, w, 70 + i) for i, (h, w) in enumerate(extra)]
    items = []
    for e in allt:
        if e[0] is not None:
            items.append((ctx.from_numpy(e[0]), e[0].shape[0].bit_length() - 1, e[0].shape[1], e[1], e[2]))
        else:
            items.append((ctx.gen_trace(5, e[5], e[3], e[4]), e[3], e[4], None, None))
    items.sort(key=lambda e: -e[1])
    chips = [(e[0], e[1], e[2]) for e in items]; progs = [e[3] for e in items]; tables = [e[4] for e in items]
    prm = Params(1, 100, 16)
    ctx.prove_machine(chips, progs, tables, pub, prm); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3): pf = ctx.prove_machine(chips, progs, tables, pub, prm)
    dt = (time.perf_counter() - t0) / 3
    cells = sum(w << ln for _, ln, w in chips)
    t1 = time.perf_counter(); r = verify_machine(pf, [c[1] for c in chips], [c[2] for c in chips], progs, tables, pub, prm); tv = time.perf_counter() - t1
    print("30 tables, tallest 2^%d, %.1f M cells: %.1f ms per proof (%.2f G cells/s), %d bytes, verify %s in %.1f ms" % (chips[0][1], cells / 1e6, dt * 1e3, cells / dt / 1e9, pf.size, r, tv * 1e3))
    for c in chips: c[0].free()
