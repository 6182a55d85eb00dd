#!/usr/bin/env python3
"""Op-level timings on one MI355X through the C ABI (HIP events on the context stream).
Exploration tool; bench.py is the contract benchmark."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from zktls_amd.device import Context  # noqa: E402

hip = C.CDLL("libamdhip64.so")


class Timer:
    def __init__(self, stream):
        self.stream = C.c_void_p(stream)
        self.e0, self.e1 = C.c_void_p(), C.c_void_p()
        assert hip.hipEventCreate(C.byref(self.e0)) == 0
        assert hip.hipEventCreate(C.byref(self.e1)) == 0

    def time(self, fn, reps=5, warm=1):
        for _ in range(warm):
            fn()
        best, tot = 1e30, 0.0
        for _ in range(reps):
            hip.hipEventRecord(self.e0, self.stream)
            fn()
            hip.hipEventRecord(self.e1, self.stream)
            hip.hipEventSynchronize(self.e1)
            ms = C.c_float()
            hip.hipEventElapsedTime(C.byref(ms), self.e0, self.e1)
            best = min(best, ms.value)
            tot += ms.value
        return best, tot / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--width", type=int, default=256)
    args = ap.parse_args()
    log_n, w = args.log_n, args.width
    ctx = Context(0)
    t = Timer(ctx.stream)
    n = 1 << log_n
    cells = n * w
    src = ctx.fill_uniform(1, log_n, w)
    lde = ctx.alloc(2 * cells)
    tree = ctx.alloc(8 * ((2 << (log_n + 1)) - 1))
    tmp = ctx.alloc(cells)

    for which in (0, 1):
        best, avg = t.time(lambda: ctx.ntt_pass(src, tmp, log_n, w, which), reps=10, warm=2)
        print("ntt_pass which=%d  2^%d x %d: best %.3f ms avg %.3f ms  -> %.2f TB/s (8 B/elem)" % (which, log_n, w, best, avg, 8.0 * cells / best / 1e9))
    best, avg = t.time(lambda: ctx.ntt_pass(src, src, log_n, w, 0), reps=10, warm=2)
    print("ntt_pass which=0 in-place: best %.3f ms -> %.2f TB/s" % (best, 8.0 * cells / best / 1e9))
    best, avg = t.time(lambda: ctx.dft(src, log_n, w, bitrev_out=True, out=tmp), reps=5)
    print("dft fwd bitrev: best %.3f ms (%.2f G elem/s)" % (best, cells / best / 1e6))
    best, avg = t.time(lambda: ctx.coset_lde(src, log_n, w, out=lde), reps=5)
    print("coset_lde blowup 2: best %.3f ms avg %.3f ms (%.2f G cells/s; 12 B/cell min -> %.2f TB/s)" % (best, avg, cells / best / 1e6, 12.0 * cells / best / 1e9))
    best, avg = t.time(lambda: ctx.hash_rows([(lde, w)], 2 * n, out=tree), reps=3)
    perms = 2 * n * ((w + 7) // 8)
    print("hash_rows 2^%d x %d: best %.3f ms  (%.2f G perm/s, %.2f TB/s read)" % (log_n + 1, w, best, perms / best / 1e6, 8.0 * cells / best / 1e9))
    best, avg = t.time(lambda: ctx.merkle_commit([(lde, w)], log_n + 1, out=tree), reps=3)
    print("merkle_commit: best %.3f ms" % best)
    st = ctx.alloc(16 << 22)
    best, avg = t.time(lambda: ctx.poseidon2_permute(st), reps=3)
    print("poseidon2_permute 2^22 states: best %.3f ms (%.2f G perm/s)" % (best, (1 << 22) / best / 1e6))
    ctx.close()


if __name__ == "__main__":
    main()
