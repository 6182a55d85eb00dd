"""exploration: the batch of 64 keyed transcripts split over P processes on ONE GPU (each process its own HIP runtime): is the one-process
batch bound by the host's launch path?  usage: python tools/batch64_procs.py [procs=2] [in_flight=8]"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (tools/archive/ -> the repository root)


def worker(rank, procs, inflight, barrier, out):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    from zktls_amd._lib import Params
    from zktls_amd.device import prove_transcripts
    prm = Params(1, 100, 16)
    base = open(os.path.join(ROOT, "tests", "golden", "reference", "guest_input0.cbor"), "rb").read()
    msgs = [base + i.to_bytes(4, "little") for i in range(64)][rank::procs]
    prove_transcripts(msgs[:8], prm, devices=[0], in_flight=inflight)
    for rep in range(3):
        barrier.wait()
        t0 = time.perf_counter()
        prove_transcripts(msgs, prm, devices=[0], in_flight=inflight)
        dt = time.perf_counter() - t0
        barrier.wait()
        out.put((rank, rep, dt))


if __name__ == "__main__":
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    inflight = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ctx = mp.get_context("spawn")
    barrier, out = ctx.Barrier(procs), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, procs, inflight, barrier, out)) for r in range(procs)]
    for p in ps:
        p.start()
    res = [out.get() for _ in range(3 * procs)]
    for p in ps:
        p.join()
    for rep in range(3):
        print("%d processes x %d transcripts, %d in flight each: slowest process %.1f ms" % (procs, 64 // procs, inflight, max(d for r, k, d in res if k == rep) * 1e3))
