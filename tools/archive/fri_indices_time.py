"""The recursion machines at the headline size (GPU box): the FRI check of one 2^20 x 256 shard proof (100 queries x 20 layers) proven in-circuit --
layers only, + commit-phase transcript, + query phase.  python tools/fri_indices_time.py [reps=5]
(under rocprofv3 --kernel-trace --stats: the per-kernel breakdown of profiles/r03_fri_indices_*)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from zktls_amd._lib import Params
from zktls_amd.device import (Context, fri_view_shard_paths, fri_view_transcript, fri_view_witness, verify_fri_indices, verify_fri_layers,
                              verify_fri_transcript)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
SEED, log_n, width, pv = 0x5A4B544C53, 20, 256, [1, 2, 3]
prm = Params(1, 100, 16)
ctx = Context(0)
trace = ctx.gen_trace(SEED, 40, log_n, width)
shard_proof = ctx.prove_shard(trace, log_n, width, pv, prm)
trace.free()
t0 = time.perf_counter()
view = fri_view_shard_paths(shard_proof, log_n, width, pv, prm)
_, _, capacity, _ = fri_view_transcript(shard_proof, log_n, width, pv, prm)
witness = fri_view_witness(shard_proof, log_n, width, pv, prm)
print("view of the shard proof (three host passes over it): %.1f ms" % ((time.perf_counter() - t0) * 1e3))


def timed(name, make_key, prove, verify):
    t0 = time.perf_counter()
    key = make_key()
    tk = time.perf_counter() - t0
    proof = prove(key)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        proof = prove(key)
    dt = (time.perf_counter() - t0) / reps
    assert verify(proof, key) == (0, 0)
    t0 = time.perf_counter()
    for _ in range(reps):
        verify(proof, key)
    tv = (time.perf_counter() - t0) / reps
    print("%-34s key %.1f ms, proof %.1f ms, %d bytes, host verification %.1f ms" % (name, tk * 1e3, dt * 1e3, proof.size, tv * 1e3), flush=True)
    key.close()


timed("layers (paths + folds)", lambda: ctx.fri_layers_key(view, prm), lambda k: ctx.prove_fri_layers(k, view, prm),
      lambda p, k: verify_fri_layers(p, view["betas"], view["final"], 100, k.root, prm))
timed("+ commit-phase transcript", lambda: ctx.fri_transcript_key(view, prm), lambda k: ctx.prove_fri_transcript(k, view, capacity, prm),
      lambda p, k: verify_fri_transcript(p, view["final"], capacity, log_n, 100, k.root, prm))
timed("+ query phase (pow, indices)", lambda: ctx.fri_indices_key(view, 16, prm), lambda k: ctx.prove_fri_indices(k, view, capacity, witness, 16, prm),
      lambda p, k: verify_fri_indices(p, view["final"], capacity, log_n, 100, 16, k.root, prm))

# ---- many shard proofs in one call (zkhip_prove_fri_indices_batch): lock-step lanes against one context per worker
from zktls_amd.device import prove_fri_indices_batch, set_lockstep

nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
shard_proofs, pvs = [], []
trace = ctx.gen_trace(SEED, 41, log_n, width)
for k in range(nb):
    pvs.append([k, 2, 3])
    shard_proofs.append(ctx.prove_shard(trace, log_n, width, pvs[-1], prm))
trace.free()
for label, batch in (("lock-step lanes", 16), ("one context per worker", 0)):
    set_lockstep(batch)
    prove_fri_indices_batch(shard_proofs, log_n, width, pvs, prm, prm, devices=[0])
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        out = prove_fri_indices_batch(shard_proofs, log_n, width, pvs, prm, prm, devices=[0])
        best = min(best, time.perf_counter() - t0)
    print("%d shard proofs in one call, %-24s %.1f ms = %.1f ms per recursion proof (views on the host included)" % (nb, label + ":", best * 1e3, best * 1e3 / nb), flush=True)
set_lockstep(16)
