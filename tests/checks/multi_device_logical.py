#!/usr/bin/env python3
"""The library's multi-device entries on TWO (or more) LOGICAL devices of a box with fewer GPUs.

Run as a fresh process (tests/test_gpu_multirank.py does, and so can a person):
    ZKHIP_LOGICAL_DEVICES=2 python tests/checks/multi_device_logical.py
It loads the A/B build (zktls_amd/libzkhip_ab.so, never shipped), whose device ordinals 0 .. K-1 are then K logical devices on the
physical ones (csrc/context.cpp): own context pools, own workers / lanes per listed device, and every zkhip_malloc allocation
remembers the logical device of its context, so that `the trace lives where the shard is dealt` is CHECKED by logical ordinal --
what two physical GPUs enforce by themselves.  Proof bytes are compared with the oracle's.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
K = int(os.environ.setdefault("ZKHIP_LOGICAL_DEVICES", "2"))
import _ab  # noqa: E402,F401  (before anything loads the library)
import oracle_lib as O  # noqa: E402
from zktls_amd import _lib  # noqa: E402
from zktls_amd._lib import Params  # noqa: E402
from zktls_amd.device import Context, prove_shards_multi, prove_transcripts, shard_device, verify_sha256_machine  # noqa: E402

SEED = 0x5A4B544C53


def main():
    O.build()
    O.set_threads(min(8, os.cpu_count() or 1))
    out = {"logical_devices": _lib.device_count()}
    assert out["logical_devices"] == K, "the A/B library does not report %d logical devices" % K
    devs = list(range(K))
    ctxs = {d: Context(d) for d in devs}
    prm, oprm = Params(1, 30, 8), O.default_params(1, 30, 8)
    # (1) device traces that live where zkhip_shard_device deals the shard; one context + stream per worker (big enough not to be "small")
    log_n, width, n_shards = 15, 544, 3 * K          # 2^15 x 544 = 17.8 M cells: above the lock-step threshold
    traces = [ctxs[shard_device(s, devs)].gen_trace(SEED, 70 + s, log_n, width) for s in range(n_shards)]
    for c in ctxs.values():
        c.sync()
    pvs = [[s] for s in range(n_shards)]
    proofs = prove_shards_multi(traces, log_n, width, pvs, prm, devices=devs, in_flight=2)
    for s in (0, n_shards - 1):
        assert proofs[s].tobytes() == O.prove_shard(O.gen_trace(SEED, 70 + s, log_n, width), pvs[s], oprm).tobytes(), "shard %d differs from the oracle's proof" % s
    assert len({p.tobytes() for p in proofs}) == n_shards
    out["device_traces_dealt_where_they_live"] = n_shards
    # (2) NULL, 0 = every visible (here: logical) device
    again = prove_shards_multi(traces, log_n, width, pvs, prm, devices=None, in_flight=2)
    assert [a.tobytes() for a in again] == [p.tobytes() for p in proofs]
    out["null_device_list_same_bytes"] = True
    # (3) a trace on the WRONG device is refused, by logical ordinal
    wrong = list(traces)
    wrong[0], wrong[1] = wrong[1], wrong[0]
    try:
        prove_shards_multi(wrong, log_n, width, pvs, prm, devices=devs, in_flight=2)
        raise AssertionError("a trace on another device than its shard's was accepted")
    except _lib.ZkHipError as e:
        assert "lives on device" in str(e), str(e)
    out["misplaced_trace_refused"] = True
    for t in traces:
        t.free()
    # (4) small shards: the lock-step dealer, lanes per device
    ln2, w2, n2 = 10, 32, 8 * K
    small = [ctxs[shard_device(s, devs)].gen_trace(SEED, 300 + s, ln2, w2) for s in range(n2)]
    for c in ctxs.values():
        c.sync()
    sp = prove_shards_multi(small, ln2, w2, [[s, 9] for s in range(n2)], prm, devices=devs, in_flight=2)
    for s in (0, 1, n2 - 1):
        assert sp[s].tobytes() == O.prove_shard(O.gen_trace(SEED, 300 + s, ln2, w2), [s, 9], oprm).tobytes()
    out["lockstep_small_shards"] = n2
    for t in small:
        t.free()
    # (5) host traces, all devices: staged by the library where the shard is dealt
    host = [O.gen_trace(SEED, 40 + s, 12, 32) for s in range(2 * K + 1)]
    hp = prove_shards_multi(host, 12, 32, [[s, 5] for s in range(len(host))], prm, devices=None, in_flight=3, host=True)
    for s in range(len(host)):
        assert hp[s].tobytes() == O.prove_shard(host[s], [s, 5], oprm).tobytes()
    out["host_traces"] = len(host)
    # (6) a batch of transcripts over the device list (every device makes its own proving key; one vk)
    import hashlib
    msgs = [b"transcript %d " % i * 40 for i in range(4 * K)]
    vk, res = prove_transcripts(msgs, Params(1, 20, 8), devices=devs)
    assert all(d == hashlib.sha256(m).digest() for m, (d, _) in zip(msgs, res))
    assert all(verify_sha256_machine(p, d, vk, Params(1, 20, 8), len(m)) == (0, 0) for m, (d, p) in zip(msgs, res))
    out["transcripts_over_the_device_list"] = len(msgs)
    # (7) the compress stage over the device list: joins of one shape, join j on device j mod K (every device makes the shape's key; one vk),
    # the bytes those of one context proving them one by one
    from zktls_amd.device import prove_shard_verifier_batch, verify_shard_recursive
    jl, jw, iq = 6, 16, Params(1, 5, 2)
    jpv = [[3, 4, 100 + s] for s in range(2 * (2 * K + 1))]
    c0 = ctxs[devs[0]]
    inner = [c0.prove_shard(c0.gen_trace(SEED, 500 + s, jl, jw), jl, jw, jpv[s], iq) for s in range(len(jpv))]
    joins, jvk = prove_shard_verifier_batch(inner, 2, jl, jw, jpv, iq, Params(1, 20, 8), devices=devs, in_flight=2, verify=True)
    jkey = c0.shard_verifier_setup(jl, jw, 5, 2, 3, Params(1, 20, 8), n_proofs=2)
    assert jvk.tolist() == jkey.root.tolist() and len(joins) == 2 * K + 1
    for j in range(len(joins)):
        assert joins[j].tobytes() == c0.prove_shard_verifier(jkey, inner[2 * j:2 * j + 2], jl, jw, jpv[2 * j:2 * j + 2], iq, Params(1, 20, 8)).tobytes()
        assert verify_shard_recursive(joins[j], jl, jw, 5, 2, [v for q in jpv[2 * j:2 * j + 2] for v in q], jvk, Params(1, 20, 8), n_proofs=2) == (0, 0)
    # (8) the tree in one call: the same joins dealt over the device list, each one's tables for the top filled on its worker's thread, the top on device 0
    from zktls_amd.device import InnerMachine, prove_shard_tree, shard_verifier_describe, verify_machine_recursive
    chips = []
    for i in range(8):
        p_, ln_, mw_, pw_ = shard_verifier_describe(jl, jw, 5, 2, 3, i, 0, 2)
        t_, _, _, _ = shard_verifier_describe(jl, jw, 5, 2, 3, i, 1, 2)
        chips.append(dict(ln=ln_, W=mw_, Pw=pw_, prog=p_, tab=t_))
    im = InnerMachine(chips, jkey.root, 20, 8, 6)
    tkey = c0.machine_verifier_setup(im, Params(1, 20, 8), len(joins))
    jflat = [jpv[2 * j] + jpv[2 * j + 1] for j in range(len(joins))]
    top = c0.prove_machine_verifier(tkey, im, joins, jflat, Params(1, 20, 8))
    top1, joins1, jvk1 = prove_shard_tree(c0, tkey, im, inner, 2, jl, jw, jpv, iq, Params(1, 20, 8), Params(1, 20, 8), devices=devs, in_flight=2)
    assert top1.tobytes() == top.tobytes() and jvk1.tolist() == jvk.tolist() and [x.tobytes() for x in joins1] == [x.tobytes() for x in joins]
    assert verify_machine_recursive(im, top1, [v for q in jflat for v in q], tkey.root, Params(1, 20, 8), len(joins)) == (0, 0)
    tkey.close()
    out["tree_in_one_call_over_the_device_list"] = True
    jkey.close()
    out["joins_over_the_device_list"] = len(joins)
    for c in ctxs.values():
        c.close()
    _lib.load().zkhip_release_cached_contexts()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
