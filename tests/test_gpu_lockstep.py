"""Lock-step batches (csrc/batch.h): the provers of one batch run as fibers of a lane and their kernel launches, copies, memsets and
waits merge -- the proofs must be byte for byte those of the one-context-per-worker path (and so of the oracle)."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import sha256_air as SA
from zktls_amd import _lib
from zktls_amd._lib import Params

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ERR_BUFFER = -5                                              # include/zkhip.h


@pytest.fixture()
def lockstep():
    from zktls_amd.device import set_lockstep
    yield set_lockstep
    set_lockstep(16, 6)                                       # the library's defaults


def test_transcripts_in_lockstep_are_the_same_bytes(ctx, oracle, lockstep):
    """transcripts of three lengths (three trace heights -> three shapes, each its own batches), lock-step on and off: same vk, same
    digests, same proof bytes; the merged launches are counted; one proof against the oracle's bytes"""
    from zktls_amd.device import lockstep_stats, prove_transcripts, verify_sha256_machine
    import machines as M
    base = open(os.path.join(HERE, "golden", "reference", "guest_input0.cbor"), "rb").read()
    msgs = [base[: 3000 + 7 * i] for i in range(5)] + [base + bytes([i]) for i in range(9)] + [b"", b"abc", base[:100]] + [base[:5000 + i] for i in range(4)]
    prm, oprm = Params(1, 20, 8), oracle.default_params(1, 20, 8)
    lockstep(0)
    vk0, ref = prove_transcripts(msgs, prm, devices=[0], in_flight=4)
    for batch, lanes in ((16, 6), (3, 2), (64, 1)):
        lockstep(batch, lanes)
        s0 = lockstep_stats()
        vk, res = prove_transcripts(msgs, prm, devices=[0], in_flight=4)
        s1 = lockstep_stats()
        assert vk.tolist() == vk0.tolist()
        for m, (d0, p0), (d, p) in zip(msgs, ref, res):
            assert d == d0 == hashlib.sha256(m).digest()
            assert p.tobytes() == p0.tobytes()
        launches, requests = s1[0] - s0[0], s1[1] - s0[1]
        assert 0 < launches < requests, (launches, requests)   # something merged
        vk2, res2 = prove_transcripts(msgs, prm, devices=[0], in_flight=4, verify=True)      # checked beside the lanes
        assert all(a[1].tobytes() == b[1].tobytes() for a, b in zip(ref, res2))
    for m, (d, p) in list(zip(msgs, res))[::7]:
        assert verify_sha256_machine(p, d, vk, prm, len(m)) == (0, 0)
    tr, pre, pg, tb, pub = M.sha256_machine(msgs[14])
    assert res[14][1].tobytes() == oracle.prove_machine_keyed(tr, pre, pg, tb, pub, oprm).tobytes()


def test_unkeyed_transcripts_in_lockstep_are_prove_sha256s_bytes(ctx, lockstep):
    """zkhip_prove_transcripts_air: the batch with every job as the chip alone (version 7) -- the bytes zkhip_prove_sha256 makes on one
    context, lock-step on and off, with and without the check inside"""
    from zktls_amd.device import prove_transcripts, verify_sha256
    base = open(os.path.join(HERE, "golden", "reference", "guest_input0.cbor"), "rb").read()
    msgs = [base[: 3000 + 7 * i] for i in range(5)] + [base + bytes([i]) for i in range(9)] + [b"", b"abc", base[:100]]
    prm = Params(1, 20, 8)
    want = [ctx.prove_sha256(m, prm) for m in msgs]
    for batch, lanes, check in ((0, 0, False), (16, 6, False), (4, 2, True)):
        lockstep(batch, lanes)
        vk, res = prove_transcripts(msgs, prm, devices=[0], in_flight=4, verify=check, keyed=False)
        assert vk is None
        for m, (d0, p0), (d, p) in zip(msgs, want, res):
            assert d == d0 == hashlib.sha256(m).digest() and p.tobytes() == p0.tobytes()
    assert verify_sha256(res[3][1], res[3][0], prm, len(msgs[3])) == (0, 0)


def test_a_failing_member_does_not_hold_up_its_batch(ctx, lockstep):
    """one job of a lock-step batch has no room for its proof: it reports ZKHIP_ERR_BUFFER, the other members' proofs are made and are
    the usual bytes"""
    from zktls_amd.device import prove_transcripts
    lib = _lib.load()
    u8p = C.POINTER(C.c_uint8)
    prm = Params(1, 20, 8)
    msgs = [b"transcript %d" % i * 20 for i in range(6)]
    lockstep(0)
    vk0, ref = prove_transcripts(msgs, prm, devices=[0])
    lockstep(16, 2)
    jobs = (_lib.TranscriptJob * len(msgs))()
    keep = []
    for i, m in enumerate(msgs):
        msg = np.frombuffer(m, dtype=np.uint8)
        size = lib.zkhip_sha256_machine_proof_size(len(m), C.byref(prm))
        buf = np.empty(size, dtype=np.uint8)
        keep.append((msg, buf))
        jobs[i].message = msg.ctypes.data_as(u8p); jobs[i].message_len = len(m)
        jobs[i].proof = buf.ctypes.data_as(u8p); jobs[i].proof_cap = size if i != 2 else 16
    vk = np.zeros(8, dtype=np.uint32)
    rc = lib.zkhip_prove_transcripts((C.c_int * 1)(0), 1, jobs, len(msgs), C.byref(prm), 4, 1, vk.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert rc == ERR_BUFFER
    for i in range(len(msgs)):
        if i == 2:
            assert jobs[i].status == ERR_BUFFER and jobs[i].proof_len == 0
        else:
            assert jobs[i].status == 0
            assert keep[i][1][: jobs[i].proof_len].tobytes() == ref[i][1].tobytes()


def test_small_shards_in_lockstep(ctx, oracle, lockstep):
    """zkhip_prove_shards on sixteen small synthetic shards and zkhip_prove_shards_air_multi on eight SHA-256 chip traces: lock-step
    bytes = one-context-per-worker bytes (= the single-shard entry's, = the oracle's)"""
    from zktls_amd.device import lockstep_stats, prove_shards, prove_shards_air_multi, sha256_air, sha256_pad
    prm, oprm = Params(1, 20, 8), oracle.default_params(1, 20, 8)
    seed, log_n, width = 0x10C557E9, 9, 24
    traces = [ctx.gen_trace(seed, s, log_n, width) for s in range(16)]
    pvs = [[s, 7] for s in range(16)]
    lockstep(0)
    ref = prove_shards(traces, log_n, width, pvs, prm, device=0, in_flight=4)
    lockstep(16, 6)
    s0 = lockstep_stats()
    got = prove_shards(traces, log_n, width, pvs, prm, device=0, in_flight=4)
    s1 = lockstep_stats()
    assert 0 < s1[0] - s0[0] < s1[1] - s0[1]
    assert all(a.tobytes() == b.tobytes() for a, b in zip(ref, got))
    assert got[3].tobytes() == oracle.prove_shard(oracle.gen_trace(seed, 3, log_n, width), pvs[3], oprm).tobytes()
    # traces of one constraint program
    prog = sha256_air()
    msgs = [b"shard-batch message %d" % i * 9 for i in range(8)]
    tr, pubs = [], []
    for m in msgs:
        d, limbs = ctx.sha256_gen_trace(sha256_pad(m))
        tr.append(d); pubs.append(limbs.tolist())
    active = len(sha256_pad(msgs[0])) // 64
    ln = 6 + max(active - 1, 0).bit_length()                   # 64 rows per block, blocks rounded up to a power of two
    lockstep(0)
    ref = prove_shards_air_multi(prog, tr, ln, SA.WIDTH, pubs, prm, devices=[0])
    lockstep(4, 2)
    got = prove_shards_air_multi(prog, tr, ln, SA.WIDTH, pubs, prm, devices=[0])
    assert all(a.tobytes() == b.tobytes() for a, b in zip(ref, got))
