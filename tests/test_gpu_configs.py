"""BASELINE.json configs at their full size, byte for byte where the oracle can follow:

  configs[1]  one 2^20 x 256 SP1-shape shard: proof bytes == the CPU oracle's (the oracle proves the same shard on all host
              cores in tens of seconds) -- closes the gap the property tests of test_gpu_fullsize.py leave for the 1024-row
              two-column NTT kernel, hash_rows_vec, quotient<4> and open_partial4 at the headline shape;
  configs[2]  a batch of 64 distinct-seed 2^20 x 256 shards through ONE zkhip_prove_shards call on one GPU (traces resident,
              generated on the device), every proof accepted by the ORACLE's verifier, three sampled ones byte-equal to the
              oracle's proofs; the same batch through zkhip_prove_shards_multi (device list) gives the same bytes;
  configs[3]  one request whose execution spans 4 shards of 2^20 rows (the "~2^22-row transcript"), through the host mirror
              of the reference's ZkProver (sp1.rs:102-133): 4 verified proofs bound to the request, one byte-equal to the oracle's;
  configs[4]  RISC Zero continuations at 2^20 cycles per segment (prover.rs:88-93): one 2^20 x 128 segment in RISC Zero's shape
              with proof bytes == the oracle's; four segments in one zkhip_prove_shards call; the same four through the `-p r0`
              host mirror -- all accepted by the oracle's verifier, one byte-equal each.
"""
import ctypes as C
import os

import numpy as np
import pytest

from zktls_amd import _lib
from zktls_amd._lib import Params
from zktls_amd.device import prove_shards, prove_shards_multi, shard_device, verify_shard

pytestmark = pytest.mark.gpu
SEED = 0x5A4B544C53
LOG_N, WIDTH = 20, 256
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def big_oracle(oracle):
    """the oracle with every host core it can use (a full-size proof is ~10^10 field operations)"""
    prev = min(8, os.cpu_count() or 1)
    oracle.set_threads(min(os.cpu_count() or 1, 96))
    yield oracle
    oracle.set_threads(prev)


def test_config1_fullsize_proof_bytes_equal_the_oracles(ctx, oracle_shard_proof):
    """(the shard is shard 31 of configs[2]'s batch below: ONE oracle proof of it serves both tests -- conftest.oracle_shard_proof)"""
    pv = [9, 8, 7, 31]
    trace = ctx.gen_trace(SEED, 1031, LOG_N, WIDTH)
    proof = ctx.prove_shard(trace, LOG_N, WIDTH, pv, Params(1, 100, 16))
    trace.free()
    oproof = oracle_shard_proof(SEED, 1031, LOG_N, WIDTH, pv)
    assert proof.size == len(oproof) == 953076
    assert proof.tobytes() == oproof, "2^20 x 256 proof bytes differ from the oracle's"


def test_config2_batch_of_64_shards_in_one_call(ctx, big_oracle, oracle_shard_proof):
    O = big_oracle
    n_shards = 64
    prm, oprm = Params(1, 100, 16), O.default_params(1, 100, 16)
    traces = [ctx.gen_trace(SEED, 1000 + s, LOG_N, WIDTH) for s in range(n_shards)]       # 64 GiB resident, generated on the device
    pvs = [[9, 8, 7, s] for s in range(n_shards)]
    ctx.sync()
    proofs = prove_shards(traces, LOG_N, WIDTH, pvs, prm, device=0, in_flight=4)
    assert len(proofs) == n_shards and all(p.size == 953076 for p in proofs)
    assert len({p.tobytes() for p in proofs}) == n_shards                                   # 64 different proofs
    for s, p in enumerate(proofs):
        assert O.verify_shard(p, LOG_N, WIDTH, pvs[s], oprm) == 0, "oracle verifier rejects shard %d" % s
    assert O.verify_shard(proofs[5], LOG_N, WIDTH, pvs[6], oprm) != 0                       # bound to its own public values
    for s in (31, 63):                                                                      # (31: the oracle proof config1 made; 63: one of its own)
        assert proofs[s].tobytes() == oracle_shard_proof(SEED, 1000 + s, LOG_N, WIDTH, pvs[s]), "shard %d differs from the oracle's proof" % s
    # the device-list entry: same shards, same bytes (one visible device here: every shard lands on it; with more devices the
    # traces would have to live where zkhip_shard_device puts the shard, see test_multi_device_* below)
    sub = list(range(0, n_shards, 8))
    again = prove_shards_multi([traces[s] for s in sub], LOG_N, WIDTH, [pvs[s] for s in sub], prm, devices=[0], in_flight=4)
    assert [a.tobytes() for a in again] == [proofs[s].tobytes() for s in sub]
    for t in traces:
        t.free()
    _lib.load().zkhip_release_cached_contexts()


def test_config2_host_traces_all_visible_devices(ctx, oracle):
    """zkhip_prove_shards_multi(NULL, 0, ...): every visible device, host traces staged by the library where the shard is dealt"""
    O = oracle
    log_n, width, n_shards = 12, 32, 11
    prm, oprm = Params(1, 30, 8), O.default_params(1, 30, 8)
    host = [O.gen_trace(SEED, 40 + s, log_n, width) for s in range(n_shards)]
    pvs = [[s, 5] for s in range(n_shards)]
    proofs = prove_shards_multi(host, log_n, width, pvs, prm, devices=None, in_flight=3, host=True)
    for s in range(n_shards):
        assert proofs[s].tobytes() == O.prove_shard(host[s], pvs[s], oprm).tobytes()
    n_dev = _lib.device_count()
    assert [shard_device(s, None, n_dev) for s in range(n_dev)] == list(range(n_dev))
    _lib.load().zkhip_release_cached_contexts()


@pytest.mark.skipif(_lib.device_count() < 2, reason="needs two GPUs in one process")
def test_multi_device_device_traces_live_where_the_shard_is_dealt(oracle):
    from zktls_amd.device import Context
    O = oracle
    log_n, width, n_shards = 14, 64, 6
    devs = [0, 1]
    ctxs = {d: Context(d) for d in devs}
    prm, oprm = Params(1, 30, 8), O.default_params(1, 30, 8)
    traces = [ctxs[shard_device(s, devs)].gen_trace(SEED, 70 + s, log_n, width) for s in range(n_shards)]
    for c in ctxs.values():
        c.sync()
    pvs = [[s] for s in range(n_shards)]
    proofs = prove_shards_multi(traces, log_n, width, pvs, prm, devices=devs, in_flight=2)
    for s in range(n_shards):
        assert proofs[s].tobytes() == O.prove_shard(O.gen_trace(SEED, 70 + s, log_n, width), pvs[s], oprm).tobytes()
    for c in ctxs.values():
        c.close()
    _lib.load().zkhip_release_cached_contexts()


class Plan(C.Structure):
    _fields_ = [("log_n", C.c_int32), ("width", C.c_uint32), ("shards", C.c_uint32), ("num_queries", C.c_int32), ("pow_bits", C.c_int32)]


def test_config3_one_request_of_four_2_20_row_shards_through_the_host_mirror(big_oracle):
    O = big_oracle
    L = C.CDLL(os.path.join(ROOT, "zktls_amd", "libzktls_guest_prover.so"))
    u8pp, szp = C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)
    L.zktls_guest_prove.argtypes = [C.c_int, C.c_int, C.POINTER(Plan), C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                    u8pp, szp, u8pp, szp, C.c_char_p, C.c_size_t]
    L.zktls_unpack_batch.argtypes = [C.c_char_p, C.c_size_t, szp, szp, C.c_int]
    L.zktls_batch_flags.argtypes = [C.c_char_p, C.c_size_t]
    L.zktls_free.argtypes = [C.c_void_p]
    cbor = open(os.path.join(ROOT, "tests", "golden", "reference", "guest_input0.cbor"), "rb").read()     # the recorded transcript
    elf = b"\x7fELFguest" + bytes(range(200))
    plan = Plan(LOG_N, WIDTH, 4, 100, 16)
    out, outn, pr, prn = C.POINTER(C.c_uint8)(), C.c_size_t(), C.POINTER(C.c_uint8)(), C.c_size_t()
    err = C.create_string_buffer(512)
    rc = L.zktls_guest_prove(0, 2, C.byref(plan), cbor, len(cbor), elf, len(elf), C.byref(out), C.byref(outn), C.byref(pr), C.byref(prn), err, 512)
    assert rc == 0, err.value
    output = C.string_at(out, outn.value)
    blob = C.string_at(pr, prn.value)
    L.zktls_free(out)
    L.zktls_free(pr)
    assert len(blob) > 4 and L.zktls_batch_flags(blob, len(blob)) == 1          # a real blob, flagged synthetic
    offs, lens = (C.c_size_t * 8)(), (C.c_size_t * 8)()
    assert L.zktls_unpack_batch(blob, len(blob), offs, lens, 8) == 4
    digest = np.frombuffer(output, dtype=np.uint32).tolist()
    d = (C.c_uint32 * 8)()
    assert _lib.load().zkhip_request_digest(cbor, len(cbor), elf, len(elf), d) == 0 and list(d) == digest
    prm, oprm = Params(1, 100, 16), O.default_params(1, 100, 16)
    proofs = [np.frombuffer(blob[offs[s]:offs[s] + lens[s]], dtype=np.uint8) for s in range(4)]
    for s in range(4):
        assert proofs[s].size == 953076                                          # public values are observed, not stored
        assert verify_shard(proofs[s], LOG_N, WIDTH, digest + [s], prm) == (0, 0)
        assert O.verify_shard(proofs[s], LOG_N, WIDTH, digest + [s], oprm) == 0
    assert verify_shard(proofs[2], LOG_N, WIDTH, digest + [1], prm)[0] == -6
    seed = 0
    for i in range(4):
        seed = ((seed << 16) ^ digest[i]) & 0xFFFFFFFFFFFFFFFF
    op = O.prove_shard(O.gen_trace(seed, 3, LOG_N, WIDTH), digest + [3], oprm)
    assert proofs[3].tobytes() == op.tobytes()
    L.zktls_release_cached()


# ------------------------------------------------------------------ configs[4]: RISC Zero continuations (prover.rs:88-93)
SEG_WIDTH = 128          # blowup 4: a 2^20 x 128 segment has the 2^22 x 128 LDE (2 GiB) of the headline shard


def test_config4_segment_proof_bytes_equal_the_oracles(ctx, oracle_shard_proof):
    """one 2^20-cycle segment in RISC Zero's shape (blowup 4, fold by 16, 256 final coefficients, 50 queries, Poseidon2 width 24):
    proof bytes == the CPU oracle's.  The LDE of this shape takes the fused middle launch twice (four cosets).
    (The segment is segment 2 of the four-segment call below: one oracle proof serves both tests.)"""
    from zktls_amd._lib import segment_params
    pv = [3, 1, 4, 2]
    trace = ctx.gen_trace(SEED, 2002, LOG_N, SEG_WIDTH)
    proof = ctx.prove_shard(trace, LOG_N, SEG_WIDTH, pv, segment_params())
    trace.free()
    oproof = oracle_shard_proof(SEED, 2002, LOG_N, SEG_WIDTH, pv, "r0")
    assert proof.size == len(oproof)
    assert proof.tobytes() == oproof, "2^20 x 128 segment proof bytes differ from the oracle's"


def test_config4_four_segments_in_one_call(ctx, big_oracle, oracle_shard_proof):
    """a multi-segment proof at 2^20 cycles per segment: four segments through ONE zkhip_prove_shards call with the segment
    parameters, every proof accepted by the oracle's verifier, one byte-equal to the oracle's proof"""
    from zktls_amd._lib import segment_params
    O = big_oracle
    n_seg = 4
    prm, oprm = segment_params(), O.segment_params()
    traces = [ctx.gen_trace(SEED, 2000 + s, LOG_N, SEG_WIDTH) for s in range(n_seg)]
    pvs = [[3, 1, 4, s] for s in range(n_seg)]
    ctx.sync()
    proofs = prove_shards(traces, LOG_N, SEG_WIDTH, pvs, prm, device=0, in_flight=4)
    assert len({p.tobytes() for p in proofs}) == n_seg
    for s, p in enumerate(proofs):
        assert verify_shard(p, LOG_N, SEG_WIDTH, pvs[s], prm) == (0, 0)
        assert O.verify_shard(p, LOG_N, SEG_WIDTH, pvs[s], oprm) == 0, "oracle verifier rejects segment %d" % s
    assert O.verify_shard(proofs[1], LOG_N, SEG_WIDTH, pvs[2], oprm) != 0
    assert proofs[2].tobytes() == oracle_shard_proof(SEED, 2002, LOG_N, SEG_WIDTH, pvs[2], "r0")
    for t in traces:
        t.free()
    _lib.load().zkhip_release_cached_contexts()


def test_config4_four_segments_through_the_r0_host_mirror(big_oracle):
    """the same through the `-p r0` twin of the reference's ZkProver (prover.rs:9-106): Risc0HipGuestProver proves the four
    segments of one request; every proof verifies in the RISC Zero shape, bound to the request, one byte-equal to the oracle's"""
    O = big_oracle
    L = C.CDLL(os.path.join(ROOT, "zktls_amd", "libzktls_guest_prover.so"))
    u8pp, szp = C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)
    L.zktls_guest_prove_r0.argtypes = [C.c_int, C.c_int, C.POINTER(Plan), C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                       u8pp, szp, u8pp, szp, C.c_char_p, C.c_size_t]
    L.zktls_unpack_batch.argtypes = [C.c_char_p, C.c_size_t, szp, szp, C.c_int]
    L.zktls_free.argtypes = [C.c_void_p]
    cbor = open(os.path.join(ROOT, "tests", "golden", "reference", "guest_input0.cbor"), "rb").read()
    elf = b"\x7fELFr0guest" + bytes(range(100))
    plan = Plan(LOG_N, SEG_WIDTH, 4, 100, 16)          # the defaults (100, 16) select RISC Zero's own 50 queries, no PoW
    out, outn, pr, prn = C.POINTER(C.c_uint8)(), C.c_size_t(), C.POINTER(C.c_uint8)(), C.c_size_t()
    err = C.create_string_buffer(512)
    rc = L.zktls_guest_prove_r0(0, 2, C.byref(plan), cbor, len(cbor), elf, len(elf), C.byref(out), C.byref(outn), C.byref(pr), C.byref(prn), err, 512)
    assert rc == 0, err.value
    output, blob = C.string_at(out, outn.value), C.string_at(pr, prn.value)
    L.zktls_free(out)
    L.zktls_free(pr)
    offs, lens = (C.c_size_t * 8)(), (C.c_size_t * 8)()
    assert L.zktls_unpack_batch(blob, len(blob), offs, lens, 8) == 4
    digest = np.frombuffer(output, dtype=np.uint32).tolist()
    prm, oprm = Params(2, 50, 0, 0, 4, 8, 24), O.segment_params()
    proofs = [np.frombuffer(blob[offs[s]:offs[s] + lens[s]], dtype=np.uint8) for s in range(4)]
    for s in range(4):
        assert verify_shard(proofs[s], LOG_N, SEG_WIDTH, digest + [s], prm) == (0, 0)
        assert O.verify_shard(proofs[s], LOG_N, SEG_WIDTH, digest + [s], oprm) == 0
    assert verify_shard(proofs[0], LOG_N, SEG_WIDTH, digest + [0], Params(1, 100, 16))[0] == -6      # not an SP1-shape proof
    seed = 0
    for i in range(4):
        seed = ((seed << 16) ^ digest[i]) & 0xFFFFFFFFFFFFFFFF
    op = O.prove_shard(O.gen_trace(seed, 1, LOG_N, SEG_WIDTH), digest + [1], oprm)
    assert proofs[1].tobytes() == op.tobytes()
    L.zktls_release_cached()
