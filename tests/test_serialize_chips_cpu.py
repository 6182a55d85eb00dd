"""ShardProof-shaped form of MULTI-CHIP proofs (csrc/serialize.cpp): the C writer against an independent Python encoder
written from the structure description in include/zkhip.h / DESIGN.md section 6b, writer and reader as exact inverses, malformed input.
One proof per multi-chip version (4, 5, 6, 9, 10, 11), made by the oracle.  Host only."""
import ctypes as C
import struct

import numpy as np
import pytest

import airs
import machines as M
from zktls_amd import _lib
from zktls_amd._lib import u8p, u32p

SEED = 0x5A4B544C53


def proofs(O):
    oprm = O.default_params(1, 5, 3)
    fib = airs.fibonacci_program()
    ft, fpub = airs.fibonacci_trace(6, 3, 5)
    cnt = airs.counter_program(8).copy()
    cnt[4] = 3
    mt, mp, mtab, mpub = M.range_machine(5, 6)
    kt, kpre, kp, ktab, kpub = M.byte_machine(6, 3)
    rt, rpre, rp, rtab, rpub = M.random_keyed_machine(303)
    return {
        4: (O.prove_chips([O.gen_trace(SEED, 1, 7, 8), O.gen_trace(SEED, 2, 6, 12), O.gen_trace(SEED, 3, 6, 4)], [1, 2], oprm), [1, 2]),
        5: (O.prove_chips([O.gen_trace_logup(SEED, 1, 7, 16, 2), O.gen_trace(SEED, 2, 6, 8)], [5], oprm, [2, 0]), [5]),
        6: (O.prove_chips([O.gen_trace_logup_cross(SEED, 0, 1, 6, 16, 8, 1), O.gen_trace_logup_cross(SEED, 1, 0, 6, 8, 16, 1), O.gen_trace(SEED, 2, 5, 4)],
                          [5], oprm, [1, 1, 0], [1, 0, -1]), [5]),
        9: (O.prove_chips_air([airs.counter_trace(7, 8, 3, 5)[0], O.gen_trace(SEED, 4, 7, 4), ft], [cnt, None, fib], fpub, oprm), fpub),
        10: (O.prove_machine(mt, mp, mtab, mpub, oprm), mpub),
        11: (O.prove_machine_keyed(kt, kpre, kp, ktab, kpub, oprm), kpub),
        111: (O.prove_machine_keyed(rt, rpre, rp, rtab, rpub, O.default_params(2, 3, 0)), rpub),
    }


def py_chips_bincode(proof_bytes, public_values):
    """independent of serialize.cpp: parses the flat multi-chip layout of DESIGN.md section 6 and emits bincode 1.x"""
    w = list(struct.unpack("<%dI" % (len(proof_bytes) // 4), proof_bytes))
    version, n, b, queries = w[1], w[2], w[3], w[4]
    per = {4: 2, 5: 3, 6: 4, 9: 3, 10: 4, 11: 5}[version]
    ent = [w[8 + per * c:8 + per * (c + 1)] for c in range(n)]
    log_ns, widths = [e[0] for e in ent], [e[1] for e in ent]
    if version in (5, 6):
        wp = [4 * (e[2] + 1) if e[2] else 0 for e in ent]
    elif version >= 10:
        wp = [4 * ((e[3] + 1) // 2 + 1) if e[3] else 0 for e in ent]
    else:
        wp = [0] * n
    pw = [e[4] for e in ent] if version == 11 else [0] * n
    n_digests = sum(e[2] for e in ent) if version >= 9 else 0
    n_digests += sum(1 for e in ent if e[3]) if version >= 10 else 0
    head = 8 + per * n + 8 * n_digests + (8 if version == 11 else 0)
    lk = any(wp)
    cross = (version == 6 and any(e[3] for e in ent)) or (version >= 10 and lk)
    h_max, L = log_ns[0] + b, log_ns[0]
    h_perm = max([log_ns[c] + b for c in range(n) if wp[c]], default=0)
    h_pre = max([log_ns[c] + b for c in range(n) if pw[c]], default=0)
    pos = 0
    out = bytearray()

    def take(k):
        nonlocal pos
        v = w[pos:pos + k]
        assert len(v) == k
        pos += k
        return v

    def u32s(vals):
        out.extend(struct.pack("<%dI" % len(vals), *vals))

    def vec(vals, elem_words):
        out.extend(struct.pack("<Q", len(vals) // elem_words))
        u32s(vals)
    vec(take(head), 1)                                                 # header envelope
    main_root = take(8)
    perm_root = take(8) if lk else None
    sums = {c: take(4) for c in range(n) if cross and wp[c]}
    quot_root = take(8)
    u32s(main_root)
    out.append(1 if lk else 0)
    if lk:
        u32s(perm_root)
    u32s(quot_root)
    out.extend(struct.pack("<Q", n))
    for c in range(n):
        for width in (pw[c], pw[c], widths[c], widths[c], wp[c], wp[c]):
            vec(take(4 * width), 4)
        out.extend(struct.pack("<Q", 2))
        vec(take(16), 4); vec(take(16), 4)
        u32s(sums.get(c, [0, 0, 0, 0]))
        out.extend(struct.pack("<Q", log_ns[c]))
    vec(take(8 * L), 8)
    final_poly, witness = take(4), take(1)[0]
    openings, steps = [], []
    for _ in range(queries):
        rounds = []
        for ws, height in (([x for x in pw if x], h_pre), (widths, h_max), ([x for x in wp if x], h_perm), ([8] * n, h_max)):
            if ws:
                rounds.append(([take(x) for x in ws], take(8 * height)))
        openings.append(rounds)
        steps.append([(take(4), take(8 * (h_max - 1 - l))) for l in range(L)])
    assert pos == len(w)
    out.extend(struct.pack("<Q", queries))
    for st in steps:
        out.extend(struct.pack("<Q", L))
        for sib, path in st:
            u32s(sib)
            vec(path, 8)
    u32s(final_poly)
    out.extend(struct.pack("<I", witness))
    out.extend(struct.pack("<Q", queries))
    for rounds in openings:
        out.extend(struct.pack("<Q", len(rounds)))
        for rows, path in rounds:
            out.extend(struct.pack("<Q", len(rows)))
            for row in rows:
                vec(row, 1)
            vec(path, 8)
    vec([int(v) for v in public_values], 1)
    return bytes(out)


def to_bincode(L, proof, pub):
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    pv = np.ascontiguousarray(np.array(pub, dtype=np.uint32))
    size = L.zkhip_chips_bincode_size(pr.ctypes.data_as(u8p), pr.size)
    out = np.zeros(max(size, 1), dtype=np.uint8)
    got = C.c_size_t(0)
    rc = L.zkhip_chips_proof_to_bincode(pr.ctypes.data_as(u8p), pr.size, pv.ctypes.data_as(u32p), pv.size, out.ctypes.data_as(u8p), size, C.byref(got))
    return rc, size, out[:got.value].tobytes()


def from_bincode(L, blob, cap, pub_cap=64):
    b = np.frombuffer(blob, dtype=np.uint8) if len(blob) else np.zeros(1, dtype=np.uint8)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    pub = np.zeros(max(pub_cap, 1), dtype=np.uint32)
    got, npub = C.c_size_t(0), C.c_size_t(0)
    rc = L.zkhip_chips_proof_from_bincode(b.ctypes.data_as(u8p), len(blob), out.ctypes.data_as(u8p), cap, C.byref(got), pub.ctypes.data_as(u32p), pub_cap, C.byref(npub))
    return rc, out[:got.value].tobytes(), pub[:npub.value].tolist()


@pytest.mark.parametrize("version", [4, 5, 6, 9, 10, 11, 111])
def test_writer_matches_python_encoder_and_round_trips(oracle, version):
    L = _lib.load()
    proof, pub = proofs(oracle)[version]
    assert np.frombuffer(proof.tobytes(), dtype=np.uint32)[1] == version % 100
    rc, size, blob = to_bincode(L, proof, pub)
    assert rc == 0 and size == len(blob) > 0
    assert blob == py_chips_bincode(proof.tobytes(), pub)
    rc, back, pub_back = from_bincode(L, blob, proof.size)
    assert rc == 0 and back == proof.tobytes() and pub_back == [int(v) for v in pub]
    # too small an output buffer, another number of public values
    assert from_bincode(L, blob, proof.size - 4)[0] != 0
    assert from_bincode(L, blob, proof.size, pub_cap=len(pub) - 1)[0] != 0
    assert to_bincode(L, proof, list(pub) + [1])[0] != 0
    # truncated and padded proofs have no such form
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    assert L.zkhip_chips_bincode_size(pr.ctypes.data_as(u8p), pr.size - 4) == 0
    # a wrong length prefix, a truncated or an extended blob, a cumulative sum on a chip without lookups: refused
    rng = np.random.default_rng(version)
    for _ in range(60):
        bad = bytearray(blob)
        kind = int(rng.integers(0, 4))
        if kind == 0:
            off = int(rng.integers(0, len(bad) - 8))
            bad[off:off + 8] = struct.pack("<Q", int(rng.integers(0, 2**40)))
        elif kind == 1:
            bad = bad[:int(rng.integers(0, len(bad)))]
        elif kind == 2:
            bad += bytes(int(rng.integers(1, 9)))
        else:
            off = int(rng.integers(0, min(len(bad), 400)))
            bad[off] ^= 1 << int(rng.integers(0, 8))
        rc, back, _ = from_bincode(L, bytes(bad), proof.size)
        assert rc != 0 or back != proof.tobytes() or bytes(bad) == blob or kind == 3     # never the original proof from altered framing


def test_single_matrix_proofs_and_noise_are_refused(oracle):
    L = _lib.load()
    O = oracle
    single = O.prove_shard(O.gen_trace(7, 1, 6, 8), [1, 2], O.default_params(1, 4, 3))
    assert L.zkhip_chips_bincode_size(single.ctypes.data_as(u8p), single.size) == 0
    rng = np.random.default_rng(1)
    for _ in range(200):
        noise = rng.integers(0, 256, int(rng.integers(0, 600)), dtype=np.uint8)
        buf = noise if noise.size else np.zeros(1, dtype=np.uint8)
        assert L.zkhip_chips_bincode_size(buf.ctypes.data_as(u8p), noise.size) == 0
        assert from_bincode(L, noise.tobytes(), 1 << 16)[0] != 0
