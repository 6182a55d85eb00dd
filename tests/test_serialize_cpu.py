"""Bincode-shaped proof form (csrc/serialize.cpp): the C writer against an independent Python encoder written
from the structure description in DESIGN.md section 6b (bincode 1.x rules: little-endian, u64 length per Vec, arrays bare),
and writer / reader as exact inverses -- on the committed golden proofs (all three proof versions).  Host only."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

from zktls_amd import _lib
from zktls_amd._lib import Params, u8p

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "oracle_kat.json")))["golden_proof_files"]


def py_bincode(proof_bytes, log_n, width, shape):
    b, queries, _pow, pairs, log_fold, log_final, hw = shape
    K, F = log_fold or 1, log_final
    H, R, arity = log_n + b, (log_n - F) // K, 1 << (log_fold or 1)
    wp = 4 * (pairs + 1) if pairs else 0
    default = b == 1 and K == 1 and F == 0 and (hw or 16) == 16
    w = list(struct.unpack("<%dI" % (len(proof_bytes) // 4), proof_bytes))
    pos = 8 if default and not pairs else (9 if default else 12)
    out = bytearray()

    def take(n):
        nonlocal pos
        v = w[pos:pos + n]
        pos += n
        return v

    def u32s(vals):
        out.extend(struct.pack("<%dI" % len(vals), *vals))

    def vec(vals, elem_words):                     # Vec<[u32; elem_words]>
        out.extend(struct.pack("<Q", len(vals) // elem_words))
        u32s(vals)
    u32s(take(8))                                  # commitments.trace
    if pairs:
        u32s(take(8))
    u32s(take(8))
    vec(take(4 * width), 4); vec(take(4 * width), 4)
    if pairs:
        vec(take(4 * wp), 4); vec(take(4 * wp), 4)
    out.extend(struct.pack("<Q", 2))
    vec(take(16), 4); vec(take(16), 4)
    vec(take(8 * R), 8)                            # commit_phase_commits
    final_poly, witness = take(4 << F), take(1)[0]
    out.extend(struct.pack("<Q", queries))
    for _ in range(queries):
        out.extend(struct.pack("<Q", 3 if pairs else 2))
        for rw in ([width] + ([wp] if pairs else []) + [8]):
            out.extend(struct.pack("<QQ", 1, rw)); u32s(take(rw))
            vec(take(8 * H), 8)
        out.extend(struct.pack("<Q", R))
        for l in range(R):
            vec(take(4 * (arity - 1)), 4)
            vec(take(8 * (H - K * (l + 1))), 8)
    vec(final_poly, 4)
    out.extend(struct.pack("<I", witness))
    out.extend(struct.pack("<Q", log_n))
    assert pos == len(w)
    return bytes(out)


def test_group_order_proofs_have_no_sp1_shaped_form():
    """version 8 (code / data groups) is RISC Zero's order; its receipts carry the seal as a flat Vec<u32>, which the proof's
    word stream already is -- the p3-uni-stark-shaped writer refuses it instead of inventing a field order"""
    g = GOLDEN["v8_groups_r0_lookup_8x16"]
    prm = Params(*g["shape"])
    assert _lib.load().zkhip_bincode_size(g["log_n"], g["width"], C.byref(prm)) == 0


@pytest.mark.parametrize("name", sorted(n for n in GOLDEN if len(GOLDEN[n]["shape"]) < 8 or not GOLDEN[n]["shape"][7]))
def test_writer_matches_python_encoder_and_round_trips(name):
    g = GOLDEN[name]
    L = _lib.load()
    proof = open(os.path.join(HERE, "golden", "proofs", name + ".bin"), "rb").read()
    prm = Params(*g["shape"])
    size = L.zkhip_bincode_size(g["log_n"], g["width"], C.byref(prm))
    assert size > 0
    src = np.frombuffer(proof, dtype=np.uint8)
    out = np.zeros(size, dtype=np.uint8)
    got = C.c_size_t(0)
    assert L.zkhip_proof_to_bincode(src.ctypes.data_as(u8p), src.size, g["log_n"], g["width"], C.byref(prm), out.ctypes.data_as(u8p), size, C.byref(got)) == 0
    assert got.value == size
    assert out.tobytes() == py_bincode(proof, g["log_n"], g["width"], g["shape"])
    back = np.zeros(len(proof), dtype=np.uint8)
    assert L.zkhip_proof_from_bincode(out.ctypes.data_as(u8p), size, g["log_n"], g["width"], C.byref(prm), len(g["public"]),
                                      back.ctypes.data_as(u8p), back.size, C.byref(got)) == 0
    assert got.value == len(proof) and back.tobytes() == proof
    # the deserialised proof still verifies (host verifier of the product)
    from zktls_amd.device import verify_shard
    assert verify_shard(back, g["log_n"], g["width"], g["public"], prm) == (0, 0)


def test_reader_rejects_malformed_input_and_small_buffers():
    g = GOLDEN["v1_6x8"]
    L = _lib.load()
    proof = np.frombuffer(open(os.path.join(HERE, "golden", "proofs", "v1_6x8.bin"), "rb").read(), dtype=np.uint8)
    prm = Params(*g["shape"])
    size = L.zkhip_bincode_size(g["log_n"], g["width"], C.byref(prm))
    out = np.zeros(size, dtype=np.uint8)
    got = C.c_size_t(0)
    assert L.zkhip_proof_to_bincode(proof.ctypes.data_as(u8p), proof.size, g["log_n"], g["width"], C.byref(prm), out.ctypes.data_as(u8p), size - 1, C.byref(got)) == -5
    assert L.zkhip_proof_to_bincode(proof.ctypes.data_as(u8p), proof.size - 4, g["log_n"], g["width"], C.byref(prm), out.ctypes.data_as(u8p), size, C.byref(got)) == -1
    assert L.zkhip_proof_to_bincode(proof.ctypes.data_as(u8p), proof.size, g["log_n"], g["width"], C.byref(prm), out.ctypes.data_as(u8p), size, C.byref(got)) == 0
    back = np.zeros(proof.size, dtype=np.uint8)
    bad = out.copy()
    bad[32 * 2] ^= 1                                   # the length prefix of trace_local
    assert L.zkhip_proof_from_bincode(bad.ctypes.data_as(u8p), size, g["log_n"], g["width"], C.byref(prm), 3, back.ctypes.data_as(u8p), back.size, C.byref(got)) == -1
    assert L.zkhip_proof_from_bincode(out.ctypes.data_as(u8p), size - 8, g["log_n"], g["width"], C.byref(prm), 3, back.ctypes.data_as(u8p), back.size, C.byref(got)) == -1
    assert L.zkhip_proof_from_bincode(out.ctypes.data_as(u8p), size, g["log_n"], g["width"], C.byref(prm), 3, back.ctypes.data_as(u8p), back.size - 1, C.byref(got)) == -5
    assert L.zkhip_bincode_size(4, 8, C.byref(prm)) == 0 and L.zkhip_bincode_size(6, 6, C.byref(prm)) == 0
