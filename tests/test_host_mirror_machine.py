"""The host mirror of ZkProver::prove (zktls_amd/host) for shards in SP1's shard STRUCTURE (MachinePlan: chips of mixed heights, LogUp pairs inside and across
tables, preprocessed columns committed by setup): setup -> prove -> verify (sp1.rs:113-120) with every shard ONE keyed-machine proof, and the compress stage
behind the same call in machine mode.  CPU: argument handling, the loud failure without a device, a blob made on an MI355X checked with no device.  GPU: the
shard proofs' bytes and the key against the oracle on tests/machines.py's traces, the compressed and the tree blob checked on the host."""
import ctypes as C
import os

import numpy as np
import pytest

import machines as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "zktls_amd", "libzktls_guest_prover.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "proofs")
SP1_SMALL = [(8, 24, 3, 1), (8, 32, 3, 0), (7, 16, 2, -1), (6, 32, 2, -1), (5, 8, 1, -1)]
PRE = ((3, 8),)
CBOR, ELF = b"\xa1machine", b"\x7fELFprog"
SYNTHETIC, COMPRESSED, TREE, MACHINE = 1, 16, 32, 64


class MPlan(C.Structure):
    _fields_ = [("n_chips", C.c_int32), ("log_ns", C.POINTER(C.c_int32)), ("widths", C.POINTER(C.c_uint32)), ("pairs", C.POINTER(C.c_uint32)), ("partners", C.POINTER(C.c_int32)),
                ("pre_widths", C.POINTER(C.c_uint32)), ("shards", C.c_uint32), ("num_queries", C.c_int32), ("pow_bits", C.c_int32), ("in_flight", C.c_uint32)]


def mplan(spec, pre, shards, q, pb, in_flight=0):
    n, pw = len(spec), dict(pre)
    keep = ((C.c_int32 * n)(*[c[0] for c in spec]), (C.c_uint32 * n)(*[c[1] for c in spec]), (C.c_uint32 * n)(*[c[2] for c in spec]), (C.c_int32 * n)(*[c[3] for c in spec]),
            (C.c_uint32 * n)(*[pw.get(c, 0) for c in range(n)]))
    p = MPlan(n, keep[0], keep[1], keep[2], keep[3], keep[4], shards, q, pb, in_flight)
    p._keep = keep
    return p


@pytest.fixture(scope="module")
def lib():
    L = C.CDLL(SO)
    u8pp, szp = C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)
    L.zktls_guest_prove_machine.argtypes = [C.c_int, C.c_int, C.POINTER(MPlan), C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, u8pp, szp, u8pp, szp, C.c_char_p,
                                            C.c_char_p, C.c_size_t]
    L.zktls_machine_setup.argtypes = [C.c_int, C.c_int, C.POINTER(MPlan), C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_size_t]
    L.zktls_verify_machine_blob.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(MPlan), C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_int)]
    L.zktls_machine_join_size.argtypes = [C.POINTER(MPlan)]
    L.zktls_machine_join_size.restype = C.c_uint32
    L.zktls_request_digest.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]
    L.zktls_unpack_batch.argtypes = [C.c_char_p, C.c_size_t, szp, szp, C.c_int]
    L.zktls_batch_flags.argtypes = [C.c_char_p, C.c_size_t]
    L.zktls_set_compress_join_size.argtypes = [C.c_uint32]
    L.zktls_free.argtypes = [C.c_void_p]
    return L


def prove(L, mode, plan, cbor=CBOR, elf=ELF, compress=0, setup_first=1):
    out, outn, pr, prn = C.POINTER(C.c_uint8)(), C.c_size_t(), C.POINTER(C.c_uint8)(), C.c_size_t()
    err, vk = C.create_string_buffer(512), C.create_string_buffer(64)
    rc = L.zktls_guest_prove_machine(0, mode, C.byref(plan) if plan is not None else None, compress, setup_first, cbor, len(cbor), elf, len(elf), C.byref(out), C.byref(outn),
                                     C.byref(pr), C.byref(prn), vk, err, 512)
    if rc != 0:
        return rc, err.value.decode(), None, None, None
    o, p = C.string_at(out, outn.value), C.string_at(pr, prn.value)
    L.zktls_free(out)
    L.zktls_free(pr)
    return 0, "", o, p, vk.raw


def entries(L, blob, cap=64):
    offs, lens = (C.c_size_t * cap)(), (C.c_size_t * cap)()
    n = L.zktls_unpack_batch(blob, len(blob), offs, lens, cap)
    return [blob[offs[i]:offs[i] + lens[i]] for i in range(n)], list(offs[:n]), list(lens[:n])


def digest_words(L, cbor, elf):
    d = (C.c_uint32 * 8)()
    L.zktls_request_digest(cbor, len(cbor), elf, len(elf), d)
    return [int(v) for v in d]


def stream_seed(words):
    seed = 0
    for i in range(4):
        seed = ((seed << 16) ^ words[i]) & 0xFFFFFFFFFFFFFFFF
    return seed


def check(L, blob, plan, vk, cbor=CBOR, elf=ELF):
    reason = C.c_int(0)
    return L.zktls_verify_machine_blob(blob, len(blob), C.byref(plan), cbor, len(cbor), elf, len(elf), vk, C.byref(reason))


def test_mock_mode_and_the_argument_checks_of_a_machine_plan(lib):
    plan = mplan(SP1_SMALL, PRE, 3, 3, 1)
    rc, err, out, proof, vk = prove(lib, 0, plan)
    assert rc == 0 and len(out) == 32 and proof == b""                         # mock: a placeholder of 4 bytes -> "no proof" (sp1.rs:128-130)
    assert out == bytes(np.array(digest_words(lib, CBOR, ELF), dtype=np.uint32))
    vkb = C.create_string_buffer(64)
    err = C.create_string_buffer(512)
    assert lib.zktls_machine_setup(0, 0, C.byref(plan), ELF, len(ELF), vkb, err, 512) == 0                      # mock setup: a zero commitment + the program's digest
    assert vkb.raw[:32] == bytes(32) and vkb.raw[32:] == bytes(np.array(digest_words(lib, b"", ELF), dtype=np.uint32))
    assert prove(lib, 2, None)[0] == -1
    assert prove(lib, 3, plan)[1].find("network") >= 0
    # malformed plans are refused before any device work, as values (nothing unwinds across the boundary: sp1.rs:80-100)
    for spec, pre, what in (([(7, 16, 2, -1), (8, 24, 3, -1)], (), "tallest first"), ([(8, 24, 3, 1), (8, 32, 2, 0)], (), "mutual"), ([(8, 24, 3, 0)], (), "partner"),
                            ([(8, 24, 4, -1)], (), "8 columns per LogUp pair"), ([(8, 24, 1, -1)], ((0, 12),), "whole in-table pairs"), ([(8, 24, 3, 1), (8, 32, 3, 0)], ((0, 8),), "whole in-table pairs"),
                            ([(8, 24, 3, -1), (7, 16, 1, -1)], (), "KEYED machine")):
        rc, err, _, _, _ = prove(lib, 2, mplan(spec, pre, 2, 3, 1), setup_first=0)
        assert rc == -1 and what in err, (spec, err)
    assert prove(lib, 2, mplan(SP1_SMALL, PRE, 0, 3, 1), setup_first=0)[1].find("empty") >= 0
    from zktls_amd import _lib
    if _lib.device_count() == 0:
        rc, err, _, _, _ = prove(lib, 2, plan)
        assert rc == -1 and "no CPU fallback" in err                          # no device: loud, no fallback
    # shard proofs per join: what one machine-mode join takes (64 proofs at most), bounded by zktls_set_compress_join_size
    assert lib.zktls_machine_join_size(C.byref(mplan(SP1_SMALL, PRE, 3, 3, 1))) == 3
    assert lib.zktls_machine_join_size(C.byref(mplan(SP1_SMALL, PRE, 100, 3, 1))) == 50
    lib.zktls_set_compress_join_size(2)
    try:
        assert lib.zktls_machine_join_size(C.byref(mplan(SP1_SMALL, PRE, 5, 3, 1))) == 2
    finally:
        lib.zktls_set_compress_join_size(0)
    assert check(lib, b"ZKTB" + bytes(40), plan, bytes(64)) == -1


def test_the_bench_plan_fits_one_join(lib):
    """the reference's benchmark execution (22 shards, benchmark.md:9) in SP1's shard structure: the six-chip machine of bench.py's `multichip` section; 22 such
    shard proofs fit ONE machine-mode join (the bound is 64 proofs, or a Poseidon2 chip of 2^22 rows)"""
    from zktls_amd.device import SP1_SHAPED_SPEC, SP1_SHAPED_PRE
    assert lib.zktls_machine_join_size(C.byref(mplan(SP1_SHAPED_SPEC, SP1_SHAPED_PRE, 22, 100, 16))) == 22


def fixture_paths():
    return os.path.join(GOLDEN, "machine_blob_small_x3.bin"), os.path.join(GOLDEN, "machine_tree_blob_small_x5.bin"), os.path.join(GOLDEN, "machine_blob_small.vk")


def test_a_verifier_without_a_gpu_checks_machine_blobs(lib, oracle):
    """blobs made on an MI355X (tests/golden/make_machine_fixture.py): three machine shards -> ONE machine-mode proof; five shards in joins of two -> three joins -> ONE
    proof above them.  Checked here with NO device from (plan, input, ELF, vk): the join's key and the top's are derived on the host's cores.  The vk itself -- the
    commitment to the preprocessed columns -- equals the ORACLE's commitment to the restatement's columns (tests/machines.py), so nothing in this check was made by the prover"""
    one, tree, vkp = fixture_paths()
    vk = open(vkp, "rb").read()
    assert len(vk) == 64 and vk[32:] == bytes(np.array(digest_words(lib, b"", ELF), dtype=np.uint32))
    key_seed = stream_seed(digest_words(lib, b"", ELF))
    _, pres, _, _, _ = M.sp1_shaped_machine(SP1_SMALL, seed=1, shard=0, pre=PRE, n_public=9, key_seed=key_seed)
    lns = [c[0] for c in SP1_SMALL]
    assert vk[:32] == oracle.machine_setup(pres, lns, oracle.default_params(1, 3, 1)).tobytes()
    blob = open(one, "rb").read()
    plan = mplan(SP1_SMALL, PRE, 3, 3, 1)
    assert lib.zktls_batch_flags(blob, len(blob)) == SYNTHETIC | MACHINE | COMPRESSED
    ent, offs, lens = entries(lib, blob)
    assert len(ent) == 2 and lens[1] == 36
    assert check(lib, blob, plan, vk) == 0
    assert check(lib, blob, plan, vk, cbor=CBOR + b"!") == -2                 # another request
    assert check(lib, blob, plan, vk, elf=ELF + b"!") == -1                   # a key made for another program
    assert check(lib, blob, mplan(SP1_SMALL, PRE, 4, 3, 1), vk) == -1         # another shard count
    other = bytes([vk[0] ^ 1]) + vk[1:]
    assert check(lib, blob, plan, other) == -2                                # another machine key: another join key, the proof does not open against it
    bad = bytearray(blob)
    bad[offs[0] + lens[0] // 2] ^= 1
    assert check(lib, bytes(bad), plan, vk) == -2
    tblob = open(tree, "rb").read()
    tplan = mplan(SP1_SMALL, PRE, 5, 3, 1)
    lib.zktls_set_compress_join_size(2)
    try:
        assert lib.zktls_batch_flags(tblob, len(tblob)) == SYNTHETIC | MACHINE | COMPRESSED | TREE
        ent, offs, lens = entries(lib, tblob)
        assert len(ent) == 2 and lens[1] == 36
        assert check(lib, tblob, tplan, vk) == 0
        assert check(lib, tblob, tplan, vk, cbor=CBOR + b"!") == -2
        bad = bytearray(tblob)
        bad[offs[0] + lens[0] // 2] ^= 1
        assert check(lib, bytes(bad), tplan, vk) == -2
        lib.zktls_set_compress_join_size(0)                                   # joins of another size: another statement
        assert check(lib, tblob, tplan, vk) != 0
    finally:
        lib.zktls_set_compress_join_size(0)


@pytest.mark.gpu
def test_machine_shards_through_the_plug_point_bytes_equal_the_oracles(lib, oracle):
    """setup -> prove -> verify (sp1.rs:113-120) for three shards of SP1's structure at a small shape: vk == the oracle's commitment to the restatement's
    preprocessed columns (the PROGRAM's stream), every shard proof in the blob == the oracle's keyed-machine proof on the restatement's traces (the REQUEST's
    stream, public values = request digest | shard index); shards in flight do not change a byte; the blob is checked on the host"""
    O = oracle
    q, pb = 3, 1
    oprm = O.default_params(1, q, pb)
    plan = mplan(SP1_SMALL, PRE, 3, q, pb)
    rc, err, out, blob, vk = prove(lib, 2, plan)
    assert rc == 0, err
    assert lib.zktls_batch_flags(blob, len(blob)) == SYNTHETIC | MACHINE
    ent, offs, lens = entries(lib, blob)
    assert len(ent) == 3
    dg = digest_words(lib, CBOR, ELF)
    seed, key_seed = stream_seed(dg), stream_seed(digest_words(lib, b"", ELF))
    lns = [c[0] for c in SP1_SMALL]
    for s in range(3):
        mains, pres, progs, tabs, _ = M.sp1_shaped_machine(SP1_SMALL, seed=seed, shard=s, pre=PRE, n_public=9, key_seed=key_seed)
        assert vk[:32] == O.machine_setup(pres, lns, oprm).tobytes(), "the key differs from the oracle's"
        assert ent[s] == O.prove_machine_keyed(mains, pres, progs, tabs, dg + [s], oprm).tobytes(), "shard %d: proof bytes differ from the oracle's" % s
    assert vk[32:] == bytes(np.array(digest_words(lib, b"", ELF), dtype=np.uint32))
    assert check(lib, blob, plan, vk) == 0
    assert check(lib, blob, plan, vk, cbor=CBOR + b"!") == -2
    swapped = bytearray(blob)
    assert lens[0] == lens[1]
    swapped[offs[0]:offs[0] + lens[0]], swapped[offs[1]:offs[1] + lens[1]] = ent[1], ent[0]
    assert check(lib, bytes(swapped), plan, vk) == -2                         # shard 1's proof in shard 0's place
    # the same request with one shard at a time, and without setup() before prove(): the same bytes, the same key
    rc, err, _, blob1, vk1 = prove(lib, 2, mplan(SP1_SMALL, PRE, 3, q, pb, in_flight=1), setup_first=0)
    assert rc == 0 and blob1 == blob and vk1 == vk
    # setup() saw another program: prove refuses
    err = C.create_string_buffer(512)
    vkb = C.create_string_buffer(64)
    assert lib.zktls_machine_setup(0, 2, C.byref(plan), ELF, len(ELF), vkb, err, 512) == 0 and vkb.raw == vk
    lib.zktls_release_cached()


@pytest.mark.gpu
def test_machine_shards_core_to_compress_behind_the_same_call(lib, ctx):
    """core -> compress (sp1.rs:116) for machine shards: ONE machine-mode proof replaces the shard proofs in the blob -- its bytes == zkhip_prove_machine_verifier's on the
    shard proofs of the plain call --; with joins of at most two, five shards -> three joins -> ONE proof above them (flag TREE).  Both checked on the host from (plan, input, ELF, vk)"""
    from zktls_amd.device import Sp1ShapedShard
    from zktls_amd._lib import Params
    q, pb = 3, 1
    plan = mplan(SP1_SMALL, PRE, 3, q, pb)
    rc, err, _, plain, vk = prove(lib, 2, plan)
    assert rc == 0, err
    rc, err, _, blob, vk2 = prove(lib, 2, plan, compress=1)
    assert rc == 0, err
    assert vk2 == vk and lib.zktls_batch_flags(blob, len(blob)) == SYNTHETIC | MACHINE | COMPRESSED
    ent, offs, lens = entries(lib, blob)
    assert len(ent) == 2 and lens[1] == 36 and int.from_bytes(ent[1][32:], "little") == 3
    # the join is the library's machine-mode proof over the plain call's shard proofs
    shape = Sp1ShapedShard(SP1_SMALL, PRE, 9)
    prm = Params(1, q, pb)
    im = shape.inner_machine(np.frombuffer(vk[:32], dtype=np.uint32), prm)
    dg = digest_words(lib, CBOR, ELF)
    shard_proofs = [np.frombuffer(e, dtype=np.uint8) for e in entries(lib, plain)[0]]
    jkey = ctx.machine_verifier_setup(im, prm, 3)
    assert jkey.root.tobytes() == ent[1][:32]
    assert ctx.prove_machine_verifier(jkey, im, shard_proofs, [dg + [s] for s in range(3)], prm).tobytes() == ent[0]
    jkey.close()
    rc, err, _, again, _ = prove(lib, 2, plan, compress=1)                        # the second request of a plan finds the join's key parked with its context: the same bytes
    assert rc == 0 and again == blob
    assert check(lib, blob, plan, vk) == 0
    assert check(lib, blob, plan, vk, cbor=CBOR + b"!") == -2
    assert check(lib, plain, plan, vk) == 0
    lib.zktls_set_compress_join_size(2)
    try:
        tplan = mplan(SP1_SMALL, PRE, 5, q, pb)
        rc, err, _, tblob, tvk = prove(lib, 2, tplan, compress=1)
        assert rc == 0, err
        assert tvk == vk and lib.zktls_batch_flags(tblob, len(tblob)) == SYNTHETIC | MACHINE | COMPRESSED | TREE
        ent, offs, lens = entries(lib, tblob)
        assert len(ent) == 2 and lens[1] == 36 and int.from_bytes(ent[1][32:], "little") == 5
        assert check(lib, tblob, tplan, vk) == 0
        assert check(lib, tblob, tplan, vk, cbor=CBOR + b"!") == -2
        bad = bytearray(tblob)
        bad[offs[0] + lens[0] // 2] ^= 1
        assert check(lib, bytes(bad), tplan, vk) == -2
        one, tree, vkp = fixture_paths()
        if os.path.exists(one):                                                # the committed fixtures are these bytes
            assert open(one, "rb").read() == blob and open(tree, "rb").read() == tblob and open(vkp, "rb").read() == vk
    finally:
        lib.zktls_set_compress_join_size(0)
        lib.zktls_release_cached()
