"""Malformed input against every HOST entry that takes untrusted bytes (verifiers, program / table validators, the bincode reader).
Nothing may crash or read out of bounds: run it against the AddressSanitizer build (tools/asan_cpu.sh).  usage: fuzz_host.py [seconds]"""
import ctypes as C
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import zktls_amd._lib as _lib
if os.environ.get("ZKHIP_FUZZ_LIB"):                       # the sanitizer build (this tool only; the package never reads the environment)
    _lib.LIB_PATH = os.environ["ZKHIP_FUZZ_LIB"]
import airs
import machines
import oracle_lib as O
import sha256_air as S
from zktls_amd._lib import Params
from zktls_amd.device import verify_chips, verify_chips_air, verify_machine, verify_machine_keyed, verify_sha256, verify_shard, verify_shard_air

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(int(os.environ.get("ZKHIP_FUZZ_SEED", "1")))
L = _lib.load()
u8p, u32p = _lib.u8p, _lib.u32p


def mutate(b):
    b = bytearray(b)
    kind = rng.integers(0, 6)
    if kind == 0 and len(b) > 8:
        del b[int(rng.integers(0, len(b))):]                                   # truncate
    elif kind == 1:
        b += bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
    elif kind == 2 and len(b) >= 4:
        for _ in range(int(rng.integers(1, 8))):
            off = 4 * int(rng.integers(0, len(b) // 4))
            b[off:off + 4] = int(rng.integers(0, 2**32)).to_bytes(4, "little")
    elif kind == 3 and len(b) >= 4:
        off = 4 * int(rng.integers(0, min(len(b) // 4, 40)))                   # header area
        b[off:off + 4] = int(rng.choice([0, 1, 5, 31, 32, 2**31, 2**32 - 1, 2013265920, 2013265921])).to_bytes(4, "little")
    elif kind == 4 and len(b) > 0:
        for _ in range(int(rng.integers(1, 16))):
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
    else:
        b = bytearray(rng.integers(0, 256, int(rng.integers(0, 4096)), dtype=np.uint8).tobytes())
    return bytes(b)


def arr(b):
    return np.frombuffer(b, dtype=np.uint8) if len(b) else np.zeros(0, dtype=np.uint8)


def words(b):
    b = b[:len(b) // 4 * 4]
    return np.frombuffer(b, dtype=np.uint32).copy() if len(b) else np.zeros(1, dtype=np.uint32)


# valid material from the oracle
prm, oprm = Params(1, 4, 3), O.default_params(1, 4, 3)
t1 = O.gen_trace(7, 1, 6, 8)
p_single = O.prove_shard(t1, [1, 2], oprm).tobytes()
fib = airs.fibonacci_program()
ft, fpub = airs.fibonacci_trace(6, 3, 5)
p_air = O.prove_shard_air(fib, ft, fpub, oprm).tobytes()
p_chips = O.prove_chips([O.gen_trace(7, 2, 7, 8), O.gen_trace(7, 3, 6, 4)], [1], oprm).tobytes()
cnt = airs.counter_program(8).copy(); cnt[4] = 3
p_chips_air = O.prove_chips_air([airs.counter_trace(7, 8, 3, 5)[0], ft], [cnt, fib], fpub, oprm).tobytes()
mt, mp, mtab, mpub = machines.range_machine(5, 6)
p_machine = O.prove_machine(mt, mp, mtab, mpub, oprm).tobytes()
mln, mws = [t.shape[0].bit_length() - 1 for t in mt], [t.shape[1] for t in mt]
kt, kpre, kp, ktab, kpub = machines.byte_machine(6, 3)
kln, kws, kpw = [t.shape[0].bit_length() - 1 for t in kt], [t.shape[1] for t in kt], [0 if p_ is None else p_.shape[1] for p_ in kpre]
kroot = O.machine_setup(kpre, kln, oprm)
p_keyed = O.prove_machine_keyed(kt, kpre, kp, ktab, kpub, oprm).tobytes()
sha_t, sha_pub = S.trace(S.pad(b"abc"))
p_sha = O.prove_shard_air(S.program(), sha_t, sha_pub, oprm).tobytes()
bc_size = L.zkhip_bincode_size(6, 8, C.byref(prm))
bc = np.zeros(bc_size, dtype=np.uint8)
got = C.c_size_t(0)
src = arr(p_single)
assert L.zkhip_proof_to_bincode(src.ctypes.data_as(u8p), src.size, 6, 8, C.byref(prm), bc.ctypes.data_as(u8p), bc_size, C.byref(got)) == 0

# a chain of two SHA-256 shard proofs (chained chip) for zkhip_verify_sha256_sharded
cblocks = S.pad(bytes(range(190)))
ct0, cout0 = S.trace(cblocks[:128], message_len=190, first_block=0)
civ1 = [cout0[2 * k] | (cout0[2 * k + 1] << 16) for k in range(8)]
ct1, cout1 = S.trace(cblocks[128:], chain_in=civ1, message_len=190, first_block=2)
civ = []
for x in S.IV:
    civ += [x & 0xffff, x >> 16]
cprog = S.program(chained=True)
cp0, cp1 = O.prove_shard_air(cprog, ct0, S.chained_publics(cout0, civ), oprm), O.prove_shard_air(cprog, ct1, S.chained_publics(cout1, cout0[:16]), oprm)
cstride = max(cp0.size, cp1.size)
cbuf = np.zeros(2 * cstride, dtype=np.uint8); cbuf[:cp0.size] = cp0; cbuf[cstride:cstride + cp1.size] = cp1
cchain = np.array([S.IV, civ1, [cout1[2 * k] | (cout1[2 * k + 1] << 16) for k in range(8)]], dtype=np.uint32)
cdigest = np.frombuffer(hashlib.sha256(bytes(range(190))).digest(), dtype=np.uint8)
szp = C.POINTER(C.c_size_t)
ksrc = arr(p_keyed)
L.zkhip_chips_bincode_size.restype = C.c_size_t
cbc = np.zeros(L.zkhip_chips_bincode_size(ksrc.ctypes.data_as(u8p), ksrc.size), dtype=np.uint8)
kpv = np.array(kpub, dtype=np.uint32)
assert L.zkhip_chips_proof_to_bincode(ksrc.ctypes.data_as(u8p), ksrc.size, kpv.ctypes.data_as(u32p), kpv.size, cbc.ctypes.data_as(u8p), cbc.size, C.byref(got)) == 0
back2, pub2, got2 = np.zeros(len(p_keyed) + 64, dtype=np.uint8), np.zeros(8, dtype=np.uint32), C.c_size_t(0)

# a golden shard proof, its FRI view, and the oracle's proof of its query-phase machine (the recursion entries' host side)
import json
import fri_air as F
from zktls_amd.device import fri_view_shard_paths, fri_view_transcript, fri_view_witness
_kat = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_kat.json")))["golden_proof_files"]["v1_6x8"]
g_proof = np.fromfile(os.path.join(ROOT, "tests", "golden", "proofs", "v1_6x8.bin"), dtype=np.uint8)
g_prm = Params(*_kat["shape"])
g_view = fri_view_shard_paths(g_proof, _kat["log_n"], _kat["width"], _kat["public"], g_prm)
_, _, g_cap, _ = fri_view_transcript(g_proof, _kat["log_n"], _kat["width"], _kat["public"], g_prm)
g_wit = fri_view_witness(g_proof, _kat["log_n"], _kat["width"], _kat["public"], g_prm)
q_tr, q_pre, q_pg, q_tb, q_pub = F.machine_layers(g_view, capacity=g_cap, query_phase=(g_wit, _kat["shape"][2]))
q_lns = [t.shape[0].bit_length() - 1 for t in q_tr]
q_root = O.machine_setup(q_pre, q_lns, oprm)
p_query = O.prove_machine_keyed(q_tr, q_pre, q_pg, q_tb, q_pub, oprm).tobytes()
g_R, g_Q = _kat["log_n"], _kat["shape"][1]
g_pv = np.array(_kat["public"], dtype=np.uint32)
L.zkhip_fri_view_path_words.restype = C.c_size_t
# (sized for the largest shape the loop asks for: the entry fills what the ARGUMENTS say, the bytes only decide whether it gets there)
g_out = [np.zeros(n_, dtype=np.uint32) for n_ in (4 * 8, 4, g_Q, 4 * g_Q, 4 * g_Q * 8, 8 * 8, L.zkhip_fri_view_path_words(8) * g_Q, 10)]
q_fin, q_capv, q_vk = np.array(g_view["final"], dtype=np.uint32), np.array(g_cap, dtype=np.uint32), np.array(q_root, dtype=np.uint32)

# the shard verifier machine: the oracle's outer proof for the golden shard proof, a join of two, and the host verifier that takes the shape,
# the public values and the key (zkhip_verify_shard_recursive); its size / describe entries with hostile shapes
import recursion_air as RA
from zktls_amd.device import verify_shard_recursive
r_sh, r_mains, r_pres, r_progs, r_tabs, r_pv = RA.machine(g_proof.tobytes(), _kat["log_n"], _kat["width"], _kat["public"], g_Q, _kat["shape"][2])
r_lns = [m_.shape[0].bit_length() - 1 for m_ in r_mains]
r_root = O.machine_setup(r_pres, r_lns, oprm)
p_rec = O.prove_machine_keyed(r_mains, r_pres, r_progs, r_tabs, r_pv, oprm).tobytes()
assert verify_shard_recursive(arr(p_rec), _kat["log_n"], _kat["width"], g_Q, _kat["shape"][2], _kat["public"], r_root, prm) == (0, 0)
L.zkhip_shard_verifier_proof_size.restype = C.c_size_t
L.zkhip_shard_verifier_describe.restype = C.c_size_t
r_desc = np.zeros(1 << 16, dtype=np.uint32)

# machine mode: the byte machine's keyed proof as the inner proof -- the host side of zkhip_prove_machine_verifier (its tables ARE the inner proof's
# verification: zkhip_machine_verifier_host_tables) with malformed proofs, and descriptions with hostile numbers
from zktls_amd.device import InnerMachine, machine_verifier_host_tables
m_chips = [dict(ln=kln[c_], W=kws[c_], Pw=kpw[c_], prog=kp[c_], tab=ktab[c_]) for c_ in range(len(kt))]
m_im = InnerMachine(m_chips, [int(x) for x in kroot], 4, 3, len(kpub))
assert machine_verifier_host_tables(m_im, [arr(p_keyed)], [kpub], 1) is not None
L.zkhip_machine_verifier_proof_size.restype = C.c_size_t

gm0 = arr(g_proof.tobytes())
b_pv, b_out, b_jl, b_vk = np.zeros(64 * 4097, dtype=np.uint32), np.zeros(32 << 20, dtype=np.uint8), (C.c_size_t * 64)(), np.zeros(8, dtype=np.uint32)
t0, n = time.time(), 0
while time.time() - t0 < budget:
    machine_verifier_host_tables(m_im, [arr(mutate(p_keyed))], [kpub if rng.random() < 0.7 else [int(rng.integers(0, 2**32))]], int(rng.choice([0, 1, 5, 9])))
    h_chips = [dict(c_) for c_ in m_chips]
    hc = h_chips[int(rng.integers(0, len(h_chips)))]
    which_ = int(rng.integers(0, 5))
    if which_ == 0: hc["ln"] = int(rng.choice([4, 5, 21, 22, 31, -1]))
    elif which_ == 1: hc["W"] = int(rng.choice([0, 2, 4, 1024, 1028, 2**31]))
    elif which_ == 2: hc["Pw"] = int(rng.choice([0, 2, 4, 1020, 1024, 2**31]))
    elif which_ == 3: hc["prog"] = words(mutate(np.asarray(hc["prog"], dtype=np.uint32).tobytes()))
    else: hc["tab"] = words(mutate(np.asarray(hc["tab"], dtype=np.uint32).tobytes()))
    h_im = InnerMachine(h_chips, [int(x) for x in rng.integers(0, 2**32, 8, dtype=np.uint64)], int(rng.choice([4, 0, 1024, 1025, -1])), int(rng.choice([3, 28, 29, -1])), int(rng.choice([len(kpub), 0, 4096, 4097])))
    L.zkhip_machine_verifier_proof_size(C.byref(h_im.desc), int(rng.choice([1, 2, 64, 65, 0])), C.byref(prm))
    machine_verifier_host_tables(h_im, [arr(p_keyed)], [[1] * h_im.n_public if h_im.n_public <= 8 else kpub], 0) if h_im.n_public in (len(kpub),) else None
    verify_shard_recursive(arr(mutate(p_rec)), int(rng.choice([_kat["log_n"], _kat["log_n"], 2, 22, 23, -1])), int(rng.choice([_kat["width"], _kat["width"], 4, 1024, 2**31])),
                           int(rng.choice([g_Q, g_Q, 1, 1024, 2**20])), int(rng.choice([_kat["shape"][2], 0, 30, 31])), _kat["public"] if rng.random() < 0.7 else [int(x) for x in rng.integers(0, 2**32, int(rng.integers(0, 70)))],
                           r_root if rng.random() < 0.7 else rng.integers(0, 2**32, 8, dtype=np.uint64).astype(np.uint32), prm, n_proofs=int(rng.choice([1, 1, 2, 64, 65, 0])))
    lr_, mw_, pw_ = C.c_int(0), C.c_uint32(0), C.c_uint32(0)
    L.zkhip_shard_verifier_describe(int(rng.choice([6, 2, 22, 23, -5])), int(rng.choice([8, 16, 1024, 12, 2**31])), int(rng.choice([4, 1, 1024, 2**33])), int(rng.choice([3, 0, 30, 99])),
                                    int(rng.choice([2, 0, 64, 65])), int(rng.choice([1, 2, 64, 65, 0])), int(rng.choice([0, 3, 7, 8, -1])), int(rng.choice([0, 1, 2, 3])),
                                    r_desc.ctypes.data_as(u32p), int(rng.choice([r_desc.size, 0, 5])), C.byref(lr_), C.byref(mw_), C.byref(pw_))
    L.zkhip_shard_verifier_proof_size(int(rng.choice([6, 20, 22, 23])), int(rng.choice([8, 256, 1024, 2**31])), int(rng.choice([4, 100, 1024, 2**33])), int(rng.choice([3, 16, 31])),
                                      int(rng.choice([2, 9, 64, 65])), int(rng.choice([1, 16, 64, 65])), C.byref(prm))
    # the batch of joins: whatever the numbers, it answers from its argument checks / the size query (no device here: ZKHIP_ERR_NO_DEVICE at the latest)
    b_n, b_j = int(rng.choice([4, 3, 0, 64])), int(rng.choice([2, 0, 3, 2**40]))
    b_ptrs = (u8p * 64)(*[gm0.ctypes.data_as(u8p)] * 64)
    b_lens = (C.c_size_t * 64)(*[gm0.size] * 64)
    L.zkhip_prove_shard_verifier_batch(None, int(rng.choice([0, 0, -1, 3])), b_ptrs, b_lens, b_n, b_j, int(rng.choice([_kat["log_n"], 2, 23])), int(rng.choice([_kat["width"], 12, 2**31])),
                                              b_pv.ctypes.data_as(u32p), int(rng.choice([2, 0, 4097])), C.byref(prm), C.byref(prm), int(rng.choice([0, 2, -5, 2**20])), 1,
                                              b_out.ctypes.data_as(u8p), int(rng.choice([b_out.size // 32, 0, 5])), b_jl, b_vk.ctypes.data_as(u32p))
    gm = arr(mutate(g_proof.tobytes()))
    L.zkhip_fri_view_all(gm.ctypes.data_as(u8p), gm.size, int(rng.choice([g_R, g_R, g_R + 1, 2])), int(rng.choice([_kat["width"], 4])), g_pv.ctypes.data_as(u32p), g_pv.size,
                         C.byref(g_prm), *[a.ctypes.data_as(u32p) for a in g_out])
    qm = arr(mutate(p_query))
    why_q = C.c_int(0)
    L.zkhip_verify_fri_indices(qm.ctypes.data_as(u8p), qm.size, int(rng.choice([g_R, g_R, 2, 22, 23])), int(rng.choice([g_Q, g_Q, 1, 65536, 2**40])), int(rng.choice([_kat["shape"][2], 0, 30, 31, -1])),
                               q_fin.ctypes.data_as(u32p), q_capv.ctypes.data_as(u32p), q_vk.ctypes.data_as(u32p), C.byref(prm), C.byref(why_q))
    verify_shard(arr(mutate(p_single)), int(rng.choice([6, 6, 6, 5, 7])), int(rng.choice([8, 8, 4, 12])), [1, 2], prm)
    verify_shard_air(fib, arr(mutate(p_air)), 6, 4, fpub, prm)
    verify_shard_air(words(mutate(fib.tobytes())), arr(p_air), 6, 4, fpub, prm)
    verify_chips(arr(mutate(p_chips)), [7, 6], [8, 4], [1], prm)
    verify_chips_air(arr(mutate(p_chips_air)), [7, 6], [8, 4], [cnt, fib], fpub, prm)
    verify_chips_air(arr(p_chips_air), [7, 6], [8, 4], [words(mutate(cnt.tobytes())), fib], fpub, prm)
    verify_machine(arr(mutate(p_machine)), mln, mws, mp, mtab, mpub, prm)
    verify_machine(arr(p_machine), mln, mws, mp, [words(mutate(mtab[0].tobytes())), mtab[1], mtab[2]], mpub, prm)
    verify_machine_keyed(arr(mutate(p_keyed)), kln, kws, kpw, kroot, kp, ktab, kpub, prm)
    verify_machine_keyed(arr(p_keyed), kln, kws, [int(x) for x in rng.choice([0, 4, 8, 1020, 1024, 2**31], len(kpw))], rng.integers(0, 2**32, 8, dtype=np.uint64).astype(np.uint32), kp, ktab, kpub, prm)
    verify_sha256(arr(mutate(p_sha)) if rng.random() < 0.7 else arr(p_sha), bytes(rng.integers(0, 256, 32, dtype=np.uint8)), prm, int(rng.choice([3, 0, 55, 56, 2**20, 2**40, 2**64 - 1])))
    w = words(mutate(fib.tobytes()))
    L.zkhip_air_validate(w.ctypes.data_as(u32p), w.size, int(rng.choice([4, 8])), int(rng.choice([3, 0])))
    out8 = (C.c_uint32 * 8)()
    L.zkhip_air_digest(w.ctypes.data_as(u32p), w.size, out8)
    m = arr(mutate(bc.tobytes()))
    back = np.zeros(len(p_single) + 64, dtype=np.uint8)
    L.zkhip_proof_from_bincode(m.ctypes.data_as(u8p), m.size, 6, 8, C.byref(prm), 2, back.ctypes.data_as(u8p), int(rng.choice([back.size, 16, 0])), C.byref(got))
    mb = np.frombuffer(mutate(cbuf.tobytes())[:cbuf.size].ljust(cbuf.size, b"\0"), dtype=np.uint8).copy()
    mc = cchain.copy()
    if rng.random() < 0.3:
        mc[int(rng.integers(0, 3)), int(rng.integers(0, 8))] ^= 1 << int(rng.integers(0, 32))
    clens = (C.c_size_t * 2)(int(rng.choice([cp0.size, 0, 15, cstride, cstride + 1, 2**40])), cp1.size)
    bad_s, why = C.c_size_t(0), C.c_int(0)
    L.zkhip_verify_sha256_sharded(mb.ctypes.data_as(u8p), cstride, clens, int(rng.choice([2, 2, 1])), mc.ctypes.data_as(u32p), int(rng.choice([1, 1, 0, 14, 99])),
                                  cdigest.ctypes.data_as(u8p), int(rng.choice([190, 190, 189, 0, 2**33, 2**64 - 1])), C.byref(prm), C.byref(bad_s), C.byref(why))
    cm = arr(mutate(p_keyed if rng.random() < 0.5 else p_machine))
    L.zkhip_chips_bincode_size(cm.ctypes.data_as(u8p), cm.size)
    cb = arr(mutate(cbc.tobytes()))
    L.zkhip_chips_proof_from_bincode(cb.ctypes.data_as(u8p), cb.size, back2.ctypes.data_as(u8p), int(rng.choice([back2.size, 64, 0])), C.byref(got),
                                     pub2.ctypes.data_as(u32p), int(rng.choice([8, 0])), C.byref(got2))
    n += 1
print("fuzz ok: %d rounds of 25 malformed calls in %.0f s (no crash; run under the sanitizer build for out-of-bounds reads)" % (n, time.time() - t0))
