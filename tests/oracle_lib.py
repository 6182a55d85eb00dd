"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package zktls_amd/.
All values crossing this API are canonical residues in [0, p).
"""
import ctypes as C
import os
import subprocess

import numpy as np

P = 2013265921
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

u32p = C.POINTER(C.c_uint32)


class Params(C.Structure):
    _fields_ = [("log_blowup", C.c_int), ("num_queries", C.c_int), ("pow_bits", C.c_int), ("logup_pairs", C.c_int),
                ("log_fold", C.c_int), ("log_final", C.c_int), ("hash_width", C.c_int), ("code_width", C.c_int)]


class ProveDebug(C.Structure):
    _fields_ = [
        ("trace_root", C.c_uint32 * 8),
        ("quotient_root", C.c_uint32 * 8),
        ("alpha", C.c_uint32 * 4),
        ("zeta", C.c_uint32 * 4),
        ("fri_alpha", C.c_uint32 * 4),
        ("pow_witness", C.c_uint32),
    ]


class Challenger(C.Structure):
    _fields_ = [
        ("state", C.c_uint32 * 16),
        ("input", C.c_uint32 * 8),
        ("n_input", C.c_int),
        ("output", C.c_uint32 * 8),
        ("n_output", C.c_int),
    ]


def build(force=False):
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(so):
        so = build()
    L = C.CDLL(so)
    L.orc_set_threads.restype = C.c_int
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_bb_mul.restype = C.c_uint32
    L.orc_bb_mul.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_bb_inv.restype = C.c_uint32
    L.orc_bb_inv.argtypes = [C.c_uint32]
    L.orc_bb_pow.restype = C.c_uint32
    L.orc_bb_pow.argtypes = [C.c_uint32, C.c_uint64]
    L.orc_two_adic_generator.restype = C.c_uint32
    L.orc_two_adic_generator.argtypes = [C.c_int]
    L.orc_synth_value.restype = C.c_uint32
    L.orc_synth_value.argtypes = [C.c_uint64, C.c_uint64]
    L.orc_check_trace.restype = C.c_size_t
    L.orc_proof_size.restype = C.c_size_t
    L.orc_prove_shard.restype = C.c_size_t
    L.orc_verify_shard.restype = C.c_int
    L.orc_chal_sample.restype = C.c_uint32
    L.orc_chal_sample_bits.restype = C.c_uint32
    L.orc_chal_grind.restype = C.c_uint32
    L.orc_chal_check_witness.restype = C.c_int
    L.orc_merkle_verify.restype = C.c_int
    _LIB = L
    return L


def _p(a):
    return a.ctypes.data_as(u32p)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def to_monty(a):
    a = np.asarray(a, dtype=np.uint64)
    return ((a << np.uint64(32)) % np.uint64(P)).astype(np.uint32)


def from_monty(a):
    a = np.asarray(a, dtype=np.uint64)
    return ((a * np.uint64(943718400)) % np.uint64(P)).astype(np.uint32)


def set_threads(n):
    return lib().orc_set_threads(int(n))


def dft_naive(mat, inverse=False):
    mat = _u32(mat)
    n, w = mat.shape
    out = np.empty_like(mat)
    lib().orc_dft_naive(_p(mat), _p(out), C.c_int(n.bit_length() - 1), C.c_size_t(w), C.c_int(int(inverse)))
    return out


def ntt(mat, inverse=False):
    a = _u32(mat).copy()
    n, w = a.shape
    lib().orc_ntt(_p(a), C.c_int(n.bit_length() - 1), C.c_size_t(w), C.c_int(int(inverse)))
    return a


def coset_lde(mat, log_blowup=1, shift=31):
    mat = _u32(mat)
    n, w = mat.shape
    out = np.empty((n << log_blowup, w), dtype=np.uint32)
    lib().orc_coset_lde(_p(mat), _p(out), C.c_int(n.bit_length() - 1), C.c_size_t(w),
                        C.c_int(log_blowup), C.c_uint32(shift))
    return out


def poseidon2(state):
    s = _u32(state).copy()
    assert s.shape == (16,)
    lib().orc_poseidon2_permute(_p(s))
    return s


def sponge_hash(vals):
    v = _u32(vals).ravel()
    out = np.empty(8, dtype=np.uint32)
    lib().orc_sponge_hash(_p(v), C.c_size_t(v.size), _p(out))
    return out


def compress(l, r):
    l, r = _u32(l), _u32(r)
    out = np.empty(8, dtype=np.uint32)
    lib().orc_compress(_p(l), _p(r), _p(out))
    return out


def poseidon2_24(state):
    s = _u32(state).copy()
    assert s.shape == (24,)
    lib().orc_poseidon2_24_permute(_p(s))
    return s


def merkle_tree_p24_colmajor(mat):
    """mat: [cols][rows] column-major"""
    m = _u32(mat)
    cols, rows = m.shape
    out = np.empty((2 * rows - 1, 8), dtype=np.uint32)
    lib().orc_merkle_tree_p24_colmajor(_p(m), C.c_size_t(cols), C.c_int(rows.bit_length() - 1), _p(out))
    return out


def merkle_tree_hw(mat, hash_width):
    """one row-major matrix, hash selected by width (16 or 24); all levels, root last"""
    m = _u32(mat)
    rows, w = m.shape
    out = np.empty((2 * rows - 1, 8), dtype=np.uint32)
    lib().orc_merkle_tree_hw(_p(m), C.c_size_t(w), C.c_int(rows.bit_length() - 1), _p(out), C.c_int(hash_width))
    return out


def _mats_args(mats):
    mats = [_u32(m) for m in mats]
    ptrs = (u32p * len(mats))(*[_p(m) for m in mats])
    widths = (C.c_size_t * len(mats))(*[m.shape[1] for m in mats])
    return mats, ptrs, widths


def hash_rows(mats):
    mats, ptrs, widths = _mats_args(mats)
    h = mats[0].shape[0]
    out = np.empty((h, 8), dtype=np.uint32)
    lib().orc_hash_rows(ptrs, widths, C.c_int(len(mats)), C.c_size_t(h), _p(out))
    return out


def merkle_tree(mats):
    """equal-height matrices -> all levels [(2^(h+1)-1), 8]; root = last row"""
    mats, ptrs, widths = _mats_args(mats)
    h = mats[0].shape[0]
    log_h = h.bit_length() - 1
    out = np.empty((2 * h - 1, 8), dtype=np.uint32)
    lib().orc_merkle_tree(ptrs, widths, C.c_int(len(mats)), C.c_int(log_h), _p(out))
    return out


def merkle_tree_mixed(mats):
    mats, ptrs, widths = _mats_args(mats)
    lhs = [m.shape[0].bit_length() - 1 for m in mats]
    h = 1 << max(lhs)
    out = np.empty((2 * h - 1, 8), dtype=np.uint32)
    lib().orc_merkle_tree_mixed(ptrs, widths, (C.c_int * len(mats))(*lhs), C.c_int(len(mats)), _p(out))
    return out


def fill_uniform(seed, log_n, width):
    out = np.empty((1 << log_n, width), dtype=np.uint32)
    lib().orc_fill_uniform(C.c_uint64(seed), C.c_int(log_n), C.c_size_t(width), _p(out))
    return out


def gen_trace(seed, shard, log_n, width):
    out = np.empty((1 << log_n, width), dtype=np.uint32)
    lib().orc_gen_trace(C.c_uint64(seed), C.c_uint64(shard), C.c_int(log_n), C.c_size_t(width), _p(out))
    return out


def check_trace(trace):
    t = _u32(trace)
    n, w = t.shape
    return lib().orc_check_trace(_p(t), C.c_int(n.bit_length() - 1), C.c_size_t(w))


def quotient_values(lde, log_n, alpha):
    lde = _u32(lde)
    w = lde.shape[1]
    a = _u32(alpha)
    out = np.empty((lde.shape[0], 4), dtype=np.uint32)
    lib().orc_quotient_values(_p(lde), C.c_int(log_n), C.c_size_t(w), _p(a), _p(out))
    return out


def open_at(lde, log_n, z):
    lde = _u32(lde)
    w = lde.shape[1]
    zz = _u32(z)
    out = np.empty((w, 4), dtype=np.uint32)
    lib().orc_open_at(_p(lde), C.c_int(log_n), C.c_size_t(w), _p(zz), _p(out))
    return out


def fri_fold(vals, beta):
    v = _u32(vals)
    h = v.shape[0]
    b = _u32(beta)
    out = np.empty((h // 2, 4), dtype=np.uint32)
    lib().orc_fri_fold(_p(v), C.c_int(h.bit_length() - 1), _p(b), _p(out))
    return out


def fri_fold_k(vals, log_arity, beta):
    v = _u32(vals)
    h = v.shape[0]
    b = _u32(beta)
    out = np.empty((h >> log_arity, 4), dtype=np.uint32)
    lib().orc_fri_fold_k(_p(v), C.c_int(h.bit_length() - 1), C.c_int(log_arity), _p(b), _p(out))
    return out


def default_params(log_blowup=1, num_queries=100, pow_bits=16, logup_pairs=0, log_fold=0, log_final=0, hash_width=0, code_width=0):
    return Params(log_blowup, num_queries, pow_bits, logup_pairs, log_fold, log_final, hash_width, code_width)


def segment_params(num_queries=50, logup_pairs=0, log_final=8, code_width=0):
    """RISC Zero's shape: blowup 4, fold by 16, final polynomial of 2^log_final coefficients,
    Poseidon2 width 24, no proof of work; code_width > 0: its code / data(/ accum) / check group order"""
    return Params(2, num_queries, 0, logup_pairs, 4, log_final, 24, code_width)


def gen_trace_logup(seed, shard, log_n, width, pairs):
    out = np.empty((1 << log_n, width), dtype=np.uint32)
    lib().orc_gen_trace_logup(C.c_uint64(seed), C.c_uint64(shard), C.c_int(log_n), C.c_size_t(width), C.c_int(pairs), _p(out))
    return out


def perm_trace(trace, pairs, gamma, beta):
    t = _u32(trace)
    n, w = t.shape
    g, b = _u32(gamma), _u32(beta)
    out = np.empty((n, 4 * (pairs + 1)), dtype=np.uint32)
    lib().orc_perm_trace(_p(t), C.c_int(n.bit_length() - 1), C.c_size_t(w), C.c_int(pairs), _p(g), _p(b), _p(out))
    return out


def quotient_values_logup(lde, log_n, perm_lde, pairs, gamma, beta, alpha):
    lde, perm_lde = _u32(lde), _u32(perm_lde)
    g, b, a = _u32(gamma), _u32(beta), _u32(alpha)
    out = np.empty((lde.shape[0], 4), dtype=np.uint32)
    lib().orc_quotient_values_logup(_p(lde), C.c_int(log_n), C.c_size_t(lde.shape[1]), _p(perm_lde), C.c_int(pairs),
                                    _p(g), _p(b), _p(a), _p(out))
    return out


def proof_size(log_n, width, params, n_public=0):
    return int(lib().orc_proof_size(C.c_int(log_n), C.c_size_t(width), C.byref(params), C.c_size_t(n_public)))


def prove_shard(trace, public_values=(), params=None):
    params = params or default_params()
    t = _u32(trace)
    n, w = t.shape
    log_n = n.bit_length() - 1
    pv = _u32(np.array(public_values, dtype=np.uint32))
    size = lib().orc_proof_size(C.c_int(log_n), C.c_size_t(w), C.byref(params), C.c_size_t(pv.size))
    buf = np.empty(size, dtype=np.uint8)
    got = lib().orc_prove_shard(_p(t), C.c_int(log_n), C.c_size_t(w), _p(pv), C.c_size_t(pv.size),
                                C.byref(params), buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(size))
    if got != size:
        raise RuntimeError("oracle prove failed")
    return buf


def prove_debug():
    d = ProveDebug()
    lib().orc_last_prove_debug(C.byref(d))
    return {k: np.array(getattr(d, k), dtype=np.uint32) if k != "pow_witness" else int(d.pow_witness)
            for k, _ in ProveDebug._fields_}


def gen_trace_logup_cross(seed, shard, partner_shard, log_n, width, partner_width, pairs):
    out = np.empty(((1 << log_n), width), dtype=np.uint32)
    lib().orc_gen_trace_logup_cross(C.c_uint64(seed), C.c_uint64(shard), C.c_uint64(partner_shard), C.c_int(log_n), C.c_size_t(width),
                                    C.c_size_t(partner_width), C.c_int(pairs), _p(out))
    return out


def prove_chips(traces, public_values=(), params=None, pairs=None, partners=None):
    """traces: list of row-major canonical matrices, tallest first; pairs: in-table LogUp pairs per chip (or None)"""
    params = params or default_params()
    ts = [_u32(t) for t in traces]
    n = len(ts)
    log_ns = (C.c_int * n)(*[t.shape[0].bit_length() - 1 for t in ts])
    widths = (C.c_size_t * n)(*[t.shape[1] for t in ts])
    ptrs = (u32p * n)(*[_p(t) for t in ts])
    pv = _u32(np.array(public_values, dtype=np.uint32))
    L = lib()
    L.orc_chips_proof_size.restype = C.c_size_t
    L.orc_prove_chips.restype = C.c_size_t
    pr_ = (C.c_int * n)(*[int(x) for x in pairs]) if pairs is not None else None
    pa_ = (C.c_int * n)(*[int(x) for x in partners]) if partners is not None else None
    size = L.orc_chips_proof_size(log_ns, widths, pr_, pa_, C.c_int(n), C.byref(params), C.c_size_t(pv.size))
    if size == 0:
        raise RuntimeError("oracle: bad chip set")
    buf = np.empty(size, dtype=np.uint8)
    got = L.orc_prove_chips(ptrs, log_ns, widths, pr_, pa_, C.c_int(n), _p(pv), C.c_size_t(pv.size), C.byref(params),
                            buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(size))
    if got != size:
        raise RuntimeError("oracle prove_chips failed")
    return buf


def _progs_args(progs):
    n = len(progs)
    keep = [(_u32(p) if p is not None else None) for p in progs]
    ptrs = (u32p * n)(*[(_p(p) if p is not None else None) for p in keep])
    words = (C.c_size_t * n)(*[(p.size if p is not None else 0) for p in keep])
    return keep, ptrs, words


def prove_chips_air(traces, progs, public_values=(), params=None):
    """chips with their own constraint programs (progs[c] None: the synthetic AIR), tallest first; proof version 9"""
    params = params or default_params()
    ts = [_u32(t) for t in traces]
    n = len(ts)
    log_ns = (C.c_int * n)(*[t.shape[0].bit_length() - 1 for t in ts])
    widths = (C.c_size_t * n)(*[t.shape[1] for t in ts])
    ptrs = (u32p * n)(*[_p(t) for t in ts])
    pv = _u32(np.array(public_values, dtype=np.uint32))
    keep, pp, pw = _progs_args(progs)
    L = lib()
    L.orc_chips_proof_size_air.restype = C.c_size_t
    L.orc_prove_chips_air.restype = C.c_size_t
    size = L.orc_chips_proof_size_air(log_ns, widths, pp, pw, C.c_int(n), C.byref(params), C.c_size_t(pv.size))
    if size == 0:
        raise RuntimeError("oracle: bad chip set or program")
    buf = np.empty(size, dtype=np.uint8)
    got = L.orc_prove_chips_air(ptrs, log_ns, widths, pp, pw, C.c_int(n), _p(pv), C.c_size_t(pv.size), C.byref(params),
                                buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(size))
    if got != size:
        raise RuntimeError("oracle prove_chips_air failed")
    return buf


def verify_chips_air(proof, log_ns, widths, progs, public_values=(), params=None):
    params = params or default_params()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    n = len(log_ns)
    ln = (C.c_int * n)(*[int(x) for x in log_ns])
    ws = (C.c_size_t * n)(*[int(x) for x in widths])
    pv = _u32(np.array(public_values, dtype=np.uint32))
    keep, pp, pw = _progs_args(progs)
    return int(lib().orc_verify_chips_air(pr.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(pr.size), ln, ws, pp, pw, C.c_int(n),
                                          _p(pv), C.c_size_t(pv.size), C.byref(params)))


LKUP_MAGIC = 0x50554B4C
SEND, RECEIVE = 0, 1


def interaction_table(interactions):
    """interactions: [(SEND | RECEIVE, multiplicity column or None for the constant 1, bus, [value columns]), ...] -> flat u32 table"""
    body = []
    for sign, mult, bus, cols in interactions:
        body += [sign, 0xFFFFFFFF if mult is None else mult, bus % P, len(cols)] + list(cols)
    return np.array([LKUP_MAGIC, len(interactions), 3 + len(body)] + body, dtype=np.uint32)


def _machine_args(traces, progs, tables):
    ts = [_u32(t) for t in traces]
    n = len(ts)
    log_ns = (C.c_int * n)(*[t.shape[0].bit_length() - 1 for t in ts])
    widths = (C.c_size_t * n)(*[t.shape[1] for t in ts])
    ptrs = (u32p * n)(*[_p(t) for t in ts])
    kp, pp, pw = _progs_args(progs)
    kt, tp, tw = _progs_args(tables)
    return ts, n, log_ns, widths, ptrs, (kp, pp, pw), (kt, tp, tw)


def prove_machine(traces, progs, tables, public_values=(), params=None):
    """chips with programs (None: the synthetic AIR) and interaction tables (None: no lookups), tallest first; proof version 10"""
    params = params or default_params()
    ts, n, log_ns, widths, ptrs, (kp, pp, pw), (kt, tp, tw) = _machine_args(traces, progs, tables)
    pv = _u32(np.array(public_values, dtype=np.uint32))
    L = lib()
    L.orc_machine_proof_size.restype = C.c_size_t
    L.orc_prove_machine.restype = C.c_size_t
    size = L.orc_machine_proof_size(log_ns, widths, pp, pw, tp, tw, C.c_int(n), C.byref(params), C.c_size_t(pv.size))
    if size == 0:
        raise RuntimeError("oracle: bad machine")
    buf = np.empty(size, dtype=np.uint8)
    got = L.orc_prove_machine(ptrs, log_ns, widths, pp, pw, tp, tw, C.c_int(n), _p(pv), C.c_size_t(pv.size), C.byref(params),
                              buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(size))
    if got != size:
        raise RuntimeError("oracle prove_machine failed")
    return buf


def verify_machine(proof, log_ns, widths, progs, tables, public_values=(), params=None):
    params = params or default_params()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    n = len(log_ns)
    ln = (C.c_int * n)(*[int(x) for x in log_ns])
    ws = (C.c_size_t * n)(*[int(x) for x in widths])
    pv = _u32(np.array(public_values, dtype=np.uint32))
    kp, pp, pw = _progs_args(progs)
    kt, tp, tw = _progs_args(tables)
    return int(lib().orc_verify_machine(pr.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(pr.size), ln, ws, pp, pw, tp, tw, C.c_int(n),
                                        _p(pv), C.c_size_t(pv.size), C.byref(params)))


def _pre_args(pre_traces, n):
    """preprocessed traces (None: the chip has none) -> (kept arrays, pointer array, width array)"""
    keep = [None if t is None else _u32(t) for t in pre_traces]
    assert len(keep) == n
    ptrs = (u32p * n)(*[None if t is None else _p(t) for t in keep])
    widths = (C.c_size_t * n)(*[0 if t is None else t.shape[1] for t in keep])
    return keep, ptrs, widths


def machine_setup(pre_traces, log_ns, params=None):
    """the keyed machine's setup: commitment to the preprocessed traces (None: the chip has none), tallest chip first -> 8-word root"""
    params = params or default_params()
    n = len(pre_traces)
    keep, ptrs, pws = _pre_args(pre_traces, n)
    ln = (C.c_int * n)(*[int(x) for x in log_ns])
    root = np.zeros(8, dtype=np.uint32)
    if lib().orc_machine_setup(ptrs, ln, pws, C.c_int(n), C.byref(params), _p(root)) != 0:
        raise RuntimeError("oracle: bad preprocessed traces")
    return root


def prove_machine_keyed(traces, pre_traces, progs, tables, public_values=(), params=None):
    """a machine whose chips may have preprocessed columns (pre_traces[c], None: none); programs and tables address [pre | main]; version 11"""
    params = params or default_params()
    ts, n, log_ns, widths, ptrs, (kp, pp, pw), (kt, tp, tw) = _machine_args(traces, progs, tables)
    keep, eptrs, pws = _pre_args(pre_traces, n)
    pv = _u32(np.array(public_values, dtype=np.uint32))
    L = lib()
    L.orc_machine_proof_size_keyed.restype = C.c_size_t
    L.orc_prove_machine_keyed.restype = C.c_size_t
    size = L.orc_machine_proof_size_keyed(log_ns, widths, pws, pp, pw, tp, tw, C.c_int(n), C.byref(params), C.c_size_t(pv.size))
    if size == 0:
        raise RuntimeError("oracle: bad keyed machine")
    buf = np.empty(size, dtype=np.uint8)
    got = L.orc_prove_machine_keyed(ptrs, eptrs, log_ns, widths, pws, pp, pw, tp, tw, C.c_int(n), _p(pv), C.c_size_t(pv.size), C.byref(params),
                                    buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(size))
    if got != size:
        raise RuntimeError("oracle prove_machine_keyed failed")
    return buf


def verify_machine_keyed(proof, log_ns, widths, pre_widths, root, progs, tables, public_values=(), params=None):
    params = params or default_params()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    n = len(log_ns)
    ln = (C.c_int * n)(*[int(x) for x in log_ns])
    ws = (C.c_size_t * n)(*[int(x) for x in widths])
    pws = (C.c_size_t * n)(*[int(x) for x in pre_widths])
    rt = _u32(np.array(root, dtype=np.uint32))
    pv = _u32(np.array(public_values, dtype=np.uint32))
    kp, pp, pw = _progs_args(progs)
    kt, tp, tw = _progs_args(tables)
    return int(lib().orc_verify_machine_keyed(pr.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(pr.size), ln, ws, pws, _p(rt), pp, pw, tp, tw, C.c_int(n),
                                              _p(pv), C.c_size_t(pv.size), C.byref(params)))


def verify_chips(proof, log_ns, widths, public_values=(), params=None, pairs=None, partners=None):
    params = params or default_params()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    n = len(log_ns)
    ln = (C.c_int * n)(*[int(x) for x in log_ns])
    ws = (C.c_size_t * n)(*[int(x) for x in widths])
    pv = _u32(np.array(public_values, dtype=np.uint32))
    pr_ = (C.c_int * n)(*[int(x) for x in pairs]) if pairs is not None else None
    pa_ = (C.c_int * n)(*[int(x) for x in partners]) if partners is not None else None
    return int(lib().orc_verify_chips(pr.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(pr.size), ln, ws, pr_, pa_, C.c_int(n),
                                      _p(pv), C.c_size_t(pv.size), C.byref(params)))


def verify_shard(proof, log_n, width, public_values=(), params=None):
    params = params or default_params()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    pv = _u32(np.array(public_values, dtype=np.uint32))
    return lib().orc_verify_shard(pr.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(pr.size),
                                  C.c_int(log_n), C.c_size_t(width), _p(pv), C.c_size_t(pv.size),
                                  C.byref(params))


class OracleChallenger:
    def __init__(self):
        self.c = Challenger()
        lib().orc_chal_init(C.byref(self.c))

    def observe(self, vals):
        v = _u32(np.atleast_1d(np.asarray(vals, dtype=np.uint32))).ravel()
        lib().orc_chal_observe_slice(C.byref(self.c), _p(v), C.c_size_t(v.size))

    def sample(self):
        return lib().orc_chal_sample(C.byref(self.c))

    def sample_ext(self):
        out = np.empty(4, dtype=np.uint32)
        lib().orc_chal_sample_ext(C.byref(self.c), _p(out))
        return out

    def sample_bits(self, bits):
        return lib().orc_chal_sample_bits(C.byref(self.c), C.c_int(bits))

    def grind(self, bits):
        return lib().orc_chal_grind(C.byref(self.c), C.c_int(bits))


# ---- RISC Zero Hal operators (oracle/hal.c): column-major vectors, canonical words; ext_w = 11 or P - 11
EXT_W = {0: 11, 1: P - 11}


def hal_ext_mul(a, b, ext_field=0):
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_hal_ext_mul(_p(_u32(a)), _p(_u32(b)), C.c_uint32(EXT_W[ext_field]), _p(out))
    return out


def hal_eltwise_add(a, b):
    a, b = _u32(a).ravel(), _u32(b).ravel()
    out = np.empty_like(a)
    lib().orc_hal_eltwise_add(_p(out), _p(a), _p(b), C.c_size_t(a.size))
    return out


def hal_eltwise_sum_ext(inp, count):
    inp = _u32(inp).ravel()
    to_add = inp.size // (4 * count)
    out = np.empty(4 * count, dtype=np.uint32)
    lib().orc_hal_eltwise_sum_ext(_p(out), _p(inp), C.c_size_t(count), C.c_size_t(to_add))
    return out


def hal_eltwise_zeroize(io):
    io = _u32(io).ravel().copy()
    lib().orc_hal_eltwise_zeroize(_p(io), C.c_size_t(io.size))
    return io


def hal_zk_shift(io, count, log_size, shift):
    io = _u32(io).ravel().copy()
    lib().orc_hal_zk_shift(_p(io), C.c_size_t(count), C.c_int(log_size), C.c_uint32(shift))
    return io


def hal_mix_poly_coeffs(out, mix_start, mix, inp, combos, input_size, count, ext_field=0):
    out = _u32(out).ravel().copy()
    lib().orc_hal_mix_poly_coeffs(_p(out), _p(_u32(mix_start)), _p(_u32(mix)), _p(_u32(inp).ravel()), _p(_u32(combos)),
                                  C.c_size_t(input_size), C.c_size_t(count), C.c_uint32(EXT_W[ext_field]))
    return out


def hal_batch_evaluate_any(coeffs, log_size, which, xs, ext_field=0):
    which, xs = _u32(which), _u32(xs).ravel()
    out = np.empty(4 * which.size, dtype=np.uint32)
    lib().orc_hal_batch_evaluate_any(_p(_u32(coeffs).ravel()), C.c_int(log_size), _p(which), _p(xs), _p(out), C.c_size_t(which.size),
                                     C.c_uint32(EXT_W[ext_field]))
    return out


def hal_gather_sample(src, idx, size, stride):
    out = np.empty(size, dtype=np.uint32)
    lib().orc_hal_gather_sample(_p(out), _p(_u32(src).ravel()), C.c_size_t(idx), C.c_size_t(size), C.c_size_t(stride))
    return out


def hal_scatter(into, index, offsets, values):
    into = _u32(into).ravel().copy()
    index = _u32(index)
    lib().orc_hal_scatter(_p(into), _p(index), _p(_u32(offsets)), _p(_u32(values)), C.c_size_t(index.size - 1))
    return into


def hal_prefix_products_ext(io, ext_field=0):
    io = _u32(io).ravel().copy()
    lib().orc_hal_prefix_products_ext(_p(io), C.c_size_t(io.size // 4), C.c_uint32(EXT_W[ext_field]))
    return io


def hal_hash_rows_sha256(mat_colmajor):
    m = _u32(mat_colmajor)
    cols, rows = m.shape
    out = np.empty(8 * rows, dtype=np.uint32)
    lib().orc_hal_hash_rows_sha256(_p(m), C.c_size_t(cols), C.c_size_t(rows), _p(out))
    return out.reshape(rows, 8)


def hal_hash_fold_sha256(children):
    ch = _u32(children).ravel()
    count = ch.size // 16
    out = np.empty(8 * count, dtype=np.uint32)
    lib().orc_hal_hash_fold_sha256(_p(ch), _p(out), C.c_size_t(count))
    return out.reshape(count, 8)


# ---- constraint programs: the AIR as data (oracle/air.c)
AIR_MAGIC = 0x50524941
SEL_ALL, SEL_FIRST, SEL_LAST, SEL_TRANSITION = 0, 1, 2, 3


def air_var(col, next_row=False, public=False):
    return (2 << 30 | col) if public else ((1 << 30 | col) if next_row else col)


def air_program(width, n_public, constraints):
    """constraints: [(selector, [(coeff, [vars...]), ...]), ...] -> flat u32 program"""
    body = []
    for sel, terms in constraints:
        body += [sel, len(terms)]
        for coeff, vs in terms:
            body += [coeff % P, len(vs)] + list(vs)
    words = 6 + len(body)
    return np.array([AIR_MAGIC, 1, width, len(constraints), n_public, words] + body, dtype=np.uint32)


def air_synthetic(width, n_public):
    out = np.empty(6 + (width // 4) * 33, dtype=np.uint32)
    lib().orc_air_synthetic.restype = C.c_size_t
    n = lib().orc_air_synthetic(C.c_size_t(width), C.c_size_t(n_public), _p(out), C.c_size_t(out.size))
    assert n == out.size
    return out


def air_validate(prog, width, n_public):
    prog = _u32(prog)
    return int(lib().orc_air_validate(_p(prog), C.c_size_t(prog.size), C.c_size_t(width), C.c_size_t(n_public)))


def air_digest(prog):
    prog = _u32(prog)
    out = np.empty(8, dtype=np.uint32)
    lib().orc_air_digest(_p(prog), C.c_size_t(prog.size), _p(out))
    return out


def air_log_quotient_degree(prog):
    return int(lib().orc_air_log_quotient_degree(_p(_u32(prog))))


def quotient_values_air(prog, lde, log_n, public_values, alpha):
    lde, prog = _u32(lde), _u32(prog)
    pv = _u32(np.array(list(public_values) or [0], dtype=np.uint32))
    lqd = air_log_quotient_degree(prog)
    out = np.empty((1 << (log_n + lqd), 4), dtype=np.uint32)
    lib().orc_quotient_values_air(_p(prog), _p(lde), C.c_int(log_n), C.c_size_t(lde.shape[1]), _p(pv), _p(_u32(alpha)), C.c_int(lqd), _p(out))
    return out


def prove_shard_air(prog, trace, public_values=(), params=None):
    params = params or default_params()
    t, prog = _u32(trace), _u32(prog)
    n, w = t.shape
    log_n = n.bit_length() - 1
    pv = _u32(np.array(list(public_values) or [0], dtype=np.uint32))
    npub = len(public_values)
    L = lib()
    L.orc_proof_size_air.restype = C.c_size_t
    L.orc_prove_shard_air.restype = C.c_size_t
    size = L.orc_proof_size_air(C.c_int(log_n), C.c_size_t(w), C.byref(params), C.c_size_t(npub), C.c_int(air_log_quotient_degree(prog)))
    buf = np.empty(size, dtype=np.uint8)
    got = L.orc_prove_shard_air(_p(prog), C.c_size_t(prog.size), _p(t), C.c_int(log_n), C.c_size_t(w), _p(pv), C.c_size_t(npub), C.byref(params),
                                buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(size))
    if got != size or size == 0:
        raise RuntimeError("oracle prove_shard_air failed (trace violates the program, or bad program)")
    return buf


def verify_shard_air(prog, proof, log_n, width, public_values=(), params=None):
    params = params or default_params()
    pr, prog = np.ascontiguousarray(proof, dtype=np.uint8), _u32(prog)
    pv = _u32(np.array(list(public_values) or [0], dtype=np.uint32))
    return int(lib().orc_verify_shard_air(_p(prog), C.c_size_t(prog.size), pr.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(pr.size),
                                          C.c_int(log_n), C.c_size_t(width), _p(pv), C.c_size_t(len(public_values)), C.byref(params)))
