"""Independent pure-Python VERIFIER for single-matrix shard proofs (versions 1-3), written from the protocol description in
DESIGN.md sections 3 and 6 on top of the first-principles primitives of tests/pyref.py (ints and pow(); Poseidon2 by explicit
matrices).  It shares no code with oracle/ (C, canonical arithmetic) or with the product's host verifier (C++, Montgomery
arithmetic): a misreading of the protocol common to those two would have to be repeated here a third time, from the prose.

Test infrastructure only.  Everything is canonical residues; a proof is a sequence of little-endian u32 words.
"""
import struct

import pyref
from pyref import P, bitrev, ext_inv, ext_mul, ext_pow, two_adic_generator

GEN = 31                       # coset shift g of every committed LDE
MAGIC = 0x41544B5A             # "ZKTA"


class Reject(Exception):
    pass


# ---------------------------------------------------------------- extension-field helpers (x^4 = 11)
def e_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def e_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def e_scale(a, k):
    return [x * k % P for x in a]


def e_base(k):
    return [k % P, 0, 0, 0]


ZERO, ONE = [0, 0, 0, 0], [1, 0, 0, 0]


def e_from_columns(four):
    """an extension column is committed as 4 base columns; its opening is sum_e X^e * (opening of column e)"""
    acc = ZERO
    for e in range(4):
        basis = [0, 0, 0, 0]
        basis[e] = 1
        acc = e_add(acc, ext_mul(basis, four[e]))
    return acc


# ---------------------------------------------------------------- hashing by shape
class Hash:
    def __init__(self, width):
        if width == 16:
            self.leaf, self.node = pyref.sponge_hash, pyref.compress
        elif width == 24:
            self.leaf, self.node = pyref.sponge24, pyref.compress24
        else:
            raise Reject("hash width")

    def root_from_path(self, row, index, siblings):
        cur = self.leaf(row)
        for lvl, sib in enumerate(siblings):
            cur = self.node(sib, cur) if (index >> lvl) & 1 else self.node(cur, sib)
        return cur


# ---------------------------------------------------------------- Fiat-Shamir: duplex sponge, rate 8, Poseidon2 width 16
class Transcript:
    def __init__(self):
        self.state = [0] * 16
        self.pending = []          # absorbed, not yet permuted
        self.ready = []            # squeezed words; taken from the END

    def _duplex(self):
        for i, v in enumerate(self.pending):
            self.state[i] = v
        self.pending = []
        self.state = pyref.poseidon2(self.state)
        self.ready = list(self.state[:8])

    def observe(self, v):
        self.ready = []
        self.pending.append(v % P)
        if len(self.pending) == 8:
            self._duplex()

    def observe_many(self, vs):
        for v in vs:
            self.observe(v)

    def sample(self):
        if self.pending or not self.ready:
            self._duplex()
        return self.ready.pop()

    def sample_ext(self):
        return [self.sample() for _ in range(4)]

    def sample_bits(self, bits):
        return self.sample() & ((1 << bits) - 1)


# ---------------------------------------------------------------- the verifier
AIR_MAGIC = 0x50524941         # "AIRP": a constraint program (the AIR as data)


def air_digest(program):
    """width-16 sponge over the 16-bit halves of every program word"""
    limbs = []
    for wd in program:
        limbs += [int(wd) & 0xFFFF, int(wd) >> 16]
    return pyref.sponge_hash(limbs)


def air_log_quotient_degree(program):
    """degree (largest number of factors in a term, + 1 under a selector) <= 3: two quotient chunks; 4 or 5: four"""
    prog = [int(x) for x in program]
    maxd, p = 0, 6
    for _ in range(prog[3]):
        sel, nt = prog[p], prog[p + 1]
        p += 2
        for _t in range(nt):
            d = prog[p + 1]
            maxd = max(maxd, d + (1 if sel else 0))
            p += 2 + d
    return 1 if maxd <= 3 else 2


def air_fold(program, loc, nxt, public_values, sel_first, sel_last, sel_trans, alpha):
    """acc = acc * alpha + selector * sum_t coeff_t * prod_j var_tj, constraint by constraint, on extension values"""
    prog = [int(x) for x in program]
    if prog[0] != AIR_MAGIC or prog[1] != 1 or prog[5] != len(prog):
        raise Reject("constraint program")
    acc, p = ZERO, 6
    for _ in range(prog[3]):
        sel, nt = prog[p], prog[p + 1]
        p += 2
        c = ZERO
        for _t in range(nt):
            prod, d = e_base(prog[p]), prog[p + 1]
            p += 2
            for _j in range(d):
                v = prog[p]
                p += 1
                kind, idx = v >> 30, v & 0xFFFF
                prod = ext_mul(prod, loc[idx] if kind == 0 else (nxt[idx] if kind == 1 else e_base(public_values[idx])))
            c = e_add(c, prod)
        if sel:
            c = ext_mul(c, {1: sel_first, 2: sel_last, 3: sel_trans}[sel])
        acc = e_add(ext_mul(acc, alpha), c)
    if p != len(prog):
        raise Reject("constraint program")
    return acc


def verify(proof_bytes, log_n, width, public_values, log_blowup=1, num_queries=100, pow_bits=16, logup_pairs=0,
           log_fold=0, log_final=0, hash_width=0, code_width=0, air=None, view=None):
    """raises Reject(reason) or returns True.  Parameter defaults = the SP1 shape.
    view: a dict that receives what the FRI check reads -- {"betas", "final", "queries": [(index, reduced opening, siblings)]} --
    for fold-by-2 proofs (tests/fri_air.py builds the FRI-fold chip's trace from it).
    air: a constraint program (u32 words) replacing the built-in synthetic AIR (proof version 7).
    code_width: RISC Zero's group order -- the first code_width columns and the rest are committed separately (version 8)."""
    if len(proof_bytes) % 4:
        raise Reject("length")
    w = list(struct.unpack("<%dI" % (len(proof_bytes) // 4), bytes(proof_bytes)))
    K = log_fold or 1
    F = log_final
    hw = hash_width or 16
    b = log_blowup
    default_shape = (b == 1 and K == 1 and F == 0 and hw == 16)
    if (log_n - F) % K or F > log_n:
        raise Reject("shape")
    R = (log_n - F) // K                       # committed FRI layers
    H = log_n + b                              # log2 of the LDE height
    N = 1 << log_n
    Q = logup_pairs
    Wp = 4 * (Q + 1) if Q else 0               # permutation-trace width in base columns
    lqd = air_log_quotient_degree(air) if air is not None else 1
    if lqd > b:
        raise Reject("the quotient domain must lie inside the committed LDE domain")
    NQ = 1 << lqd                              # quotient chunks, each committed as 4 base columns
    QW = 4 * NQ
    hasher = Hash(hw)

    # ---- header
    CW = code_width
    if CW and (air is not None or CW % 4 or not 0 < CW < width):
        raise Reject("shape")
    version = 7 if air is not None else (8 if CW else (1 if default_shape and not Q else (2 if default_shape else 3)))
    head = [MAGIC, version, log_n, width, b, num_queries, pow_bits, len(public_values)]
    if version in (3, 7, 8):
        head += [Q, K, F, hw]
    elif version == 2:
        head += [Q]
    if air is not None:
        if Q:
            raise Reject("a constraint program excludes the built-in lookup argument")
        head += air_digest(air)
    if CW:
        head += [CW]
    if w[:len(head)] != head:
        raise Reject("header")
    pos = len(head)
    if any(v >= P for v in w[pos:]) or any(v >= P for v in public_values):
        raise Reject("non-canonical word")

    def take(n):
        nonlocal pos
        out = w[pos:pos + n]
        if len(out) != n:
            raise Reject("truncated")
        pos += n
        return out

    def take_ext(n):
        flat = take(4 * n)
        return [flat[4 * i:4 * i + 4] for i in range(n)]

    # ---- transcript up to zeta
    ts = Transcript()
    ts.observe_many(head[2:])                  # every header word after magic and version
    code_root = None
    if CW:                                     # code group first, then data (then accum = permutation, then check = quotient)
        code_root = take(8)
        ts.observe_many(code_root)
    trace_root = take(8)
    ts.observe_many(trace_root)
    ts.observe_many(public_values)
    gamma = beta_l = perm_root = None
    if Q:
        gamma, beta_l = ts.sample_ext(), ts.sample_ext()
        perm_root = take(8)
        ts.observe_many(perm_root)
    quot_root = take(8)
    alpha = ts.sample_ext()                    # constraint-folding challenge is drawn BEFORE the quotient root is observed
    ts.observe_many(quot_root)
    zeta = ts.sample_ext()
    wN = two_adic_generator(log_n)
    zeta_next = e_scale(zeta, wN)

    loc, nxt = take_ext(width), take_ext(width)
    pl, pn = take_ext(Wp), take_ext(Wp)
    qz = take_ext(QW)
    for group in (loc, nxt, pl, pn, qz):       # opened values are observed before the FRI batching challenge is drawn
        for e in group:
            ts.observe_many(e)

    # ---- (a) the AIR identity at zeta
    zeta_n = ext_pow(zeta, N)
    zh = e_sub(zeta_n, ONE)
    wN_inv = pow(wN, -1, P)
    sel_first = ext_mul(zh, ext_inv(e_sub(zeta, ONE)))
    sel_last = ext_mul(zh, ext_inv(e_sub(zeta, e_base(wN_inv))))
    sel_trans = e_sub(zeta, e_base(wN_inv))
    acc = ZERO

    def fold(c):
        nonlocal acc
        acc = e_add(ext_mul(acc, alpha), c)
    if air is not None:
        acc = air_fold(air, loc, nxt, public_values, sel_first, sel_last, sel_trans, alpha)
    for g in range(width // 4 if air is None else 0):
        a, bb, c, d, dn = loc[4 * g], loc[4 * g + 1], loc[4 * g + 2], loc[4 * g + 3], nxt[4 * g + 3]
        fold(e_sub(e_sub(c, ext_mul(ext_mul(a, a), bb)), e_base(g + 1)))
        fold(ext_mul(sel_trans, e_sub(e_sub(e_sub(dn, ext_mul(a, bb)), c), e_base(2 * g + 3))))
        fold(ext_mul(sel_first, e_sub(d, e_base(5 * g + 7))))
    if Q:
        sum_l = sum_n = ZERO
        for q in range(Q):
            den_s = e_add(e_add(gamma, loc[8 * q]), ext_mul(beta_l, loc[8 * q + 1]))
            den_r = e_add(e_add(gamma, loc[8 * q + 4]), ext_mul(beta_l, loc[8 * q + 5]))
            phi, phin = e_from_columns(pl[4 * q:4 * q + 4]), e_from_columns(pn[4 * q:4 * q + 4])
            fold(e_sub(ext_mul(ext_mul(phi, den_s), den_r), e_sub(den_r, den_s)))
            sum_l, sum_n = e_add(sum_l, phi), e_add(sum_n, phin)
        S, Sn = e_from_columns(pl[4 * Q:4 * Q + 4]), e_from_columns(pn[4 * Q:4 * Q + 4])
        fold(ext_mul(sel_first, e_sub(S, sum_l)))
        fold(ext_mul(sel_trans, e_sub(e_sub(Sn, S), sum_n)))
        fold(ext_mul(sel_last, S))
    # quotient = sum_k zps_k(zeta) * q_k(zeta); chunk k lives on the coset s_k <w_N>, s_k = g w_{NQ N}^k; zps_k = prod_{j != k}
    # Z_j(zeta) / Z_j(s_k) with Z_j(x) = (x / s_j)^N - 1 vanishes on every other chunk's coset
    wq = two_adic_generator(log_n + lqd)
    sN = [pow(GEN * pow(wq, k, P) % P, N, P) for k in range(NQ)]
    quotient = ZERO
    for k in range(NQ):
        zps = ONE
        for j in range(NQ):
            if j == k:
                continue
            sjn_inv = pow(sN[j], -1, P)
            num = e_sub(e_scale(zeta_n, sjn_inv), ONE)
            den = (sN[k] * sjn_inv - 1) % P
            zps = ext_mul(zps, e_scale(num, pow(den, -1, P)))
        quotient = e_add(quotient, ext_mul(zps, e_from_columns(qz[4 * k:4 * k + 4])))
    if ext_mul(acc, ext_inv(zh)) != quotient:
        raise Reject("constraints do not match the quotient at zeta")

    # ---- (b) FRI: batching challenge, layer roots, final polynomial, proof of work
    fa = ts.sample_ext()
    npow = max(width, Wp, QW)
    fap = [ONE]
    for _ in range(npow - 1):
        fap.append(ext_mul(fap[-1], fa))

    def batch(values):                         # sum_j fa^j * values[j] for extension values
        t = ZERO
        for j, v in enumerate(values):
            t = e_add(t, ext_mul(fap[j], v))
        return t

    def batch_base(row):                       # the same for base-field row entries
        t = ZERO
        for j, v in enumerate(row):
            t = e_add(t, e_scale(fap[j], v))
        return t
    y_loc, y_nxt, y_pl, y_pn, y_q = batch(loc), batch(nxt), batch(pl), batch(pn), batch(qz)
    off_next = ext_pow(fa, width)
    off_pl, off_pn, off_q = ext_pow(fa, 2 * width), ext_pow(fa, 2 * width + Wp), ext_pow(fa, 2 * width + 2 * Wp)
    layer_roots, betas = [], []
    for _ in range(R):
        r = take(8)
        ts.observe_many(r)
        layer_roots.append(r)
        betas.append(ts.sample_ext())
    final_poly = take_ext(1 << F)
    for c in final_poly:
        ts.observe_many(c)
    witness = take(1)[0]
    ts.observe(witness)
    if ts.sample_bits(pow_bits) != 0:
        raise Reject("proof of work")

    # ---- queries
    wM = two_adic_generator(H)
    arity = 1 << K

    def fold_pair(index, log_folded, beta, e0, e1):
        # entries e0, e1 are f(x), f(-x) with x = w_{2^(log_folded+1)}^bitrev(index); the folded value is
        # (e0 + e1)/2 + beta (e0 - e1)/(2x)
        x = pow(two_adic_generator(log_folded + 1), bitrev(index, log_folded), P)
        half = pow(2, -1, P)
        even = e_scale(e_add(e0, e1), half)
        odd = e_scale(e_sub(e0, e1), half * pow(x, -1, P) % P)
        return e_add(even, ext_mul(beta, odd))

    for _ in range(num_queries):
        index = ts.sample_bits(H)
        trow = take(width)
        if CW:
            cpath = [take(8) for _ in range(H)]
            if hasher.root_from_path(trow[:CW], index, cpath) != code_root:
                raise Reject("code path")
        tpath = [take(8) for _ in range(H)]
        if hasher.root_from_path(trow[CW:], index, tpath) != trace_root:
            raise Reject("trace opening")
        prow = None
        if Q:
            prow, ppath = take(Wp), [take(8) for _ in range(H)]
            if hasher.root_from_path(prow, index, ppath) != perm_root:
                raise Reject("permutation opening")
        qrow, qpath = take(QW), [take(8) for _ in range(H)]
        if hasher.root_from_path(qrow, index, qpath) != quot_root:
            raise Reject("quotient opening")
        x = GEN * pow(wM, bitrev(index, H), P) % P
        inv1 = ext_inv(e_sub(e_base(x), zeta))
        inv2 = ext_inv(e_sub(e_base(x), zeta_next))
        at, aq = batch_base(trow), batch_base(qrow)
        val = ext_mul(e_sub(at, y_loc), inv1)
        val = e_add(val, ext_mul(off_next, ext_mul(e_sub(at, y_nxt), inv2)))
        if Q:
            ap = batch_base(prow)
            val = e_add(val, ext_mul(off_pl, ext_mul(e_sub(ap, y_pl), inv1)))
            val = e_add(val, ext_mul(off_pn, ext_mul(e_sub(ap, y_pn), inv2)))
        val = e_add(val, ext_mul(off_q, ext_mul(e_sub(aq, y_q), inv1)))
        idx = index
        seen = (index, list(val), [], [])
        for l in range(R):
            lh = H - K * (l + 1)                # log2 of the rows of this layer's matrix (rows of 2^K adjacent entries)
            row, own = idx >> K, idx & (arity - 1)
            entries = []
            for j in range(arity):
                entries.append(val if j == own else take(4))

            path = [take(8) for _ in range(lh)]
            if K == 1:
                seen[2].append(list(entries[1 - own]))
                seen[3].append([list(d) for d in path])
            flat = [c for e in entries for c in e]
            if hasher.root_from_path(flat, row, path) != layer_roots[l]:
                raise Reject("FRI layer %d opening" % l)
            beta, cnt = betas[l], arity
            for j in range(K):
                cnt >>= 1
                entries = [fold_pair(row * cnt + t, lh + (K - 1 - j), beta, entries[2 * t], entries[2 * t + 1]) for t in range(cnt)]
                beta = ext_mul(beta, beta)
            val, idx = entries[0], row
        lf = F + b
        xf = pow(two_adic_generator(lf), bitrev(idx, lf), P) if lf else 1
        v = ZERO
        for c in reversed(final_poly):
            v = e_add(e_scale(v, xf), c)
        if v != val:
            raise Reject("final polynomial")
        if view is not None:
            view.setdefault("queries", []).append(seen[:3])
            view.setdefault("paths", []).append(seen[3])
            # what a verifier INSIDE a proof needs besides (tests/recursion_air.py): the opened rows with their paths
            view.setdefault("openings", []).append({"index": index, "trow": list(trow), "tpath": [list(d) for d in tpath],
                                                    "qrow": list(qrow), "qpath": [list(d) for d in qpath]})
    if pos != len(w):
        raise Reject("trailing words")
    if view is not None:
        view["betas"] = [list(x) for x in betas]
        view["roots"] = [list(r) for r in layer_roots]
        view["final"] = list(final_poly[0])
        view["trace_root"], view["quot_root"], view["witness"] = list(trace_root), list(quot_root), witness
        view["loc"], view["nxt"], view["qz"] = [list(e) for e in loc], [list(e) for e in nxt], [list(e) for e in qz]
        view["alpha"], view["zeta"], view["fa"] = list(alpha), list(zeta), list(fa)
    return True
