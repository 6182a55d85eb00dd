"""Independent pure-Python VERIFIER for multi-chip proofs: versions 4 (several tables of different heights), 5 (in-table lookups),
6 (lookups between tables), 9 (tables with their own constraint programs), 10 (the machine: programs + interaction tables) and 11 (the
keyed machine: some tables have preprocessed columns, committed once; the commitment is an argument of the verifier).
Written from the protocol description in DESIGN.md sections 3, 3b and 6 on top of tests/pyref.py and the helpers of
tests/pyverify.py; it shares no code with oracle/chips.c or with the product's host verifier.  Test infrastructure only.
"""
import struct

import pyref
from pyref import P, bitrev, ext_inv, ext_mul, ext_pow, two_adic_generator
from pyverify import GEN, MAGIC, ONE, ZERO, Reject, Transcript, air_digest, air_fold, e_add, e_base, e_from_columns, e_scale, e_sub

LKUP_MAGIC = 0x50554B4C


def parse_table(table, width):
    """-> [(sign, mult column or None, bus, [value columns])] of an interaction table"""
    t = [int(x) for x in table]
    if len(t) < 3 or t[0] != LKUP_MAGIC or not 1 <= t[1] <= 64 or t[2] != len(t):
        raise Reject("interaction table")
    out, p = [], 3
    for _ in range(t[1]):
        sign, mult, bus, nv = t[p:p + 4]
        cols = t[p + 4:p + 4 + nv]
        p += 4 + nv
        if sign > 1 or bus >= P or not 1 <= nv <= 8 or len(cols) != nv or any(c >= width for c in cols) or (mult != 0xFFFFFFFF and mult >= width):
            raise Reject("interaction table")
        out.append((sign, None if mult == 0xFFFFFFFF else mult, bus, cols))
    if p != len(t):
        raise Reject("interaction table")
    return out


def program_degree(program):
    """largest degree of a term of a constraint program, its selector counted"""
    w = [int(x) for x in program]
    pos, deg = 6, 0
    for _ in range(w[3]):
        sel, nt = w[pos], w[pos + 1]
        pos += 2
        for _ in range(nt):
            d = w[pos + 1]
            deg = max(deg, d + (1 if sel else 0))
            pos += 2 + d
    return deg


def mixed_root(rows_by_height, h_max, index, siblings):
    """Merkle root of a mixed-height commitment from one opening: the rows of the tallest matrices (concatenated in chip order)
    form the leaf; after the node with 2^h siblings below it has been formed, the rows of the matrices of height 2^h join as
    compress(node, sponge(rows))"""
    cur = pyref.sponge_hash(rows_by_height.get(h_max, []))
    for lvl in range(h_max):
        sib = siblings[lvl]
        cur = pyref.compress(sib, cur) if (index >> lvl) & 1 else pyref.compress(cur, sib)
        h = h_max - lvl - 1
        if rows_by_height.get(h):
            cur = pyref.compress(cur, pyref.sponge_hash(rows_by_height[h]))
    return cur


def verify(proof_bytes, log_ns, widths, public_values, log_blowup=1, num_queries=100, pow_bits=16, pairs=None, partners=None,
           programs=None, tables=None, pre_widths=None, pre_root=None, view=None):
    """raises Reject(reason) or returns True.  pairs / partners: versions 5 / 6; programs: version 9; programs + tables (a list,
    entries may be None): version 10; with pre_widths (per chip, 0: none) and pre_root (the key's 8 words): version 11, where
    programs and tables address the combined row [preprocessed | main].
    view: a dict that receives everything a verifier INSIDE a proof needs (tests/recursion_machine.py): roots, challenges, opened values,
    cumulative sums, and per query the opened rows with their paths and the FRI siblings with theirs."""
    if len(proof_bytes) % 4:
        raise Reject("length")
    w = list(struct.unpack("<%dI" % (len(proof_bytes) // 4), bytes(proof_bytes)))
    n, b = len(log_ns), log_blowup
    machine = tables is not None
    keyed = pre_widths is not None
    if keyed and (not machine or pre_root is None or len(pre_root) != 8 or any(v >= P for v in pre_root) or not any(pre_widths)):
        raise Reject("key")
    pws = [int(x) for x in pre_widths] if keyed else [0] * n
    if any(pw % 4 or pw + widths[c] > 1024 or (pw and (programs or [None] * n)[c] is None) for c, pw in enumerate(pws)):
        raise Reject("preprocessed widths")
    programs = programs or [None] * n
    inter = [parse_table(t, pws[c] + widths[c]) if (machine and t is not None) else None for c, t in enumerate(tables or [None] * n)]
    if machine:
        cols = [(len(it) + 1) // 2 if it else 0 for it in inter]
    else:
        cols = [int(x) for x in (pairs or [0] * n)]
    lk = any(cols)
    cross = (machine and lk) or (partners is not None and any(p_ >= 0 for p_ in partners))
    any_prog = any(p_ is not None for p_ in programs)
    # a program of degree 4 or 5 (a selector counts one degree) has four quotient chunks instead of two: its log_quotient_degree is what
    # the header's has-program word carries
    lqs = [(2 if program_degree(p_) > 3 else 1) if p_ is not None else 1 for p_ in programs]
    if any(q > b for q in lqs):
        raise Reject("a program of degree 4 or 5 needs log_blowup >= 2")
    qws = [4 << q for q in lqs]
    version = (11 if keyed else 10) if machine else (9 if any_prog else (6 if cross else (5 if lk else 4)))
    wp = [4 * (q + 1) if q else 0 for q in cols]
    lh = [ln + b for ln in log_ns]
    h_max, L = lh[0], log_ns[0]
    if any(log_ns[c] > log_ns[c - 1] for c in range(1, n)):
        raise Reject("tallest first")

    # ---- header and transcript start: version, chip count, parameters, the chip entries, then program and table digests
    head = [MAGIC, version, n, b, num_queries, pow_bits, len(public_values), 16]
    entries = []
    for c in range(n):
        entries += [log_ns[c], widths[c]]
        if machine:
            entries += [lqs[c] if programs[c] is not None else 0, len(inter[c]) if inter[c] else 0]
            if keyed:
                entries.append(pws[c])
            continue
        if lk:
            entries.append(cols[c])
        if cross:
            entries.append(partners[c] + 1)
        if any_prog:
            entries.append(lqs[c] if programs[c] is not None else 0)
    digests = []
    for c in range(n):
        if programs[c] is not None:
            digests += air_digest(programs[c])
    if machine:
        for c in range(n):
            if inter[c]:
                digests += air_digest(tables[c])
    if keyed:
        digests += [int(v) for v in pre_root]            # the key's commitment sits after the digests and is observed with them
    pos = len(head) + len(entries) + len(digests)
    if w[:pos] != head + entries + digests:
        raise Reject("header")
    if any(v >= P for v in w[pos:]) or any(v >= P for v in public_values):
        raise Reject("non-canonical word")

    def take(k):
        nonlocal pos
        out = w[pos:pos + k]
        if len(out) != k:
            raise Reject("truncated")
        pos += k
        return out

    def take_ext(k):
        flat = take(4 * k)
        return [flat[4 * i:4 * i + 4] for i in range(k)]

    ts = Transcript()
    ts.observe_many(head[1:7])                 # version .. n_public (not the trailing 16)
    ts.observe_many(entries)
    ts.observe_many(digests)
    trace_root = take(8)
    ts.observe_many(trace_root)
    ts.observe_many(public_values)
    gamma = beta_l = perm_root = None
    cumsum = [ZERO] * n
    if lk:
        gamma, beta_l = ts.sample_ext(), ts.sample_ext()
        perm_root = take(8)
        ts.observe_many(perm_root)
        if cross:
            total = ZERO
            for c in range(n):
                if wp[c]:
                    cumsum[c] = take(4)
                    ts.observe_many(cumsum[c])
                    total = e_add(total, cumsum[c])
            if total != ZERO:
                raise Reject("the lookups of the shard do not balance")
    alpha = ts.sample_ext()
    quot_root = take(8)
    ts.observe_many(quot_root)
    zeta = ts.sample_ext()
    opened = []
    for c in range(n):
        pre_parts = (take_ext(pws[c]), take_ext(pws[c]))                      # preprocessed columns at zeta, at zeta g: first
        opened.append((take_ext(widths[c]), take_ext(widths[c]), take_ext(wp[c]), take_ext(wp[c]), take_ext(qws[c])) + pre_parts)
    for c in range(n):
        for part in opened[c][5:] + opened[c][:5]:
            for e in part:
                ts.observe_many(e)

    # ---- (a) every chip's AIR identity at zeta (same alpha for every chip; its own trace domain)
    zeta_next = []
    for c in range(n):
        loc, nxt, pl, pn, qz, pre_l, pre_n = opened[c]
        loc, nxt = pre_l + loc, pre_n + nxt             # what programs and interactions address: [preprocessed | main]
        N = 1 << log_ns[c]
        wN = two_adic_generator(log_ns[c])
        wN_inv = pow(wN, -1, P)
        zeta_next.append(e_scale(zeta, wN))
        zeta_n = ext_pow(zeta, N)
        zh = e_sub(zeta_n, ONE)
        sel_first = ext_mul(zh, ext_inv(e_sub(zeta, ONE)))
        sel_last = ext_mul(zh, ext_inv(e_sub(zeta, e_base(wN_inv))))
        sel_trans = e_sub(zeta, e_base(wN_inv))
        acc = ZERO

        def fold(cst):
            nonlocal acc
            acc = e_add(ext_mul(acc, alpha), cst)
        if programs[c] is not None:
            acc = air_fold(programs[c], loc, nxt, public_values, sel_first, sel_last, sel_trans, alpha)
        else:
            for g in range(widths[c] // 4):             # (a chip without a program has no preprocessed columns)
                a, bb, cc, d, dn = loc[4 * g], loc[4 * g + 1], loc[4 * g + 2], loc[4 * g + 3], nxt[4 * g + 3]
                fold(e_sub(e_sub(cc, ext_mul(ext_mul(a, a), bb)), e_base(g + 1)))
                fold(ext_mul(sel_trans, e_sub(e_sub(e_sub(dn, ext_mul(a, bb)), cc), e_base(2 * g + 3))))
                fold(ext_mul(sel_first, e_sub(d, e_base(5 * g + 7))))
        if wp[c]:
            Q = cols[c]
            phis = [e_from_columns(pl[4 * q:4 * q + 4]) for q in range(Q + 1)]
            phins = [e_from_columns(pn[4 * q:4 * q + 4]) for q in range(Q + 1)]
            if machine:
                bpow = [ONE]
                for _ in range(9):
                    bpow.append(ext_mul(bpow[-1], beta_l))

                def finger(it):
                    d = e_add(gamma, e_base(it[2]))
                    for t, col in enumerate(it[3]):
                        d = e_add(d, ext_mul(bpow[t + 1], loc[col]))
                    return d

                def mult(it):
                    m = ONE if it[1] is None else loc[it[1]]
                    return e_sub(ZERO, m) if it[0] else m
                its = inter[c]
                for j in range(Q):
                    da, ma = finger(its[2 * j]), mult(its[2 * j])
                    if 2 * j + 1 < len(its):
                        db, mb = finger(its[2 * j + 1]), mult(its[2 * j + 1])
                        fold(e_sub(ext_mul(ext_mul(phis[j], da), db), e_add(ext_mul(ma, db), ext_mul(mb, da))))
                    else:
                        fold(e_sub(ext_mul(phis[j], da), ma))
            else:
                for q in range(Q):
                    den_s = e_add(e_add(gamma, loc[8 * q]), ext_mul(beta_l, loc[8 * q + 1]))
                    den_r = e_add(e_add(gamma, loc[8 * q + 4]), ext_mul(beta_l, loc[8 * q + 5]))
                    fold(e_sub(ext_mul(ext_mul(phis[q], den_s), den_r), e_sub(den_r, den_s)))
            sum_l = sum_n = ZERO
            for q in range(Q):
                sum_l, sum_n = e_add(sum_l, phis[q]), e_add(sum_n, phins[q])
            fold(ext_mul(sel_first, e_sub(phis[Q], sum_l)))
            fold(ext_mul(sel_trans, e_sub(e_sub(phins[Q], phis[Q]), sum_n)))
            fold(ext_mul(sel_last, e_sub(phis[Q], cumsum[c])))
        nq = 1 << lqs[c]
        wq = two_adic_generator(log_ns[c] + lqs[c])
        sN = [pow(GEN * pow(wq, k, P) % P, N, P) for k in range(nq)]
        quotient = ZERO
        for k in range(nq):                              # chunk k lives on the coset s_k <w_N>; zps_k vanishes on the other chunks' cosets
            zps = ONE
            for j in range(nq):
                if j == k:
                    continue
                sjn_inv = pow(sN[j], -1, P)
                num = e_sub(e_scale(zeta_n, sjn_inv), ONE)
                den = (sN[k] * sjn_inv - 1) % P
                zps = ext_mul(zps, e_scale(num, pow(den, -1, P)))
            quotient = e_add(quotient, ext_mul(zps, e_from_columns(qz[4 * k:4 * k + 4])))
        if ext_mul(acc, ext_inv(zh)) != quotient:
            raise Reject("chip %d: constraints do not match the quotient at zeta" % c)

    # ---- (b) FRI: one reduced-opening vector per height; the batching powers run on across the chips of a height
    fa = ts.sample_ext()
    npow = max(qws + list(widths) + wp + pws)
    fap = [ONE]
    for _ in range(npow - 1):
        fap.append(ext_mul(fap[-1], fa))

    def batch(values):
        t = ZERO
        for j, v in enumerate(values):
            t = e_add(t, ext_mul(fap[j], v))
        return t

    def batch_base(row):
        t = ZERO
        for j, v in enumerate(row):
            t = e_add(t, e_scale(fap[j], v))
        return t
    ys, offs = [], []
    for c in range(n):
        ys.append([batch(part) for part in opened[c]])
        off0 = sum(2 * pws[d] + 2 * widths[d] + 2 * wp[d] + qws[d] for d in range(c) if log_ns[d] == log_ns[c])
        W, Wp, Pw = widths[c], wp[c], pws[c]
        off = off0 + 2 * Pw                              # the chip's preprocessed columns take the first 2 Pw powers of its stretch
        offs.append([ext_pow(fa, off), ext_pow(fa, off + W), ext_pow(fa, off + 2 * W), ext_pow(fa, off + 2 * W + Wp), ext_pow(fa, off + 2 * W + 2 * Wp),
                     ext_pow(fa, off0), ext_pow(fa, off0 + Pw)])
    layer_roots, betas = [], []
    for _ in range(L):
        r = take(8)
        ts.observe_many(r)
        layer_roots.append(r)
        betas.append(ts.sample_ext())
    final = take(4)
    ts.observe_many(final)
    witness = take(1)[0]
    ts.observe(witness)
    if ts.sample_bits(pow_bits) != 0:
        raise Reject("proof of work")

    perm_chips = [c for c in range(n) if wp[c]]
    h_perm = max([lh[c] for c in perm_chips], default=0)
    pre_chips = [c for c in range(n) if pws[c]]
    h_pre = max([lh[c] for c in pre_chips], default=0)
    half = pow(2, -1, P)
    for _ in range(num_queries):
        index = ts.sample_bits(h_max)
        erows = {}
        if keyed:
            for c in pre_chips:
                erows[c] = take(pws[c])
            epath = [take(8) for _ in range(h_pre)]
        trows = [take(widths[c]) for c in range(n)]
        tpath = [take(8) for _ in range(h_max)]
        prows = {}
        if lk:
            for c in perm_chips:
                prows[c] = take(wp[c])
            ppath = [take(8) for _ in range(h_perm)]
        qrows = [take(qws[c]) for c in range(n)]
        qpath = [take(8) for _ in range(h_max)]

        def by_height(rows, chips):
            out = {}
            for c in chips:
                out.setdefault(lh[c], []).extend(rows[c])
            return out
        if keyed and mixed_root(by_height(erows, pre_chips), h_pre, index >> (h_max - h_pre), epath) != [int(v) for v in pre_root]:
            raise Reject("preprocessed opening")
        if mixed_root(by_height(trows, range(n)), h_max, index, tpath) != trace_root:
            raise Reject("trace opening")
        if lk and mixed_root(by_height(prows, perm_chips), h_perm, index >> (h_max - h_perm), ppath) != perm_root:
            raise Reject("permutation opening")
        if mixed_root(by_height(qrows, range(n)), h_max, index, qpath) != quot_root:
            raise Reject("quotient opening")
        roh = {}
        for c in range(n):
            ic = index >> (h_max - lh[c])
            x = GEN * pow(two_adic_generator(lh[c]), bitrev(ic, lh[c]), P) % P
            inv1 = ext_inv(e_sub(e_base(x), zeta))
            inv2 = ext_inv(e_sub(e_base(x), zeta_next[c]))
            at, aq = batch_base(trows[c]), batch_base(qrows[c])
            y_loc, y_nxt, y_pl, y_pn, y_q, y_el, y_en = ys[c]
            r = ext_mul(offs[c][0], ext_mul(e_sub(at, y_loc), inv1))
            r = e_add(r, ext_mul(offs[c][1], ext_mul(e_sub(at, y_nxt), inv2)))
            if wp[c]:
                ap = batch_base(prows[c])
                r = e_add(r, ext_mul(offs[c][2], ext_mul(e_sub(ap, y_pl), inv1)))
                r = e_add(r, ext_mul(offs[c][3], ext_mul(e_sub(ap, y_pn), inv2)))
            r = e_add(r, ext_mul(offs[c][4], ext_mul(e_sub(aq, y_q), inv1)))
            if pws[c]:
                ae = batch_base(erows[c])
                r = e_add(r, ext_mul(offs[c][5], ext_mul(e_sub(ae, y_el), inv1)))
                r = e_add(r, ext_mul(offs[c][6], ext_mul(e_sub(ae, y_en), inv2)))
            roh[lh[c]] = e_add(roh.get(lh[c], ZERO), r)
        val, idx = roh.get(h_max, ZERO), index
        qview = {"index": index, "erows": {c: list(v) for c, v in erows.items()}, "epath": [list(d) for d in epath] if keyed else [], "trows": [list(r) for r in trows],
                 "tpath": [list(d) for d in tpath], "prows": {c: list(v) for c, v in prows.items()}, "ppath": [list(d) for d in ppath] if lk else [],
                 "qrows": [list(r) for r in qrows], "qpath": [list(d) for d in qpath], "roh": {h: list(v) for h, v in roh.items()}, "sibs": [], "paths": []}
        for l in range(L):
            rows_log = h_max - 1 - l
            sib = take(4)
            path = [take(8) for _ in range(rows_log)]
            qview["sibs"].append(list(sib)), qview["paths"].append([list(d) for d in path])
            pair = [None, None]
            pair[idx & 1], pair[(idx & 1) ^ 1] = val, sib
            cur = pyref.sponge_hash(pair[0] + pair[1])
            row = idx >> 1
            for lvl, s in enumerate(path):
                cur = pyref.compress(s, cur) if (row >> lvl) & 1 else pyref.compress(cur, s)
            if cur != layer_roots[l]:
                raise Reject("FRI layer %d opening" % l)
            x = pow(two_adic_generator(rows_log + 1), bitrev(row, rows_log), P)
            even = e_scale(e_add(pair[0], pair[1]), half)
            odd = e_scale(e_sub(pair[0], pair[1]), half * pow(x, -1, P) % P)
            val = e_add(e_add(even, ext_mul(betas[l], odd)), roh.get(rows_log, ZERO))      # the vector of the height just reached joins
            idx = row
        if val != final:
            raise Reject("final value")
        if view is not None:
            view.setdefault("queries", []).append(qview)
    if pos != len(w):
        raise Reject("trailing words")
    if view is not None:
        view.update({"head": head, "entries": entries, "digests": digests, "trace_root": list(trace_root), "perm_root": list(perm_root) if lk else None,
                     "quot_root": list(quot_root), "cumsum": [list(c) for c in cumsum], "gamma": gamma, "beta": beta_l, "alpha": list(alpha), "zeta": list(zeta),
                     "fa": list(fa), "opened": opened, "layer_roots": layer_roots, "betas": [list(b_) for b_ in betas], "final": list(final), "witness": witness,
                     "wp": wp, "pws": pws, "lh": lh, "cols": cols, "inter": inter})
    return True
