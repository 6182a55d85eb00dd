"""MACHINE MODE of the shard verifier machine, written in Python FIRST (docs/RECURSION_NEXT.md, "machine mode"): a keyed machine that checks
WHOLE KEYED-MACHINE PROOFS (version 11: several chips of mixed heights, each with its constraint program, its interaction table and --
most of them -- preprocessed columns committed by a key) in-circuit.  With it the join's own output is joinable: a TREE of joins
(crates/guest-prover-sp1/src/sp1.rs:116 core -> compress; RISC Zero lift -> join, prover.rs:90).

Status: the executable design.  The product's shard_verifier.inl does not have this mode yet; this file is what its C++ is to be written
from, as tests/recursion_air.py's air mode was written after the C++.  Everything here is checked the way the other restatements are:
the programs hold row by row in plain integers on the witness of oracle-made version-11 proofs, every bus balances, the oracle's generic
keyed-machine prover proves the machine and tests/pyverify_chips.py accepts that proof (tests/test_recursion_machine_cpu.py).

What is restated is tests/pyverify_chips.py's verify(..., pre_widths=, pre_root=): the inner proofs are proofs of ONE machine
(the same chips, heights, programs, tables and key), `n_proofs` of them per outer proof.

Chips (the version-1 / version-7 machine's, changed where the comments say; tests/recursion_air.py has the originals):
  P2R      + injection rows (a shorter matrix's row digest joins a path), digests sent from where a row's sponge ends to where it is
           injected, four commitments per query (key, main, permutation, quotient) with the rows of one height concatenated, a second
           sampled challenge per row (gamma and beta come from one permutation)
  TS       + the long constant header (entries, digests, the inner key's root), the permutation root, the cumulative sums
  OPENED   one row per 8-word block of the opened-value stream: the batching power runs down the stream (the exponents of a height's
           reduced opening are consecutive in transcript order), the values go out under preprocessed keys
  ROWSUM   one row per sponge block of an opened row; a block holds up to two SEGMENTS (matrices), Horner restarts between them; a
           segment's sum is weighted with its two powers of the batching challenge and added to its height's running sums
  QUERY    one row per (query, height): the point, two inversions, the height's reduced opening -> the fold chain
  FOLD     + the reduced opening of the height just reached joins the folded value
  SCALARS  one row per (proof, chip): zeta^N of the chip's own N, selectors, the chip's AIR identity; the cumulative sums add up to zero
  EVAL     the programs of all chips, one row per term (air mode's chip, per chip its own selectors)
  LOGUP    one row per pair of interactions of a chip: the two fingerprints, phi d_a d_b = m_a d_b + m_b d_a, the running-sum constraints
  SAMPLES  unchanged"""
import numpy as np

import fri_air as F
import oracle_lib as O
import poseidon2_air as P2
import pyref
import pyverify_chips
from pyref import P, ext_mul, two_adic_generator
from recursion_air import (Cons, Cols, pc, pv, padd, pscale, pneg, pmul, ev, ec, eb, eadd, esub, escale, emul, egate, rup4, lg, _e4, e_add, e_sub,
                           SEND, RECV, GEN, EXT_W, BUS_IN0, BUS_IN1, BUS_TC, BUS_BETA, BUS_SC, BUS_QI, BUS_FIN, BUS_VAL, BUS_EA, BUS_OA, M_KP, _p2row)

V = O.air_var
# buses of this mode (the others keep their numbers from tests/recursion_air.py / tests/fri_air.py)
BUS_DG0, BUS_DG1 = 90, 91            # a row digest from the last sponge row of a shorter matrix's rows to the injection row of its path
BUS_TC2 = 92                         # the second challenge of a sponge row (its output words 3 .. 0)
BUS_PW = 93                          # (key, fa^e): the power of the batching challenge a segment's sum is weighted with, from the opened-value stream to ROWSUM
BUS_AH = 94                          # (query, height, Az, An): a height's weighted row sums from ROWSUM to QUERY
BUS_YH = 95                          # (proof, height, Yz, Yn): the opened values' sums from OPENED to QUERY
BUS_XH = 96                          # (query number x layers + layer, XS): the fold chain's point at a layer to the QUERY row of the height reached there
BUS_RO = 97                          # ... and the reduced opening back
BUS_CS = 98                          # (proof x chips + chip, cumulative sum) from TS to LOGUP's last row of the chip and to SCALARS
BUS_QZ = 99                          # (key, quotient value) from OPENED to SCALARS
BUS_ACC = 100                        # (proof x chips + chip, fold so far) EVAL -> LOGUP -> SCALARS


def parse_table(table, width):
    return pyverify_chips.parse_table(table, width)


class Plan:
    """the duplex sponge of tests/pyverify.py's Transcript, run on SOURCES instead of values: which sponge row absorbs what, and behind
    which row (and which half of its output) every challenge is sampled"""
    def __init__(self):
        self.rows = []                # per sponge row: the list of absorbed sources (0 .. 8 of them)
        self.pending = []
        self.ready = 0
        self.at = {}                  # challenge name -> (row, half): half 0 = output words 7 .. 4, half 1 = words 3 .. 0

    def _duplex(self):
        self.rows.append(self.pending)
        self.pending, self.ready = [], 8

    def observe(self, sources):
        for s in sources:
            self.ready = 0
            self.pending.append(s)
            if len(self.pending) == 8:
                self._duplex()

    def sample(self, name):
        if self.pending or self.ready < 4:
            self._duplex()
        self.at[name] = (len(self.rows) - 1, 0 if self.ready == 8 else 1)
        self.ready -= 4


KINDS = ("el", "en", "tl", "tn", "pl", "pn", "q")           # a chip's opened values, in proof / transcript order
TREES = ("E", "T", "P", "Q")                                # the key's tree (preprocessed), main, permutation, quotient


class MShape:
    """the inner MACHINE (its chips: log_n, main width, preprocessed width, program, interaction table; its key's root; its public values)
    and how its proofs are made (queries, proof-of-work bits; blowup 2, fold by 2, constant final value), times n_proofs"""
    def __init__(self, chips, key_root, n_queries, pow_bits, n_public, n_proofs=1):
        self.chips = chips
        self.C = C = len(chips)
        self.Q, self.PB, self.NPUB, self.NP = n_queries, pow_bits, n_public, n_proofs
        self.key_root = [int(v) for v in key_root]
        self.ln = [int(c["ln"]) for c in chips]
        self.W = [int(c["W"]) for c in chips]
        self.Pw = [int(c["Pw"]) for c in chips]
        assert all(self.ln[i] >= self.ln[i + 1] for i in range(C - 1)) and all(w % 4 == 0 for w in self.W + self.Pw) and 1 <= C <= 16
        self.inter = [parse_table(c["tab"], c["Pw"] + c["W"]) for c in chips]
        assert all(len(it) >= 1 for it in self.inter), "every chip of a machine talks to another"
        self.cols = [(len(it) + 1) // 2 for it in self.inter]
        self.Wp = [4 * (q + 1) for q in self.cols]
        self.lh = [l + 1 for l in self.ln]
        self.H, self.R = self.lh[0], self.ln[0]
        self.hs = sorted(set(self.lh), reverse=True)                            # the heights (log2 of the LDE rows), tallest first
        # ---- the header: what the transcript starts from (tests/pyverify_chips.py): six words, the chips' entries, the digests, the key's root
        head = [11, C, 1, n_queries, pow_bits, n_public]
        for c in range(C):
            head += [self.ln[c], self.W[c], 1, len(self.inter[c]), self.Pw[c]]
        for c in chips:
            head += [int(x) for x in O.air_digest(np.asarray(c["prog"], dtype=np.uint32))]
        for c in chips:
            head += [int(x) for x in O.air_digest(np.asarray(c["tab"], dtype=np.uint32))]
        self.head = head + self.key_root
        self.HL = len(self.head)
        # ---- the opened-value stream: per chip el | en | tl | tn | pl | pn | q, extension values; the batching exponent runs on across the chips
        # of one height (tallest first: they are adjacent) and restarts where the height changes
        self.segs = []                                                          # (chip, kind, first stream position, length, first exponent), stream order
        pos, e = 0, 0
        for c in range(C):
            if c and self.lh[c] != self.lh[c - 1]:
                e = 0
            for kind, n in zip(KINDS, (self.Pw[c], self.Pw[c], self.W[c], self.W[c], self.Wp[c], self.Wp[c], 8)):
                self.segs.append((c, kind, pos, n, e))
                pos, e = pos + n, e + n
        self.NV = pos                                                           # opened extension values per proof (even: every length is a multiple of 4)
        self.seg_at = {(c, kind): (pos_, n, e_) for c, kind, pos_, n, e_ in self.segs}
        # ---- the transcript
        pl = Plan()
        pl.observe([("c", v) for v in self.head] + [("troot", j) for j in range(8)] + [("pub", i) for i in range(n_public)])
        pl.sample("gamma"), pl.sample("beta")
        self.TG = pl.at["gamma"][0]
        assert pl.at["gamma"] == (self.TG, 0) and pl.at["beta"] == (self.TG, 1)
        pl.observe([("proot", j) for j in range(8)] + [("cum", c, j) for c in range(C) for j in range(4)])
        pl.sample("alpha")
        self.TPR, self.TA = self.TG + 1, pl.at["alpha"][0]
        pl.observe([("qroot", j) for j in range(8)])
        pl.sample("zeta")
        self.TQ = pl.at["zeta"][0]
        pl.observe([("op", i) for i in range(4 * self.NV)])
        pl.sample("fa")
        self.TO0, self.TF = self.TQ + 1, pl.at["fa"][0]
        assert self.TF - self.TO0 + 1 == self.NV // 2 and all(h == 0 for name, (_, h) in pl.at.items() if name != "beta")
        for l in range(self.R):
            pl.observe([("lroot", l, j) for j in range(8)])
            pl.sample(("beta", l))
        self.TL0 = self.TF + 1
        pl.observe([("fin", j) for j in range(4)] + [("wit",)])
        pl.sample("pow")
        self.TP = pl.at["pow"][0]
        assert self.TP == self.TL0 + self.R
        self.plan = pl.rows                                                     # rows 0 .. TP: what each absorbs
        self.NS = F.sample_rows(n_queries)
        self.NT, self.NTS = self.TP + self.NS, self.TP + 1
        self.f0 = (self.HL + 8) // 8                                             # (rows before the one the trace root ends in)
        self.pub_rows = sorted({(self.HL + 8 + i) // 8 for i in range(n_public)})
        # ---- the four commitments: which chips, which heights
        self.tree_chips = {"E": [c for c in range(C) if self.Pw[c]], "T": list(range(C)), "P": list(range(C)), "Q": list(range(C))}
        self.tree_w = {"E": self.Pw, "T": self.W, "P": self.Wp, "Q": [8] * C}
        self.trees = [t for t in TREES if self.tree_chips[t]]
        self.tree_hs = {t: sorted({self.lh[c] for c in self.tree_chips[t]}, reverse=True) for t in self.trees}
        # a height's rows in a tree = the rows of its chips, concatenated in chip order: (chip, first word, width) per segment
        self.leaf_segs = {}
        for t in self.trees:
            for h in self.tree_hs[t]:
                at, lst = 0, []
                for c in self.tree_chips[t]:
                    if self.lh[c] == h:
                        lst.append((c, at, self.tree_w[t][c]))
                        at += self.tree_w[t][c]
                self.leaf_segs[(t, h)] = (lst, at)
        # ---- P2R row layout (per proof): transcript | FRI layers per query | per tree, per query: the shorter heights' sponges, the tallest's, the path
        self.fri_rows = self.R + self.R * (self.R + 1) // 2
        self.p2_fri0 = self.NT
        at = self.p2_fri0 + self.Q * self.fri_rows
        self.p2_tree0, self.tree_rows = {}, {}
        for t in self.trees:
            hs = self.tree_hs[t]
            per = sum((self.leaf_segs[(t, h)][1] + 7) // 8 for h in hs) + hs[0] + (len(hs) - 1)
            self.p2_tree0[t], self.tree_rows[t] = at, per
            at += self.Q * per
        self.p2_rows = at
        # tags: transcript rows, then the leaf blocks of every (query, tree, height, block)
        self.blk0 = {}
        n = self.NT
        for t in self.trees:
            for h in self.tree_hs[t]:
                self.blk0[(t, h)] = n
                n += (self.leaf_segs[(t, h)][1] + 7) // 8
        self.BLKSPAN = n - self.NT                                               # leaf blocks per query
        self.TAGSPAN = self.NT + self.Q * self.BLKSPAN
        self.NTREES = self.R + 4                                                 # trees of one proof on the root buses: the FRI layers, then E, T, P, Q
        self.KSPAN = 4 * self.NV + 64                                            # keys of one proof on the value bus: its opened values (one per stream position), then constants

    def ttag(self, p, T):
        return p * self.TAGSPAN + T

    def blk_tag(self, p, q, t, h, b):
        return p * self.TAGSPAN + self.NT + q * self.BLKSPAN + (self.blk0[(t, h)] - self.NT) + b

    def dg_tag(self, p, q, t, h):
        return ((p * self.Q + q) * 4 + TREES.index(t)) * 32 + h

    def tree_id(self, p, t):
        return p * self.NTREES + (t if isinstance(t, int) else self.R + TREES.index(t))

    def absorbed(self, T):
        return len(self.plan[T]) if T <= self.TP else 0


# ---------------------------------------------------------------------------------------------------------------- P2R
P2_PRE = 28
(PP_SS, PP_SPG, PP_CH, PP_END, PP_K, PP_RIN, PP_TAG, PP_SROOT, PP_TREE, PP_SCH, PP_SSMP, PP_QIDX, PP_QN, PP_RPAIR,
 PP_RIN1, PP_SDG, PP_RDG, PP_CHN, PP_SCH2, PP_DTAG, PP_HALF) = 0, 1, 2, 3, 4, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27
P2_MAIN = 360


def p2r_program(sh):
    M0 = P2_PRE
    cons = Cons()
    for sel, terms in P2.permutation_constraints():
        cons.add(sel, [(c, [v + M0 if (v >> 30) == 0 else v for v in vs]) for c, vs in terms])
    IN, OUT, D, BIT, KP = M0 + P2.IN, M0 + P2.OUTE(7), M0 + P2.D, M0 + P2.BIT, M0 + M_KP
    for j in range(8):
        cons.add(O.SEL_ALL, padd(pv(D + j), pneg(pv(IN + j)), pmul(pv(BIT), pv(IN + j)), pneg(pmul(pv(BIT), pv(IN + 8 + j)))))
    cons.add(O.SEL_ALL, padd(pmul(pv(BIT), pv(BIT)), pneg(pv(BIT))))
    cons.add(O.SEL_ALL, pmul(padd(pv(PP_SS), pv(PP_SPG), pv(PP_RDG)), pv(BIT)))   # leaf, transcript and injection rows: no direction
    for j in range(8):
        cons.add(O.SEL_ALL, pmul(pv(PP_SS), pv(IN + 8 + j)))
    for j in range(4):                                                          # a row of four words that starts a sponge: the other four rate words are zero
        cons.add(O.SEL_ALL, pmul(pv(PP_HALF), pv(IN + 4 + j)))
    for j in range(8):
        cons.add(O.SEL_TRANSITION, pmul(pv(PP_SPG, True), padd(pv(IN + 8 + j, True), pneg(pv(OUT + 8 + j)))))
    for j in range(8):                                                          # a path row or an injection row carries on from the previous row's digest
        cons.add(O.SEL_TRANSITION, pmul(padd(pv(PP_CH, True), pv(PP_CHN, True)), padd(pv(D + j, True), pneg(pv(OUT + j)))))
    for j in range(8):
        cons.add(O.SEL_TRANSITION, pmul(pv(PP_K + j, True), padd(pv(IN + j, True), pneg(pv(OUT + j)))))
    cons.add(O.SEL_TRANSITION, pmul(pv(PP_CH, True), padd(pv(KP), pscale(pv(KP, True), P - 2), pneg(pv(BIT)))))
    cons.add(O.SEL_TRANSITION, pmul(pv(PP_CHN, True), padd(pv(KP), pneg(pv(KP, True)))))      # the row behind an injection: the same level's index
    cons.add(O.SEL_ALL, pmul(pv(PP_END), padd(pv(KP), pneg(pv(BIT)))))
    return O.air_program(P2_PRE + P2_MAIN, sh.NP * sh.NPUB, cons.c)


def p2r_table():
    M0, o = P2_PRE, P2_PRE + P2.OUTE(7)
    IN, KP = M0 + P2.IN, M0 + M_KP
    return O.interaction_table([
        (RECV, PP_RIN, BUS_IN0, [PP_TAG, IN, IN + 1, IN + 2, IN + 3]), (RECV, PP_RIN1, BUS_IN1, [PP_TAG, IN + 4, IN + 5, IN + 6, IN + 7]),
        (RECV, PP_RPAIR, F.BUS_E0, [PP_TREE, KP, IN, IN + 1, IN + 2, IN + 3]), (RECV, PP_RPAIR, F.BUS_E1, [PP_TREE, KP, IN + 4, IN + 5, IN + 6, IN + 7]),
        (SEND, PP_SROOT, F.BUS_R0, [PP_TREE, o, o + 1, o + 2, o + 3]), (SEND, PP_SROOT, F.BUS_R1, [PP_TREE, o + 4, o + 5, o + 6, o + 7]),
        (SEND, PP_SCH, BUS_TC, [PP_TAG, o + 7, o + 6, o + 5, o + 4]), (SEND, PP_SCH2, BUS_TC2, [PP_TAG, o + 3, o + 2, o + 1, o]),
        (SEND, PP_SSMP, F.BUS_S0, [PP_TAG, o + 7, o + 6, o + 5, o + 4]), (SEND, PP_SSMP, F.BUS_S1, [PP_TAG, o + 3, o + 2, o + 1, o]),
        (RECV, PP_QIDX, BUS_QI, [PP_QN, KP]),
        (SEND, PP_SDG, BUS_DG0, [PP_DTAG, o, o + 1, o + 2, o + 3]), (SEND, PP_SDG, BUS_DG1, [PP_DTAG, o + 4, o + 5, o + 6, o + 7]),
        (RECV, PP_RDG, BUS_DG0, [PP_DTAG, IN + 8, IN + 9, IN + 10, IN + 11]), (RECV, PP_RDG, BUS_DG1, [PP_DTAG, IN + 12, IN + 13, IN + 14, IN + 15])])


def p2r_pre(sh, log_rows):
    t = np.zeros((1 << log_rows, P2_PRE), dtype=np.uint32)
    for p in range(sh.NP):
        _p2r_pre_one(sh, t[p * sh.p2_rows:(p + 1) * sh.p2_rows], p)
    return t


def qn_of(sh, p, q, t):
    """the number under which a query's index for tree t travels from QUERY to the tree's first path row: the query and the tree's height"""
    return (p * sh.Q + q) * 32 + sh.tree_hs[t][0]


def _p2r_pre_one(sh, t, p):
    challenge_rows = {sh.TG, sh.TA, sh.TQ, sh.TF} | set(range(sh.TL0, sh.TP))
    for T in range(sh.NT):
        r = t[T]
        k = sh.absorbed(T)
        if T == 0:
            r[PP_SS] = 1
        else:
            r[PP_SPG] = 1
            for j in range(k, 8):
                r[PP_K + j] = 1
        r[PP_TAG] = sh.ttag(p, T)
        if k:
            r[PP_RIN] = r[PP_RIN1] = 1                                          # (the transcript table sends all eight words, the kept ones too)
        if T in challenge_rows:
            r[PP_SCH] = 1
        if T == sh.TG:
            r[PP_SCH2] = 1
        if T >= sh.TP:
            r[PP_SSMP] = 1
    row = sh.p2_fri0
    for q in range(sh.Q):
        for l in range(sh.R):
            t[row, PP_SS], t[row, PP_RPAIR], t[row, PP_TREE] = 1, 1, sh.tree_id(p, l)
            row += 1
            depth = sh.H - (l + 1)
            for lvl in range(depth):
                t[row, PP_CH], t[row, PP_TREE] = 1, sh.tree_id(p, l)
                if lvl == depth - 1:
                    t[row, PP_END] = t[row, PP_SROOT] = 1
                row += 1
    for tr in sh.trees:
        assert row == sh.p2_tree0[tr]
        hs = sh.tree_hs[tr]
        for q in range(sh.Q):
            for h in hs[1:] + hs[:1]:                                           # the shorter heights' rows first (their digests travel), the tallest's last
                words = sh.leaf_segs[(tr, h)][1]
                nb = (words + 7) // 8
                for b in range(nb):
                    r = t[row]
                    r[PP_SS if b == 0 else PP_SPG] = 1
                    k = min(8, words - 8 * b)
                    if b == 0 and k < 8:
                        r[PP_HALF] = 1
                    else:
                        for j in range(k, 8):
                            r[PP_K + j] = 1
                    r[PP_RIN], r[PP_RIN1], r[PP_TAG] = 1, 1 if k == 8 else 0, sh.blk_tag(p, q, tr, h, b)
                    if b == nb - 1 and h != hs[0]:
                        r[PP_SDG], r[PP_DTAG] = 1, sh.dg_tag(p, q, tr, h)
                    row += 1
            chained = False
            for lvl in range(hs[0]):
                r = t[row]
                r[PP_CHN if chained else PP_CH], r[PP_TREE] = 1, sh.tree_id(p, tr)
                chained = False
                if lvl == 0:
                    r[PP_QIDX], r[PP_QN] = 1, qn_of(sh, p, q, tr)
                if lvl == hs[0] - 1:
                    r[PP_END] = r[PP_SROOT] = 1
                row += 1
                h = hs[0] - lvl - 1
                if h in hs[1:]:
                    r = t[row]
                    r[PP_CH], r[PP_RDG], r[PP_DTAG], r[PP_TREE] = 1, 1, sh.dg_tag(p, q, tr, h), sh.tree_id(p, tr)
                    chained = True
                    row += 1
    assert row == sh.p2_rows


def leaf_words(sh, w, q, tr, h):
    """the concatenated rows of height h in tree tr at query q"""
    qv = w["queries"][q]
    rows = {"E": qv["erows"], "T": qv["trows"], "P": qv["prows"], "Q": qv["qrows"]}[tr]
    out = []
    for c, _, _ in sh.leaf_segs[(tr, h)][0]:
        out += list(rows[c])
    return out


def p2r_main(sh, ws, log_rows):
    rows, samples, chals = [], [], []
    for w in ws:
        r, s, c = _p2r_main_one(sh, w)
        rows += r
        samples.append(s)
        chals.append(c)
    pad, _ = _p2row([0] * 16)
    rows += [pad] * ((1 << log_rows) - len(rows))
    return np.array(rows, dtype=np.uint64).astype(np.uint32), samples, chals


def source_value(sh, w, src):
    kind = src[0]
    if kind == "c":
        return src[1]
    if kind == "op":
        return w["stream"][src[1]]
    if kind == "cum":
        return w["cumsum"][src[1]][src[2]]
    if kind == "lroot":
        return w["layer_roots"][src[1]][src[2]]
    if kind == "wit":
        return w["witness"]
    return {"troot": w["trace_root"], "pub": w["pubs"], "proot": w["perm_root"], "qroot": w["quot_root"], "fin": w["final"]}[kind][src[1]]


def _p2r_main_one(sh, w):
    rows, chal, samples = [], {}, []
    state = [0] * 16
    for T in range(sh.NT):
        blk = [source_value(sh, w, s) for s in sh.plan[T]] if T <= sh.TP else []
        k = len(blk)
        state = [blk[j] if j < k else state[j] for j in range(8)] + (state[8:] if T else [0] * 8)
        r, out = _p2row(state)
        rows.append(r)
        chal[T] = ([out[7], out[6], out[5], out[4]], [out[3], out[2], out[1], out[0]])
        if T >= sh.TP:
            samples.append([out[7 - j] for j in range(8)])
        state = list(out)
    for q in range(sh.Q):
        qv = w["queries"][q]
        idx, own = qv["index"], qv["roh"].get(sh.H, [0, 0, 0, 0])
        for l in range(sh.R):
            bit, k = idx & 1, idx >> 1
            sib = qv["sibs"][l]
            pair = (list(sib) + list(own)) if bit else (list(own) + list(sib))
            r, out = _p2row(pair + [0] * 8, 0, 2 * k)
            rows.append(r)
            digest = out[:8]
            for lvl, s in enumerate(qv["paths"][l]):
                b = (k >> lvl) & 1
                r, out = _p2row(list(s) + digest if b else digest + list(s), b, k >> lvl)
                rows.append(r)
                digest = out[:8]
            assert digest == w["layer_roots"][l], "a FRI layer path does not end in the layer's root"
            own = e_add(F.fold_pair(k, sh.H - (l + 1), w["betas"][l], pair[:4], pair[4:])[0], qv["roh"].get(sh.H - 1 - l, [0, 0, 0, 0]))
            idx = k
        assert list(own) == list(w["final"])
    roots = {"E": sh.key_root, "T": w["trace_root"], "P": w["perm_root"], "Q": w["quot_root"]}
    for tr in sh.trees:
        hs = sh.tree_hs[tr]
        for q in range(sh.Q):
            qv = w["queries"][q]
            index = qv["index"] >> (sh.H - hs[0])
            path = {"E": qv["epath"], "T": qv["tpath"], "P": qv["ppath"], "Q": qv["qpath"]}[tr]
            digests = {}
            for h in hs[1:] + hs[:1]:
                vals = leaf_words(sh, w, q, tr, h)
                nb = (len(vals) + 7) // 8
                state = [0] * 16
                for b in range(nb):
                    blk = vals[8 * b:8 * b + 8]
                    state = [blk[j] if j < len(blk) else state[j] for j in range(8)] + state[8:]
                    r, out = _p2row(state, 0, 2 * index if (b == nb - 1 and h == hs[0]) else 0)
                    rows.append(r)
                    state = list(out)
                digests[h] = out[:8]
            digest = digests[hs[0]]
            for lvl, sib in enumerate(path):
                bit = (index >> lvl) & 1
                r, out = _p2row(list(sib) + digest if bit else digest + list(sib), bit, index >> lvl)
                rows.append(r)
                digest = out[:8]
                h = hs[0] - lvl - 1
                if h in hs[1:]:
                    r, out = _p2row(digest + digests[h], 0, index >> (lvl + 1))
                    rows.append(r)
                    digest = out[:8]
            assert digest == list(roots[tr]), "an opening does not end in its root (%s)" % tr
    assert len(rows) == sh.p2_rows
    return rows, samples, chal


# ---------------------------------------------------------------------------------------------------------------- the witness
def witness(sh, proof, public_values, programs, tables, pre_root):
    view = {}
    pyverify_chips.verify(proof, sh.ln, sh.W, public_values, num_queries=sh.Q, pow_bits=sh.PB, programs=programs, tables=tables, pre_widths=sh.Pw, pre_root=pre_root, view=view)
    w = dict(view)
    w["pubs"] = [int(v) % P for v in public_values]
    stream = []
    for c in range(sh.C):
        loc, nxt, pl, pn, qz, pre_l, pre_n = view["opened"][c]
        for part in (pre_l, pre_n, loc, nxt, pl, pn, qz):
            for e in part:
                stream += [int(x) for x in e]
    w["stream"] = stream
    assert len(stream) == 4 * sh.NV
    return w


# ---------------------------------------------------------------------------------------------------------------- TS
# One row per absorbing sponge row of the transcript.  Against tests/recursion_air.py's: the header is long (entries, digests, the inner
# key's root: constants of the key), the permutation root and the cumulative sums are observed, gamma and beta are the two halves of ONE
# sponge output, and five challenges go to SCALARS (kinds alpha, zeta, fa, gamma, beta).
TS_MAIN = 24
N_CHAL = 5                                                                      # kinds on BUS_SC: 5 p + (0 alpha, 1 zeta, 2 fa, 3 gamma, 4 beta)


def ts_cols(sh):
    c = Cols()
    for name, w in (("T", 1), ("ACT", 1), ("NSEND", 1), ("CF", 8), ("CV", 8), ("IND0", 1), ("IP", sh.NP * len(sh.pub_rows)), ("NROOT", 1), ("NTR", 1), ("TREE", 1),
                    ("HASCH", 1), ("NBETA", 1), ("NSC", 1), ("KIND", 1), ("NFIN", 1), ("PT", 1), ("HASCH2", 1), ("NSC2", 1), ("KIND2", 1),
                    ("CK0", 1), ("CM0", 1), ("CK1", 1), ("CM1", 1), ("PK", 8), ("PM", 8), ("Z", 1)):
        c(name, w)
    pre = rup4(c.n)
    m = Cols(pre)
    m("W", 8), m("TR", 8), m("CH", 4), m("CH2", 4)
    return c, m, pre


def ts_program(sh):
    c, m, pre = ts_cols(sh)
    cons = Cons()
    W, TR = m["W"], m["TR"]
    for j in range(8):
        cons.add(O.SEL_ALL, pmul(pv(c["CF"] + j), padd(pv(W + j), pneg(pv(c["CV"] + j)))))
    npr = len(sh.pub_rows)
    for p in range(sh.NP):
        for i in range(sh.NPUB):
            pos = sh.HL + 8 + i
            cons.add(O.SEL_ALL, pmul(pv(c["IP"] + p * npr + sh.pub_rows.index(pos // 8)), padd(pv(W + pos % 8), [(P - 1, [V(p * sh.NPUB + i, public=True)])])))
    o = sh.HL % 8
    for j in range(8 - o):
        cons.add(O.SEL_ALL, pmul(pv(c["IND0"]), padd(pv(W + o + j), pneg(pv(TR + j)))))
    for j in range(o):
        cons.add(O.SEL_TRANSITION, pmul(pv(c["IND0"]), padd(pv(W + j, True), pneg(pv(TR + 8 - o + j)))))
    return O.air_program(pre + TS_MAIN, sh.NP * sh.NPUB, cons.c)


def ts_table(sh):
    c, m, _ = ts_cols(sh)
    W, TR, CH, CH2 = m["W"], m["TR"], m["CH"], m["CH2"]
    rows = [
        (SEND, c["NSEND"], BUS_IN0, [c["T"], W, W + 1, W + 2, W + 3]), (SEND, c["NSEND"], BUS_IN1, [c["T"], W + 4, W + 5, W + 6, W + 7]),
        (RECV, c["HASCH"], BUS_TC, [c["T"]] + _e4(CH)), (RECV, c["HASCH2"], BUS_TC2, [c["T"]] + _e4(CH2)),
        (SEND, c["NBETA"], BUS_BETA, [c["TREE"]] + _e4(CH)),
        (SEND, c["NSC"], BUS_SC, [c["KIND"]] + _e4(CH)), (SEND, c["NSC2"], BUS_SC, [c["KIND2"]] + _e4(CH2)),
        (RECV, c["NROOT"], F.BUS_R0, [c["TREE"], W, W + 1, W + 2, W + 3]), (RECV, c["NROOT"], F.BUS_R1, [c["TREE"], W + 4, W + 5, W + 6, W + 7]),
        (RECV, c["NTR"], F.BUS_R0, [c["TREE"]] + _e4(TR)), (RECV, c["NTR"], F.BUS_R1, [c["TREE"]] + _e4(TR + 4)),
        (RECV, c["NFIN"], BUS_FIN, [c["PT"], W, W + 1, W + 2, W + 3]),
        (SEND, c["CM0"], BUS_CS, [c["CK0"], W, W + 1, W + 2, W + 3]), (SEND, c["CM1"], BUS_CS, [c["CK1"], W + 4, W + 5, W + 6, W + 7])]
    rows += [(SEND, c["PM"] + j, BUS_VAL, [c["PK"] + j, W + j, c["Z"], c["Z"], c["Z"]]) for j in range(8)]
    return O.interaction_table(rows)


def ts_pre(sh, log_rows):
    c, _, pre = ts_cols(sh)
    t = np.zeros((1 << log_rows, pre), dtype=np.uint32)
    npr = len(sh.pub_rows)
    for p in range(sh.NP):
        for T in range(sh.NTS):
            r = t[p * sh.NTS + T]
            r[c["T"]], r[c["ACT"]], r[c["NSEND"]] = sh.ttag(p, T), 1, 2 if sh.TO0 <= T <= sh.TF else 1
            for j, src in enumerate(sh.plan[T]):
                if src[0] == "c":
                    r[c["CF"] + j], r[c["CV"] + j] = 1, src[1]
                if src[0] == "pub":
                    r[c["PK"] + j], r[c["PM"] + j] = p * sh.KSPAN + key_pub(sh, src[1]), sh.mult[key_pub(sh, src[1])]
                if src[0] == "cum" and src[2] == 0:
                    r[c["CK0" if j == 0 else "CK1"]], r[c["CM0" if j == 0 else "CM1"]] = p * sh.C + src[1], 2
            if T in sh.pub_rows:
                r[c["IP"] + p * npr + sh.pub_rows.index(T)] = 1
            if T == sh.HL // 8:
                r[c["IND0"]], r[c["NTR"]], r[c["TREE"]] = 1, sh.Q, sh.tree_id(p, "T")
            if T == sh.TPR:
                r[c["NROOT"]], r[c["TREE"]] = sh.Q, sh.tree_id(p, "P")
            if T == sh.TQ:
                r[c["NROOT"]], r[c["TREE"]] = sh.Q, sh.tree_id(p, "Q")
            if sh.TL0 <= T < sh.TP:
                r[c["NROOT"]], r[c["TREE"]], r[c["NBETA"]] = sh.Q, sh.tree_id(p, T - sh.TL0), sh.Q
            for kind, Tk in enumerate((sh.TA, sh.TQ, sh.TF, sh.TG)):
                if T == Tk:
                    r[c["NSC"]], r[c["KIND"]] = 1, N_CHAL * p + kind
            if T == sh.TG:
                r[c["HASCH2"]], r[c["NSC2"]], r[c["KIND2"]] = 1, 1, N_CHAL * p + 4
            if T in (sh.TG, sh.TA, sh.TQ, sh.TF) or sh.TL0 <= T < sh.TP:
                r[c["HASCH"]] = 1
            if T == sh.TP:
                r[c["NFIN"]], r[c["PT"]] = sh.Q, p * sh.NTREES
    return t


def ts_main(sh, ws, chals, p2_main, log_rows):
    t = np.zeros((1 << log_rows, TS_MAIN), dtype=np.uint32)
    for p, (w, chal) in enumerate(zip(ws, chals)):
        for T in range(sh.NTS):
            r = t[p * sh.NTS + T]
            r[0:8] = p2_main[p * sh.p2_rows + T, P2.IN:P2.IN + 8]
            if T in (sh.TG, sh.TA, sh.TQ, sh.TF) or sh.TL0 <= T < sh.TP:
                r[16:20] = chal[T][0]
            if T == sh.TG:
                r[20:24] = chal[T][1]
        t[p * sh.NTS + sh.HL // 8, 8:16] = w["trace_root"]
    return t


# ---------------------------------------------------------------------------------------------------------------- the value bus
# (key, extension value) from where a value of the inner proof lives to where a constraint reads it.  Keys of proof p: p KSPAN +
#   0                      the constant one (SCALARS)
#   1 + i                  the opened value at stream position i (OPENED)
#   1 + NV + i             public value i (TS)
#   1 + NV + NPUB + 3 c + w   selector w (0 first row, 1 last row, 2 transition) of chip c's trace domain (SCALARS)
def key_op(sh, c, kind, col):
    pos, n, _ = sh.seg_at[(c, kind)]
    assert 0 <= col < n
    return 1 + pos + col


def key_pub(sh, i):
    return 1 + sh.NV + i


def key_sel(sh, c, which):
    return 1 + sh.NV + sh.NPUB + 3 * c + which


def key_var(sh, c, v):
    """a variable of chip c's program (it addresses the combined row [preprocessed | main])"""
    kind, idx = v >> 30, v & 0xFFFF
    if kind == 2:
        return key_pub(sh, idx)
    if idx < sh.Pw[c]:
        return key_op(sh, c, "en" if kind else "el", idx)
    return key_op(sh, c, "tn" if kind else "tl", idx - sh.Pw[c])


def build_reads(sh):
    """the EVAL chip's terms and the LOGUP chip's rows, and with them how often every key is read (the senders' multiplicities)"""
    sh.KSPAN = 1 + sh.NV + sh.NPUB + 3 * sh.C
    sh.mult = [0] * sh.KSPAN
    sh.terms = []                                                               # (chip, coefficient, [three keys], first term of its constraint)
    for c, chip in enumerate(sh.chips):
        prog = [int(x) for x in chip["prog"]]
        at = 6
        for _ in range(prog[3]):
            sel, nt = prog[at], prog[at + 1]
            at += 2
            if nt == 0:
                sh.terms.append((c, 0, [0, 0, 0], 1))
            for t in range(nt):
                coeff, d = prog[at], prog[at + 1]
                at += 2
                keys = [key_var(sh, c, v) for v in prog[at:at + d]]
                at += d
                if sel:
                    keys.append(key_sel(sh, c, sel - 1))
                assert len(keys) <= 3, "log_quotient_degree 1"
                keys += [0] * (3 - len(keys))
                sh.terms.append((c, coeff, keys, 1 if t == 0 else 0))
    sh.term0 = [min(i for i, t in enumerate(sh.terms) if t[0] == c) for c in range(sh.C)]
    sh.term1 = [max(i for i, t in enumerate(sh.terms) if t[0] == c) for c in range(sh.C)]
    for _, _, keys, _ in sh.terms:
        for k in keys:
            sh.mult[k] += 1
    # LOGUP: per chip one row per pair of interactions, then a boundary row
    sh.lrows = []
    for c in range(sh.C):
        its = sh.inter[c]
        comb = lambda col: key_var(sh, c, col)          # noqa: E731 (a column of the combined row at zeta)
        for j in range(sh.cols[c]):
            row = {"chip": c, "kind": "pair", "j": j, "phi": [key_op(sh, c, "pl", 4 * j + k) for k in range(4)], "phin": [key_op(sh, c, "pn", 4 * j + k) for k in range(4)]}
            for side, i in (("a", 2 * j), ("b", 2 * j + 1)):
                if i < len(its):
                    sign, mcol, bus, cols = its[i]
                    row[side] = {"sign": P - 1 if sign else 1, "mkey": 0 if mcol is None else comb(mcol), "bus": bus, "vkeys": [comb(x) for x in cols]}
                else:
                    row[side] = None
            sh.lrows.append(row)
        Qc = sh.cols[c]
        sh.lrows.append({"chip": c, "kind": "bnd", "phi": [key_op(sh, c, "pl", 4 * Qc + k) for k in range(4)], "phin": [key_op(sh, c, "pn", 4 * Qc + k) for k in range(4)],
                         "sels": [key_sel(sh, c, w) for w in range(3)]})
    for row in sh.lrows:
        for k in row["phi"] + row["phin"] + row.get("sels", []):
            sh.mult[k] += 1
        for side in ("a", "b"):
            if row.get(side):
                sh.mult[row[side]["mkey"]] += 1
                for k in row[side]["vkeys"]:
                    sh.mult[k] += 1
    for c in range(sh.C):                                                       # SCALARS reads the chip's eight quotient values
        for k in range(8):
            sh.mult[key_op(sh, c, "q", k)] += 1
    sh.lrow0 = [min(i for i, r in enumerate(sh.lrows) if r["chip"] == c) for c in range(sh.C)]
    sh.lrow1 = [max(i for i, r in enumerate(sh.lrows) if r["chip"] == c) for c in range(sh.C)]


# ---------------------------------------------------------------------------------------------------------------- EVAL
# tests/recursion_air.py's chip, for the programs of ALL chips of the inner machine: the fold restarts with every chip (the chip's first
# term stands on a row with ACCIN = 0) and leaves at the chip's last term for the LOGUP chip, which folds the chip's lookup constraints on.
EV_PRE = 12
EP_COEF, EP_K0, EP_FIRSTC, EP_ACT, EP_LAST, EP_PID, EP_NFC, EP_CFIRST, EP_CKEY = 0, 1, 4, 5, 6, 7, 8, 9, 10
EV_F0, EV_M, EV_TV, EV_ACCIN, EV_ACCO, EV_ALPHA, EV_MAIN = 0, 12, 16, 20, 24, 28, 32


def eval_program(sh):
    M0 = EV_PRE
    cons = Cons()
    f0, f1, f2, mm, tv, ai, ao, al = (ev(M0 + c) for c in (EV_F0, EV_F0 + 4, EV_F0 + 8, EV_M, EV_TV, EV_ACCIN, EV_ACCO, EV_ALPHA))
    cons.ext(O.SEL_ALL, esub(mm, emul(f0, f1)))
    cons.ext(O.SEL_ALL, esub(tv, egate(pv(EP_COEF), emul(mm, f2))))
    cons.ext(O.SEL_ALL, esub(ao, eadd(ai, egate(pv(EP_FIRSTC), esub(emul(ai, al), ai)), tv)))
    cons.ext(O.SEL_TRANSITION, egate(pv(EP_NFC, True), esub(ev(M0 + EV_ACCIN, True), ao)))       # NFC: the row continues its chip's fold
    cons.ext(O.SEL_ALL, egate(pv(EP_CFIRST), ai))
    return O.air_program(EV_PRE + EV_MAIN, sh.NP * sh.NPUB, cons.c)


def eval_table():
    M0 = EV_PRE
    it = [(RECV, EP_ACT, BUS_VAL, [EP_K0 + j] + _e4(M0 + EV_F0 + 4 * j)) for j in range(3)]
    it += [(RECV, EP_ACT, BUS_EA, [EP_PID] + _e4(M0 + EV_ALPHA)), (SEND, EP_LAST, BUS_ACC, [EP_CKEY] + _e4(M0 + EV_ACCO))]
    return O.interaction_table(it)


def acc_key(sh, p, c, stage):
    """the fold of chip c of proof p on its way: stage 0 EVAL -> LOGUP, stage 1 LOGUP -> SCALARS"""
    return (p * sh.C + c) * 2 + stage


def eval_pre(sh, log_rows):
    t = np.zeros((1 << log_rows, EV_PRE), dtype=np.uint32)
    nt = len(sh.terms)
    for p in range(sh.NP):
        for i, (c, coeff, keys, first) in enumerate(sh.terms):
            r = t[p * nt + i]
            r[EP_COEF], r[EP_FIRSTC], r[EP_ACT], r[EP_PID] = coeff, first, 1, p
            r[EP_K0:EP_K0 + 3] = [p * sh.KSPAN + k for k in keys]
            r[EP_LAST], r[EP_CFIRST] = int(i == sh.term1[c]), int(i == sh.term0[c])
            r[EP_NFC], r[EP_CKEY] = int(i != sh.term0[c]), acc_key(sh, p, c, 0)
    return t


def value_of(sh, w, sc_rows, key):
    """the extension value behind a key of the value bus, for one proof (sc_rows: its SCALARS values per chip)"""
    if key == 0:
        return [1, 0, 0, 0]
    if key <= sh.NV:
        return list(w["stream"][4 * (key - 1):4 * key])
    if key <= sh.NV + sh.NPUB:
        return [w["pubs"][key - 1 - sh.NV], 0, 0, 0]
    c, which = divmod(key - 1 - sh.NV - sh.NPUB, 3)
    return sc_rows[c][("SELF", "SELL", "SELT")[which]]


def eval_main(sh, ws, scs, log_rows):
    """-> (trace, per proof and chip the program folded with alpha at zeta)"""
    t = np.zeros((1 << log_rows, EV_MAIN), dtype=np.uint64)
    nt = len(sh.terms)
    accs = []
    for p, (w, sc) in enumerate(zip(ws, scs)):
        alpha, run, per = w["alpha"], [0] * 4, {}
        for i, (c, coeff, keys, first) in enumerate(sh.terms):
            r = t[p * nt + i]
            if i == sh.term0[c]:
                run = [0] * 4
            f0, f1, f2 = (value_of(sh, w, sc, k) for k in keys)
            mm = ext_mul(f0, f1)
            tv = [x * coeff % P for x in ext_mul(mm, f2)]
            r[EV_F0:EV_F0 + 4], r[EV_F0 + 4:EV_F0 + 8], r[EV_F0 + 8:EV_F0 + 12], r[EV_M:EV_M + 4], r[EV_TV:EV_TV + 4] = f0, f1, f2, mm, tv
            r[EV_ACCIN:EV_ACCIN + 4] = run
            run = e_add(ext_mul(run, alpha) if first else run, tv)
            r[EV_ACCO:EV_ACCO + 4], r[EV_ALPHA:EV_ALPHA + 4] = run, alpha
            per[c] = run
        accs.append(per)
    return t.astype(np.uint32), accs


# ---------------------------------------------------------------------------------------------------------------- LOGUP
# The lookup constraints of the inner machine's chips at zeta (tests/pyverify_chips.py, "machine"): per chip one row per PAIR of interactions
# (2 j, 2 j + 1) -- the two fingerprints d = gamma + bus + sum_t beta^(t + 1) v_t, the signed multiplicities, phi_j from its four opened
# columns: phi d_a d_b - (m_a d_b + m_b d_a) -- and one BOUNDARY row: is_first (S - sum phi), is_transition (S' - S - sum phi'),
# is_last (S - C).  The fold of the chip's program arrives from EVAL, runs down the chip's rows and leaves for SCALARS.
LG_PRE = 64
(LP_ACT, LP_ISP, LP_ISB, LP_NF, LP_PFIRST, LP_NFP, LP_PID, LP_KIN, LP_KOUT, LP_CFIRST, LP_KCUM, LP_BUSA, LP_BUSB, LP_SA, LP_SB, LP_HASB, LP_NOB,
 LP_KMA, LP_KMB, LP_MB) = range(20)
LP_KA, LP_FA, LP_KB, LP_FB, LP_KP, LP_KPN = 20, 28, 36, 44, 52, 56            # value keys of side a (8), their receive flags (8), side b, phi's four columns, phi' 's


def logup_cols():
    m = Cols(LG_PRE)
    for name, w in (("VA", 32), ("VB", 32), ("MA", 4), ("MB", 4), ("PH", 16), ("PN", 16), ("PHI", 4), ("PHIN", 4), ("DA", 4), ("DB", 4), ("CST", 4), ("ACCIN", 4), ("ACCO", 4),
                    ("U1", 4), ("U2", 4), ("SUML", 4), ("SUMN", 4), ("CUM", 4), ("ALPHA", 4), ("GAMMA", 4), ("BP", 32)):
        m(name, w)
    return m


LG_MAIN = logup_cols().n - LG_PRE


def _from_columns(m, name):
    """sum_k X^k column_k: an extension column committed as four base columns, opened as four extension values"""
    out = [[], [], [], []]
    for k in range(4):
        e = ev(m[name] + 4 * k)
        for i in range(4):
            j = i + k
            out[j % 4] += pscale(e[i], EXT_W) if j >= 4 else e[i]
    return out


def logup_program(sh):
    m = logup_cols()
    cons = Cons()
    e = lambda n, nxt=False: ev(m[n], nxt)          # noqa: E731
    for name in ("ALPHA", "GAMMA", "BP"):
        for k in range(8 if name == "BP" else 1):
            cons.ext(O.SEL_TRANSITION, egate(pv(LP_NFP, True), esub(ev(m[name] + 4 * k, True), ev(m[name] + 4 * k))))
    for k in range(1, 8):                                                       # BP[k] = beta^(k + 1)
        cons.ext(O.SEL_ALL, esub(ev(m["BP"] + 4 * k), emul(ev(m["BP"] + 4 * (k - 1)), ev(m["BP"]))))
    cons.ext(O.SEL_ALL, esub(e("PHI"), _from_columns(m, "PH")))
    cons.ext(O.SEL_ALL, esub(e("PHIN"), _from_columns(m, "PN")))
    for side, V_, D_, BUS_, F_ in (("a", "VA", "DA", LP_BUSA, LP_FA), ("b", "VB", "DB", LP_BUSB, LP_FB)):
        d = eadd(egate(pv(LP_ISP) if side == "a" else pv(LP_HASB), e("GAMMA")), eb(pv(BUS_)))
        if side == "b":
            d = eadd(d, eb(pv(LP_NOB)))                                        # a pair without its second interaction: d_b = 1, m_b = 0
        for t in range(8):
            d = eadd(d, egate(pv(F_ + t), emul(ev(m["BP"] + 4 * t), ev(m[V_] + 4 * t))))
        cons.ext(O.SEL_ALL, esub(e(D_), d))
    ma, mb = egate(pv(LP_SA), e("MA")), egate(pv(LP_SB), e("MB"))
    cons.ext(O.SEL_ALL, esub(e("CST"), esub(emul(emul(e("PHI"), e("DA")), e("DB")), eadd(emul(ma, e("DB")), emul(mb, e("DA"))))))
    al = e("ALPHA")
    cons.ext(O.SEL_ALL, esub(e("U1"), eadd(emul(e("ACCIN"), al), emul(ev(m["VA"]), esub(e("PHI"), e("SUML"))))))                       # boundary row: VA[0] = is_first
    cons.ext(O.SEL_ALL, esub(e("U2"), eadd(emul(e("U1"), al), emul(ev(m["VA"] + 8), esub(esub(e("PHIN"), e("PHI")), e("SUMN"))))))      # VA[2] = is_transition
    cons.ext(O.SEL_ALL, esub(e("ACCO"), eadd(egate(pv(LP_ISP), eadd(emul(e("ACCIN"), al), e("CST"))),
                                             egate(pv(LP_ISB), eadd(emul(e("U2"), al), emul(ev(m["VA"] + 4), esub(e("PHI"), e("CUM"))))))))   # VA[1] = is_last
    cons.ext(O.SEL_TRANSITION, egate(pv(LP_NF, True), esub(e("ACCIN", True), e("ACCO"))))
    cons.ext(O.SEL_TRANSITION, egate(pv(LP_NF, True), esub(e("SUML", True), eadd(e("SUML"), e("PHI")))))
    cons.ext(O.SEL_TRANSITION, egate(pv(LP_NF, True), esub(e("SUMN", True), eadd(e("SUMN"), e("PHIN")))))
    cons.ext(O.SEL_ALL, egate(pv(LP_CFIRST), e("SUML")))
    cons.ext(O.SEL_ALL, egate(pv(LP_CFIRST), e("SUMN")))
    return O.air_program(LG_PRE + LG_MAIN, sh.NP * sh.NPUB, cons.c)


def logup_table():
    m = logup_cols()
    it = [(RECV, LP_FA + t, BUS_VAL, [LP_KA + t] + _e4(m["VA"] + 4 * t)) for t in range(8)]
    it += [(RECV, LP_FB + t, BUS_VAL, [LP_KB + t] + _e4(m["VB"] + 4 * t)) for t in range(8)]
    it += [(RECV, LP_ISP, BUS_VAL, [LP_KMA] + _e4(m["MA"])), (RECV, LP_MB, BUS_VAL, [LP_KMB] + _e4(m["MB"]))]
    it += [(RECV, LP_ACT, BUS_VAL, [LP_KP + k] + _e4(m["PH"] + 4 * k)) for k in range(4)]
    it += [(RECV, LP_ACT, BUS_VAL, [LP_KPN + k] + _e4(m["PN"] + 4 * k)) for k in range(4)]
    it += [(RECV, LP_ISB, BUS_CS, [LP_KCUM] + _e4(m["CUM"])), (RECV, LP_CFIRST, BUS_ACC, [LP_KIN] + _e4(m["ACCIN"])), (SEND, LP_ISB, BUS_ACC, [LP_KOUT] + _e4(m["ACCO"])),
           (RECV, LP_PFIRST, BUS_KL, [LP_PID] + _e4(m["ALPHA"])), (RECV, LP_PFIRST, BUS_KL + 1, [LP_PID] + _e4(m["GAMMA"])), (RECV, LP_PFIRST, BUS_KL + 2, [LP_PID] + _e4(m["BP"]))]
    return O.interaction_table(it)


BUS_KL = 101                                                                    # .. 103: alpha, gamma, beta from SCALARS to the LOGUP chip's first row of a proof


def logup_pre(sh, log_rows):
    t = np.zeros((1 << log_rows, LG_PRE), dtype=np.uint32)
    n = len(sh.lrows)
    for p in range(sh.NP):
        K = p * sh.KSPAN
        for i, row in enumerate(sh.lrows):
            r, c = t[p * n + i], row["chip"]
            r[LP_ACT], r[LP_PID] = 1, p
            r[LP_PFIRST], r[LP_NFP] = int(i == 0), int(i != 0)
            r[LP_CFIRST], r[LP_NF] = int(i == sh.lrow0[c]), int(i != sh.lrow0[c])
            r[LP_KIN], r[LP_KOUT], r[LP_KCUM] = acc_key(sh, p, c, 0), acc_key(sh, p, c, 1), p * sh.C + c
            for k in range(4):
                r[LP_KP + k], r[LP_KPN + k] = K + row["phi"][k], K + row["phin"][k]
            if row["kind"] == "bnd":
                r[LP_ISB] = 1
                for k in range(3):                                              # the selectors arrive in side a's value slots 0 (first), 1 (last), 2 (transition)
                    r[LP_KA + k], r[LP_FA + k] = K + row["sels"][k], 1
                continue
            r[LP_ISP] = 1
            a, b = row["a"], row["b"]
            r[LP_BUSA], r[LP_SA], r[LP_KMA] = a["bus"], a["sign"], K + a["mkey"]
            for tt, k in enumerate(a["vkeys"]):
                r[LP_KA + tt], r[LP_FA + tt] = K + k, 1
            if b:
                r[LP_HASB], r[LP_MB], r[LP_BUSB], r[LP_SB], r[LP_KMB] = 1, 1, b["bus"], b["sign"], K + b["mkey"]
                for tt, k in enumerate(b["vkeys"]):
                    r[LP_KB + tt], r[LP_FB + tt] = K + k, 1
            else:
                r[LP_NOB] = 1
    return t


def logup_main(sh, ws, scs, accs, log_rows):
    """-> (trace, per proof and chip the fold after the chip's lookup constraints)"""
    m = logup_cols()
    t = np.zeros((1 << log_rows, LG_MAIN), dtype=np.uint64)
    n = len(sh.lrows)
    outs = []

    def put(r, name, val, k=0):
        r[m[name] - LG_PRE + 4 * k:m[name] - LG_PRE + 4 * k + 4] = val

    def from_columns(four):
        acc = [0, 0, 0, 0]
        for k in range(4):
            basis = [0, 0, 0, 0]
            basis[k] = 1
            acc = e_add(acc, ext_mul(basis, four[k]))
        return acc
    for p, (w, sc, acc_in) in enumerate(zip(ws, scs, accs)):
        alpha, gamma, beta = w["alpha"], w["gamma"], w["beta"]
        bp = [beta]
        for _ in range(7):
            bp.append(ext_mul(bp[-1], beta))
        per, acc, suml, sumn = {}, None, None, None
        for i, row in enumerate(sh.lrows):
            r, c = t[p * n + i], row["chip"]
            val = lambda k: value_of(sh, w, sc, k)          # noqa: E731
            if i == sh.lrow0[c]:
                acc, suml, sumn = acc_in[c], [0] * 4, [0] * 4
            put(r, "ALPHA", alpha), put(r, "GAMMA", gamma)
            for k in range(8):
                put(r, "BP", bp[k], k)
            ph, pn = [val(k) for k in row["phi"]], [val(k) for k in row["phin"]]
            for k in range(4):
                put(r, "PH", ph[k], k), put(r, "PN", pn[k], k)
            phi, phin = from_columns(ph), from_columns(pn)
            put(r, "PHI", phi), put(r, "PHIN", phin), put(r, "ACCIN", acc), put(r, "SUML", suml), put(r, "SUMN", sumn)
            va, vb, ma, mb = [[0] * 4] * 8, [[0] * 4] * 8, [0] * 4, [0] * 4
            fa_, fb_, busa, busb, sa, sb, isp, hasb, nob = [0] * 8, [0] * 8, 0, 0, 0, 0, 0, 0, 0
            if row["kind"] == "bnd":
                va = [val(k) for k in row["sels"]] + [[0] * 4] * 5
                fa_ = [1, 1, 1, 0, 0, 0, 0, 0]
            else:
                isp, a, b = 1, row["a"], row["b"]
                va = [val(k) for k in a["vkeys"]] + [[0] * 4] * (8 - len(a["vkeys"]))
                fa_ = [1] * len(a["vkeys"]) + [0] * (8 - len(a["vkeys"]))
                ma, busa, sa = val(a["mkey"]), a["bus"], a["sign"]
                if b:
                    vb = [val(k) for k in b["vkeys"]] + [[0] * 4] * (8 - len(b["vkeys"]))
                    fb_ = [1] * len(b["vkeys"]) + [0] * (8 - len(b["vkeys"]))
                    mb, busb, sb, hasb = val(b["mkey"]), b["bus"], b["sign"], 1
                else:
                    nob = 1
            for k in range(8):
                put(r, "VA", va[k], k), put(r, "VB", vb[k], k)
            put(r, "MA", ma), put(r, "MB", mb)
            da = e_add([x * isp % P for x in gamma], [busa, 0, 0, 0])
            db = e_add([x * hasb % P for x in gamma], [(busb + nob) % P, 0, 0, 0])
            for k in range(8):
                if fa_[k]:
                    da = e_add(da, ext_mul(bp[k], va[k]))
                if fb_[k]:
                    db = e_add(db, ext_mul(bp[k], vb[k]))
            sma, smb = [x * sa % P for x in ma], [x * sb % P for x in mb]
            cst = e_sub(ext_mul(ext_mul(phi, da), db), e_add(ext_mul(sma, db), ext_mul(smb, da)))
            u1 = e_add(ext_mul(acc, alpha), ext_mul(va[0], e_sub(phi, suml)))
            u2 = e_add(ext_mul(u1, alpha), ext_mul(va[2], e_sub(e_sub(phin, phi), sumn)))
            cum = w["cumsum"][c] if row["kind"] == "bnd" else [0] * 4
            if row["kind"] == "bnd":
                acco = e_add(ext_mul(u2, alpha), ext_mul(va[1], e_sub(phi, cum)))
            else:
                acco = e_add(ext_mul(acc, alpha), cst)
            put(r, "DA", da), put(r, "DB", db), put(r, "CST", cst), put(r, "U1", u1), put(r, "U2", u2), put(r, "CUM", cum), put(r, "ACCO", acco)
            acc, suml, sumn = acco, e_add(suml, phi), e_add(sumn, phin)
            per[c] = acco
        outs.append(per)
    return t.astype(np.uint32), outs


# ---------------------------------------------------------------------------------------------------------------- SCALARS
# One row per (proof, chip of the inner machine): zeta^N for the chip's own N (the squarings of zeta are on every row, a preprocessed
# one-hot picks the chip's), Z_H, the three selectors, the chunk weights, quotient(zeta) from the chip's eight opened quotient values, and
# the chip's AIR identity fold = quotient Z_H with the fold that arrives from LOGUP.  A proof's first row receives the five challenges and
# hands them on; its rows add up the cumulative sums (zero at the last); the key's tree must end in the inner key's root.
def sc_cols(sh):
    c = Cols()
    for name, w in (("ACT", 1), ("PFIRST", 1), ("NFP", 1), ("PID", 1), ("KACC", 1), ("KCUM", 1), ("LASTC", 1), ("OH", sh.R + 1), ("WINV", 1), ("WN", 1), ("ZA0", 1), ("ZB0", 1), ("ZA1", 1),
                    ("ZB1", 1), ("KQZ", 8), ("KSEL", 3), ("MSEL", 3), ("KONE", 1), ("MONE", 1), ("Z", 1), ("KZH", 1), ("MZH", 1), ("MKR", 1), ("TREE", 1), ("KIND", N_CHAL),
                    ("MEA", 1), ("MFA", 1)):
        c(name, w)
    pre = rup4(c.n)
    m = Cols(pre)
    for name in ("ALPHA", "ZETA", "FA", "GAMMA", "BETA"):
        m(name)
    m("ZP", 4 * sh.R)
    for name in ("ZN", "INVF", "INVT", "SELF", "SELL", "SELT", "ZNX"):
        m(name)
    m("QZ", 32)
    for name in ("Q0", "Q1", "QUO", "ACC", "CUM", "TOTIN", "TOTO"):
        m(name)
    m("KR", 8)
    return c, m, pre


def scalars_program(sh):
    c, m, pre = sc_cols(sh)
    cons = Cons()
    e = lambda n, nxt=False: ev(m[n], nxt)          # noqa: E731
    for name in ("ALPHA", "ZETA", "FA", "GAMMA", "BETA"):
        cons.ext(O.SEL_TRANSITION, egate(pv(c["NFP"], True), esub(e(name, True), e(name))))
    zp = [e("ZETA")] + [ev(m["ZP"] + 4 * k) for k in range(sh.R)]              # zeta^(2^k), k = 0 .. R
    for k in range(sh.R):
        cons.ext(O.SEL_ALL, esub(zp[k + 1], emul(zp[k], zp[k])))
    zn = [[], [], [], []]
    for k in range(sh.R + 1):
        zn = eadd(zn, egate(pv(c["OH"] + k), zp[k]))
    cons.ext(O.SEL_ALL, esub(e("ZN"), zn))
    zh = esub(e("ZN"), eb(pv(c["ACT"])))
    one = eb(pv(c["ACT"]))
    cons.ext(O.SEL_ALL, esub(emul(esub(e("ZETA"), one), e("INVF")), one))
    cons.ext(O.SEL_ALL, esub(emul(esub(e("ZETA"), eb(pv(c["WINV"]))), e("INVT")), one))
    cons.ext(O.SEL_ALL, esub(e("SELF"), emul(zh, e("INVF"))))
    cons.ext(O.SEL_ALL, esub(e("SELL"), emul(zh, e("INVT"))))
    cons.ext(O.SEL_ALL, esub(e("SELT"), esub(e("ZETA"), eb(pv(c["WINV"])))))
    cons.ext(O.SEL_ALL, esub(e("ZNX"), egate(pv(c["WN"]), e("ZETA"))))
    for k, name in ((0, "Q0"), (1, "Q1")):
        q = [[], [], [], []]
        for j in range(4):
            v = ev(m["QZ"] + 16 * k + 4 * j)
            for i in range(4):
                q[(i + j) % 4] += pscale(v[i], EXT_W) if i + j >= 4 else v[i]
        cons.ext(O.SEL_ALL, esub(e(name), q))
    zps0 = eadd(egate(pv(c["ZA0"]), e("ZN")), eb(pv(c["ZB0"])))
    zps1 = eadd(egate(pv(c["ZA1"]), e("ZN")), eb(pv(c["ZB1"])))
    cons.ext(O.SEL_ALL, esub(e("QUO"), eadd(emul(zps0, e("Q0")), emul(zps1, e("Q1")))))
    cons.ext(O.SEL_ALL, esub(e("ACC"), emul(e("QUO"), zh)))
    cons.ext(O.SEL_ALL, egate(pv(c["PFIRST"]), e("TOTIN")))
    cons.ext(O.SEL_ALL, esub(e("TOTO"), eadd(e("TOTIN"), e("CUM"))))
    cons.ext(O.SEL_TRANSITION, egate(pv(c["NFP"], True), esub(e("TOTIN", True), e("TOTO"))))
    cons.ext(O.SEL_ALL, egate(pv(c["LASTC"]), e("TOTO")))
    if "E" in sh.trees:
        for j in range(8):
            cons.add(O.SEL_ALL, pmul(pv(c["PFIRST"]), padd(pv(m["KR"] + j), pc(P - sh.key_root[j]))))
    return O.air_program(pre + rup4(m.n - pre), sh.NP * sh.NPUB, cons.c)


BUS_ZH0, BUS_ZH1, BUS_KFA = 104, 105, 106


def scalars_table(sh):
    c, m, _ = sc_cols(sh)
    it = [(RECV, c["ACT"], BUS_ACC, [c["KACC"]] + _e4(m["ACC"])), (RECV, c["ACT"], BUS_CS, [c["KCUM"]] + _e4(m["CUM"]))]
    it += [(RECV, c["ACT"], BUS_VAL, [c["KQZ"] + k] + _e4(m["QZ"] + 4 * k)) for k in range(8)]
    it += [(SEND, c["MSEL"] + i, BUS_VAL, [c["KSEL"] + i] + _e4(m[name])) for i, name in enumerate(("SELF", "SELL", "SELT"))]
    it += [(SEND, c["MONE"], BUS_VAL, [c["KONE"], c["PFIRST"], c["Z"], c["Z"], c["Z"]])]
    it += [(RECV, c["PFIRST"], BUS_SC, [c["KIND"] + k] + _e4(m[name])) for k, name in enumerate(("ALPHA", "ZETA", "FA", "GAMMA", "BETA"))]
    it += [(SEND, c["MEA"], BUS_EA, [c["PID"]] + _e4(m["ALPHA"]))]
    it += [(SEND, c["PFIRST"], BUS_KL + k, [c["PID"]] + _e4(m[name])) for k, name in enumerate(("ALPHA", "GAMMA", "BETA"))]
    it += [(SEND, c["MFA"], BUS_KFA, [c["PID"]] + _e4(m["FA"]))]
    it += [(SEND, c["MZH"], BUS_ZH0, [c["KZH"]] + _e4(m["ZETA"])), (SEND, c["MZH"], BUS_ZH1, [c["KZH"]] + _e4(m["ZNX"]))]
    if "E" in sh.trees:
        it += [(RECV, c["MKR"], F.BUS_R0, [c["TREE"]] + _e4(m["KR"])), (RECV, c["MKR"], F.BUS_R1, [c["TREE"]] + _e4(m["KR"] + 4))]
    return O.interaction_table(it)


def chunk_weights(ln):
    """zps_k(zeta) = A_k zeta^N + B_k for the two quotient chunks of a trace domain of 2^ln rows (tests/pyverify_chips.py)"""
    N = 1 << ln
    wq = two_adic_generator(ln + 1)
    sN = [pow(GEN * pow(wq, k, P) % P, N, P) for k in range(2)]
    out = []
    for k in range(2):
        j = 1 - k
        sjn_inv = pow(sN[j], -1, P)
        den_inv = pow((sN[k] * sjn_inv - 1) % P, -1, P)
        out.append((sjn_inv * den_inv % P, (P - den_inv) % P))
    return out


def scalars_pre(sh, log_rows):
    c, _, pre = sc_cols(sh)
    t = np.zeros((1 << log_rows, pre), dtype=np.uint32)
    for p in range(sh.NP):
        K = p * sh.KSPAN
        for ch in range(sh.C):
            r = t[p * sh.C + ch]
            ln = sh.ln[ch]
            wN = two_adic_generator(ln)
            r[c["ACT"]], r[c["PID"]], r[c["PFIRST"]], r[c["NFP"]], r[c["LASTC"]] = 1, p, int(ch == 0), int(ch != 0), int(ch == sh.C - 1)
            r[c["KACC"]], r[c["KCUM"]] = acc_key(sh, p, ch, 1), p * sh.C + ch
            r[c["OH"] + ln] = 1
            r[c["WINV"]], r[c["WN"]] = pow(wN, -1, P), wN
            (r[c["ZA0"]], r[c["ZB0"]]), (r[c["ZA1"]], r[c["ZB1"]]) = chunk_weights(ln)
            for k in range(8):
                r[c["KQZ"] + k] = K + key_op(sh, ch, "q", k)
            for i in range(3):
                r[c["KSEL"] + i], r[c["MSEL"] + i] = K + key_sel(sh, ch, i), sh.mult[key_sel(sh, ch, i)]
            if ch == 0:
                r[c["KONE"]], r[c["MONE"]] = K, sh.mult[0]
                r[c["MEA"]], r[c["MFA"]] = len(sh.terms), 2
                for k in range(N_CHAL):
                    r[c["KIND"] + k] = N_CHAL * p + k
                if "E" in sh.trees:
                    r[c["MKR"]], r[c["TREE"]] = sh.Q, sh.tree_id(p, "E")
            if ch == 0 or sh.lh[ch] != sh.lh[ch - 1]:                           # the first chip of its height speaks for the height: zeta and zeta g_h to the QUERY rows
                r[c["KZH"]], r[c["MZH"]] = p * 32 + sh.lh[ch], sh.Q
    return t


def scalars_values(sh, w):
    """per chip of one proof: the row's values by name"""
    zeta, out = w["zeta"], []
    zp = [zeta]
    for _ in range(sh.R):
        zp.append(ext_mul(zp[-1], zp[-1]))
    tot = [0] * 4
    for ch in range(sh.C):
        ln = sh.ln[ch]
        wN = two_adic_generator(ln)
        winv = pow(wN, -1, P)
        v = {"ALPHA": w["alpha"], "ZETA": zeta, "FA": w["fa"], "GAMMA": w["gamma"], "BETA": w["beta"], "ZP": zp[1:], "ZN": zp[ln]}
        zh = e_sub(v["ZN"], [1, 0, 0, 0])
        v["INVF"], v["INVT"] = pyref.ext_inv(e_sub(zeta, [1, 0, 0, 0])), pyref.ext_inv(e_sub(zeta, [winv, 0, 0, 0]))
        v["SELF"], v["SELL"], v["SELT"] = ext_mul(zh, v["INVF"]), ext_mul(zh, v["INVT"]), e_sub(zeta, [winv, 0, 0, 0])
        v["ZNX"] = [x * wN % P for x in zeta]
        qz = w["opened"][ch][4]
        v["QZ"] = [list(x) for x in qz]
        qs = []
        for k in range(2):
            acc = [0] * 4
            for j in range(4):
                basis = [0, 0, 0, 0]
                basis[j] = 1
                acc = e_add(acc, ext_mul(basis, qz[4 * k + j]))
            qs.append(acc)
        v["Q0"], v["Q1"] = qs
        (a0, b0), (a1, b1) = chunk_weights(ln)
        zps0, zps1 = e_add([x * a0 % P for x in v["ZN"]], [b0, 0, 0, 0]), e_add([x * a1 % P for x in v["ZN"]], [b1, 0, 0, 0])
        v["QUO"] = e_add(ext_mul(zps0, qs[0]), ext_mul(zps1, qs[1]))
        v["ACC"] = ext_mul(v["QUO"], zh)
        v["CUM"], v["TOTIN"] = list(w["cumsum"][ch]), tot
        tot = e_add(tot, w["cumsum"][ch])
        v["TOTO"] = tot
        out.append(v)
    assert tot == [0] * 4
    return out


def scalars_main(sh, scs, log_rows):
    c, m, pre = sc_cols(sh)
    width = rup4(m.n - pre)
    t = np.zeros((1 << log_rows, width), dtype=np.uint64)
    for p, sc in enumerate(scs):
        for ch, v in enumerate(sc):
            r = t[p * sh.C + ch]
            for name, col in m.at.items():
                if name == "ZP":
                    for k in range(sh.R):
                        r[col - pre + 4 * k:col - pre + 4 * k + 4] = v["ZP"][k]
                elif name == "QZ":
                    for k in range(8):
                        r[col - pre + 4 * k:col - pre + 4 * k + 4] = v["QZ"][k]
                elif name == "KR":
                    if ch == 0 and "E" in sh.trees:
                        r[col - pre:col - pre + 8] = sh.key_root
                else:
                    r[col - pre:col - pre + 4] = v[name]
    return t.astype(np.uint32)


# ---------------------------------------------------------------------------------------------------------------- OPENED (the stream)
# One row per 8-word block of the opened values as the transcript absorbs them: two extension values with consecutive batching exponents
# e, e + 1.  PW = fa^e runs down the rows and restarts where the height changes; the values add to the height's sum at zeta or at zeta g
# (a preprocessed flag says which); the row's two values go out on the value bus, its PW to the ROWSUM rows that weight a segment with it.
OS_PRE = 24
(OS_ACT, OS_TAG, OS_PFIRST, OS_NFP, OS_PID, OS_ISN, OS_NISN, OS_RST, OS_NRST, OS_K0, OS_M0, OS_K1, OS_M1, OS_KPW, OS_MPW, OS_KYH, OS_MYH) = range(17)
OS_W, OS_FA, OS_FA2, OS_PW, OS_M, OS_YZIN, OS_YNIN, OS_YZO, OS_YNO, OS_PWN, OS_MAIN = 0, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44
BUS_YH0, BUS_YH1 = 107, 108
PWSPAN = 1 << 14                                                                # exponents of one (proof, height) on BUS_PW


def pw_key(sh, p, h, e):
    return (p * 32 + h) * PWSPAN + e


def stream_rows(sh):
    """per stream row (two extension values): (chip, kind, column of the first value, height, exponent, first row of its height, last row of its height)"""
    out = []
    for c, kind, pos, n, e in sh.segs:
        for j in range(0, n, 2):
            out.append([c, kind, j, sh.lh[c], e + j, False, False])
    for i, r in enumerate(out):
        r[5] = i == 0 or out[i - 1][3] != r[3]
        r[6] = i + 1 == len(out) or out[i + 1][3] != r[3]
    return out


def opened_program(sh):
    M0 = OS_PRE
    cons = Cons()
    v0, v1, fa, fa2, pw, mm = ev(M0 + OS_W), ev(M0 + OS_W + 4), ev(M0 + OS_FA), ev(M0 + OS_FA2), ev(M0 + OS_PW), ev(M0 + OS_M)
    for col in (OS_FA, OS_FA2):
        cons.ext(O.SEL_TRANSITION, egate(pv(OS_NFP, True), esub(ev(M0 + col, True), ev(M0 + col))))
    cons.ext(O.SEL_ALL, esub(fa2, emul(fa, fa)))
    cons.ext(O.SEL_ALL, esub(mm, eadd(v0, emul(fa, v1))))
    cons.ext(O.SEL_ALL, egate(pv(OS_RST), esub(pw, ec(1))))
    cons.ext(O.SEL_ALL, esub(ev(M0 + OS_PWN), emul(pw, fa2)))                  # (a column of its own: a transition constraint under a flag has one degree left)
    cons.ext(O.SEL_TRANSITION, egate(pv(OS_NRST, True), esub(ev(M0 + OS_PW, True), ev(M0 + OS_PWN))))
    cons.ext(O.SEL_ALL, esub(ev(M0 + OS_YZO), eadd(ev(M0 + OS_YZIN), egate(pv(OS_NISN), emul(pw, mm)))))
    cons.ext(O.SEL_ALL, esub(ev(M0 + OS_YNO), eadd(ev(M0 + OS_YNIN), egate(pv(OS_ISN), emul(pw, mm)))))
    cons.ext(O.SEL_TRANSITION, egate(pv(OS_NRST, True), esub(ev(M0 + OS_YZIN, True), ev(M0 + OS_YZO))))
    cons.ext(O.SEL_TRANSITION, egate(pv(OS_NRST, True), esub(ev(M0 + OS_YNIN, True), ev(M0 + OS_YNO))))
    cons.ext(O.SEL_ALL, egate(pv(OS_RST), ev(M0 + OS_YZIN)))
    cons.ext(O.SEL_ALL, egate(pv(OS_RST), ev(M0 + OS_YNIN)))
    return O.air_program(OS_PRE + OS_MAIN, sh.NP * sh.NPUB, cons.c)


def opened_table():
    M0 = OS_PRE
    W = M0 + OS_W
    return O.interaction_table([
        (RECV, OS_ACT, BUS_IN0, [OS_TAG, W, W + 1, W + 2, W + 3]), (RECV, OS_ACT, BUS_IN1, [OS_TAG, W + 4, W + 5, W + 6, W + 7]),
        (SEND, OS_M0, BUS_VAL, [OS_K0] + _e4(W)), (SEND, OS_M1, BUS_VAL, [OS_K1] + _e4(W + 4)),
        (SEND, OS_MPW, BUS_PW, [OS_KPW] + _e4(M0 + OS_PW)),
        (SEND, OS_MYH, BUS_YH0, [OS_KYH] + _e4(M0 + OS_YZO)), (SEND, OS_MYH, BUS_YH1, [OS_KYH] + _e4(M0 + OS_YNO)),
        (RECV, OS_PFIRST, BUS_KFA, [OS_PID] + _e4(M0 + OS_FA))])


def pw_uses(sh):
    """how many ROWSUM rows (per query) weight a segment with fa^e of height h: {(h, e): count}"""
    uses = {}
    for c in range(sh.C):
        for tr, kz, kn in (("E", "el", "en"), ("T", "tl", "tn"), ("P", "pl", "pn"), ("Q", "q", None)):
            if sh.tree_w[tr][c] == 0:
                continue
            for kind in (kz, kn):
                if kind:
                    key = (sh.lh[c], sh.seg_at[(c, kind)][2])
                    uses[key] = uses.get(key, 0) + 1
    return uses


def opened_pre(sh, log_rows):
    t = np.zeros((1 << log_rows, OS_PRE), dtype=np.uint32)
    rows, uses = stream_rows(sh), pw_uses(sh)
    n = len(rows)
    assert n == sh.NV // 2
    for p in range(sh.NP):
        K = p * sh.KSPAN
        for i, (c, kind, j, h, e, first, last) in enumerate(rows):
            r = t[p * n + i]
            isn = int(kind in ("en", "tn", "pn"))
            r[OS_ACT], r[OS_TAG], r[OS_PFIRST], r[OS_NFP], r[OS_PID] = 1, sh.ttag(p, sh.TO0 + i), int(i == 0), int(i != 0), p
            r[OS_ISN], r[OS_NISN], r[OS_RST], r[OS_NRST] = isn, 1 - isn, int(first), int(not first)
            k0 = key_op(sh, c, kind, j)
            r[OS_K0], r[OS_M0], r[OS_K1], r[OS_M1] = K + k0, sh.mult[k0], K + k0 + 1, sh.mult[k0 + 1]
            if (h, e) in uses:
                r[OS_KPW], r[OS_MPW] = pw_key(sh, p, h, e), sh.Q * uses[(h, e)]
            if last:
                r[OS_KYH], r[OS_MYH] = p * 32 + h, sh.Q
    return t


def opened_main(sh, ws, log_rows):
    """-> (trace, per proof {height: (Yz, Yn)}, per proof {(height, exponent): fa^e})"""
    t = np.zeros((1 << log_rows, OS_MAIN), dtype=np.uint64)
    rows = stream_rows(sh)
    n = len(rows)
    ys, pws = [], []
    for p, w in enumerate(ws):
        fa = w["fa"]
        fa2 = ext_mul(fa, fa)
        yh, pwh = {}, {}
        pw = yz = yn = None
        for i, (c, kind, j, h, e, first, last) in enumerate(rows):
            r = t[p * n + i]
            if first:
                pw, yz, yn = [1, 0, 0, 0], [0] * 4, [0] * 4
            words = w["stream"][8 * i:8 * i + 8]
            v0, v1 = words[:4], words[4:]
            mm = e_add(v0, ext_mul(fa, v1))
            r[OS_W:OS_W + 8], r[OS_FA:OS_FA + 4], r[OS_FA2:OS_FA2 + 4], r[OS_PW:OS_PW + 4], r[OS_M:OS_M + 4] = words, fa, fa2, pw, mm
            r[OS_YZIN:OS_YZIN + 4], r[OS_YNIN:OS_YNIN + 4] = yz, yn
            add = ext_mul(pw, mm)
            if kind in ("en", "tn", "pn"):
                yn = e_add(yn, add)
            else:
                yz = e_add(yz, add)
            r[OS_YZO:OS_YZO + 4], r[OS_YNO:OS_YNO + 4] = yz, yn
            pwh[(h, e)] = pw
            pw = ext_mul(pw, fa2)
            r[OS_PWN:OS_PWN + 4] = pw
            if last:
                yh[h] = (yz, yn)
        ys.append(yh), pws.append(pwh)
    return t.astype(np.uint32), ys, pws


# ---------------------------------------------------------------------------------------------------------------- ROWSUM
# One row per 8-word sponge block of an opened row.  The rows of a (query, height) stand together -- the blocks of the key's tree, the main,
# the permutation and the quotient tree at that height, each tree's blocks from the last to the first (Horner from the back) -- so that the
# height's two weighted sums Az, An run down them.  A block holds up to two SEGMENTS (a matrix's row; every width is a multiple of 4):
# Horner restarts behind a segment's last word (R7: behind word 7, R3: behind word 3); where a segment STARTS (word 0 or word 4) its sum
# sum_j fa^j row[j] stands in T[0] or T[4] and is weighted with the two powers its chip and kind have in the reduced opening.
RS_PRE = 24
(RP_TAG, RP_ACT, RP_RIN1, RP_HALF, RP_NR7, RP_NR3, RP_F0, RP_F4, RP_N0, RP_N4, RP_KZ0, RP_KN0, RP_KZ4, RP_KN4, RP_GFIRST, RP_NG, RP_GLAST, RP_KAH, RP_PFIRST, RP_NFP,
 RP_PID) = range(21)
RS_V, RS_ACCIN, RS_T, RS_FA, RS_KZ0, RS_KN0, RS_KZ4, RS_KN4, RS_AZIN, RS_ANIN, RS_AZO, RS_ANO, RS_MAIN = 0, 8, 12, 44, 48, 52, 56, 60, 64, 68, 72, 76, 80
BUS_AH0, BUS_AH1 = 109, 110


def rowsum_program(sh):
    M0 = RS_PRE
    cons = Cons()
    fa = ev(M0 + RS_FA)
    cons.ext(O.SEL_TRANSITION, egate(pv(RP_NFP, True), esub(ev(M0 + RS_FA, True), fa)))
    prev = ev(M0 + RS_ACCIN)
    for s in range(7, -1, -1):
        cur = ev(M0 + RS_T + 4 * s)
        carried = emul(prev, fa)
        if s == 7:
            carried = egate(pv(RP_NR7), carried)
        if s == 3:
            carried = egate(pv(RP_NR3), carried)
        cons.ext(O.SEL_ALL, esub(cur, eadd(carried, eb(pv(M0 + RS_V + s)))))
        prev = cur
    cons.ext(O.SEL_TRANSITION, egate(pv(RP_NR7, True), esub(ev(M0 + RS_ACCIN, True), ev(M0 + RS_T))))
    for j in range(4):
        cons.add(O.SEL_ALL, pmul(pv(RP_HALF), pv(M0 + RS_V + 4 + j)))
    t0, t4 = ev(M0 + RS_T), ev(M0 + RS_T + 16)
    cons.ext(O.SEL_ALL, esub(ev(M0 + RS_AZO), eadd(ev(M0 + RS_AZIN), egate(pv(RP_F0), emul(ev(M0 + RS_KZ0), t0)), egate(pv(RP_F4), emul(ev(M0 + RS_KZ4), t4)))))
    cons.ext(O.SEL_ALL, esub(ev(M0 + RS_ANO), eadd(ev(M0 + RS_ANIN), egate(pv(RP_N0), emul(ev(M0 + RS_KN0), t0)), egate(pv(RP_N4), emul(ev(M0 + RS_KN4), t4)))))
    cons.ext(O.SEL_TRANSITION, egate(pv(RP_NG, True), esub(ev(M0 + RS_AZIN, True), ev(M0 + RS_AZO))))
    cons.ext(O.SEL_TRANSITION, egate(pv(RP_NG, True), esub(ev(M0 + RS_ANIN, True), ev(M0 + RS_ANO))))
    cons.ext(O.SEL_ALL, egate(pv(RP_GFIRST), ev(M0 + RS_AZIN)))
    cons.ext(O.SEL_ALL, egate(pv(RP_GFIRST), ev(M0 + RS_ANIN)))
    return O.air_program(RS_PRE + RS_MAIN, sh.NP * sh.NPUB, cons.c)


def rowsum_table():
    M0 = RS_PRE
    v = M0 + RS_V
    return O.interaction_table([
        (SEND, RP_ACT, BUS_IN0, [RP_TAG, v, v + 1, v + 2, v + 3]), (SEND, RP_RIN1, BUS_IN1, [RP_TAG, v + 4, v + 5, v + 6, v + 7]),
        (RECV, RP_F0, BUS_PW, [RP_KZ0] + _e4(M0 + RS_KZ0)), (RECV, RP_N0, BUS_PW, [RP_KN0] + _e4(M0 + RS_KN0)),
        (RECV, RP_F4, BUS_PW, [RP_KZ4] + _e4(M0 + RS_KZ4)), (RECV, RP_N4, BUS_PW, [RP_KN4] + _e4(M0 + RS_KN4)),
        (SEND, RP_GLAST, BUS_AH0, [RP_KAH] + _e4(M0 + RS_AZO)), (SEND, RP_GLAST, BUS_AH1, [RP_KAH] + _e4(M0 + RS_ANO)),
        (RECV, RP_PFIRST, BUS_KFA, [RP_PID] + _e4(M0 + RS_FA))])


def ah_key(sh, p, q, h):
    return (p * sh.Q + q) * 32 + h


def rowsum_rows(sh):
    """per proof: (q, h, tree, block, words in the block, {0 / 4: (chip, kind z, kind n)} for the segments that START there, restart behind word 7, behind word 3)"""
    kinds = {"E": ("el", "en"), "T": ("tl", "tn"), "P": ("pl", "pn"), "Q": ("q", None)}
    out = []
    for q in range(sh.Q):
        for h in sh.hs:
            for tr in sh.trees:
                if h not in sh.tree_hs[tr]:
                    continue
                segs, words = sh.leaf_segs[(tr, h)]
                starts = {at: c for c, at, _ in segs}
                nb = (words + 7) // 8
                for b in range(nb - 1, -1, -1):
                    st = {s: (starts[8 * b + s],) + kinds[tr] for s in (0, 4) if 8 * b + s in starts}
                    r7 = b == nb - 1 or (8 * b + 8) in starts
                    r3 = (8 * b + 4) in starts
                    out.append((q, h, tr, b, min(8, words - 8 * b), st, r7, r3))
    return out


def rowsum_pre(sh, log_rows):
    t = np.zeros((1 << log_rows, RS_PRE), dtype=np.uint32)
    rows = rowsum_rows(sh)
    n = len(rows)
    for p in range(sh.NP):
        for i, (q, h, tr, b, k, st, r7, r3) in enumerate(rows):
            r = t[p * n + i]
            r[RP_TAG], r[RP_ACT], r[RP_RIN1], r[RP_HALF], r[RP_PID] = sh.blk_tag(p, q, tr, h, b), 1, int(k == 8), int(k < 8), p
            r[RP_NR7], r[RP_NR3] = int(not r7), int(not r3)
            for s, (fl, nl, kz, kn) in ((0, (RP_F0, RP_N0, RP_KZ0, RP_KN0)), (4, (RP_F4, RP_N4, RP_KZ4, RP_KN4))):
                if s in st:
                    c, kindz, kindn = st[s]
                    r[fl], r[kz] = 1, pw_key(sh, p, h, sh.seg_at[(c, kindz)][2])
                    if kindn:
                        r[nl], r[kn] = 1, pw_key(sh, p, h, sh.seg_at[(c, kindn)][2])
            first = i == 0 or rows[i - 1][:2] != (q, h)
            last = i + 1 == n or rows[i + 1][:2] != (q, h)
            r[RP_GFIRST], r[RP_NG], r[RP_GLAST], r[RP_KAH] = int(first), int(not first), int(last), ah_key(sh, p, q, h)
            r[RP_PFIRST], r[RP_NFP] = int(i == 0), int(i != 0)
    return t


def rowsum_main(sh, ws, pws, log_rows):
    """-> (trace, per proof {(q, h): (Az, An)})"""
    t = np.zeros((1 << log_rows, RS_MAIN), dtype=np.uint64)
    rows = rowsum_rows(sh)
    n = len(rows)
    outs = []
    for p, (w, pwh) in enumerate(zip(ws, pws)):
        fa = w["fa"]
        ah, acc, az, an = {}, [0] * 4, None, None
        for i, (q, h, tr, b, k, st, r7, r3) in enumerate(rows):
            r = t[p * n + i]
            if i == 0 or rows[i - 1][:2] != (q, h):
                az, an = [0] * 4, [0] * 4
            vals = leaf_words(sh, w, q, tr, h)[8 * b:8 * b + 8]
            vals = list(vals) + [0] * (8 - len(vals))
            r[RS_V:RS_V + 8], r[RS_FA:RS_FA + 4], r[RS_ACCIN:RS_ACCIN + 4], r[RS_AZIN:RS_AZIN + 4], r[RS_ANIN:RS_ANIN + 4] = vals, fa, acc, az, an
            steps, prev = [None] * 8, acc
            for s in range(7, -1, -1):
                carried = [0] * 4 if (s == 7 and r7) or (s == 3 and r3) else ext_mul(prev, fa)
                prev = [(carried[0] + vals[s]) % P] + carried[1:]
                steps[s] = prev
                r[RS_T + 4 * s:RS_T + 4 * s + 4] = prev
            acc = steps[0]
            for s, (cz, cn) in ((0, (RS_KZ0, RS_KN0)), (4, (RS_KZ4, RS_KN4))):
                if s in st:
                    c, kindz, kindn = st[s]
                    kz = pwh[(h, sh.seg_at[(c, kindz)][2])]
                    r[cz:cz + 4] = kz
                    az = e_add(az, ext_mul(kz, steps[s]))
                    if kindn:
                        kn = pwh[(h, sh.seg_at[(c, kindn)][2])]
                        r[cn:cn + 4] = kn
                        an = e_add(an, ext_mul(kn, steps[s]))
            r[RS_AZO:RS_AZO + 4], r[RS_ANO:RS_ANO + 4] = az, an
            ah[(q, h)] = (az, an)
        outs.append(ah)
    return t.astype(np.uint32), outs


# ---------------------------------------------------------------------------------------------------------------- QUERY
# One row per (query, height of the inner machine): the point x_h = g XS (XS: the fold chain's point at the layer where the height is
# reached -- layer 0 for the tallest), 1 / (x_h - zeta), 1 / (x_h - zeta g_h), and the height's reduced opening
# (Az - Yz) / (x_h - zeta) + (An - Yn) / (x_h - zeta g_h).  The fold chain sends (layer's name, index at that layer, XS, what it takes in
# there): its first row the value it starts from, a later row what joins its folded value -- both must be this row's reduced opening.
Q_PRE = 12
QP_ACT, QP_KEY, QP_KAH, QP_KYH, QP_KZH, QP_TOP, QP_QNS, QP_QN, QP_NQI, QP_LOW, QP_NQ = range(11)      # LOW: a height below the tallest; NQ: the row continues its query


def query_cols():
    m = Cols(Q_PRE)
    m("IDX", 1), m("XQ", 1), m("IDX0", 1), m("PAD", 1)
    for name in ("RO", "AZ", "AN", "YZ", "YN", "ZETA", "ZNX", "I1", "I2", "P1", "P2"):
        m(name)
    return m


Q_MAIN = rup4(query_cols().n - Q_PRE)


def query_program(sh):
    m = query_cols()
    cons = Cons()
    x = eb(pscale(pv(m["XQ"]), GEN))
    one = eb(pv(QP_ACT))
    cons.ext(O.SEL_ALL, esub(emul(esub(x, ev(m["ZETA"])), ev(m["I1"])), one))
    cons.ext(O.SEL_ALL, esub(emul(esub(x, ev(m["ZNX"])), ev(m["I2"])), one))
    cons.ext(O.SEL_ALL, esub(ev(m["P1"]), emul(esub(ev(m["AZ"]), ev(m["YZ"])), ev(m["I1"]))))
    cons.ext(O.SEL_ALL, esub(ev(m["P2"]), emul(esub(ev(m["AN"]), ev(m["YN"])), ev(m["I2"]))))
    cons.ext(O.SEL_ALL, esub(ev(m["RO"]), eadd(ev(m["P1"]), ev(m["P2"]))))
    # the rows of a query carry its index: what the fold chain hands a LOWER height is named by the query (two heights of two queries cannot be exchanged)
    cons.add(O.SEL_ALL, pmul(pv(QP_TOP), padd(pv(m["IDX0"]), pneg(pv(m["IDX"])))))
    cons.add(O.SEL_TRANSITION, pmul(pv(QP_NQ, True), padd(pv(m["IDX0"], True), pneg(pv(m["IDX0"])))))
    return O.air_program(Q_PRE + Q_MAIN, sh.NP * sh.NPUB, cons.c)


def query_table():
    m = query_cols()
    return O.interaction_table([
        (RECV, QP_TOP, F.BUS_I, [QP_QNS, m["IDX"]]),
        (RECV, QP_TOP, F.BUS_Q, [QP_KEY, m["IDX"], m["XQ"]] + _e4(m["RO"])),
        (RECV, QP_LOW, F.BUS_Q, [QP_KEY, m["IDX0"], m["IDX"], m["XQ"]] + _e4(m["RO"])),
        (RECV, QP_ACT, BUS_AH0, [QP_KAH] + _e4(m["AZ"])), (RECV, QP_ACT, BUS_AH1, [QP_KAH] + _e4(m["AN"])),
        (RECV, QP_ACT, BUS_YH0, [QP_KYH] + _e4(m["YZ"])), (RECV, QP_ACT, BUS_YH1, [QP_KYH] + _e4(m["YN"])),
        (RECV, QP_ACT, BUS_ZH0, [QP_KZH] + _e4(m["ZETA"])), (RECV, QP_ACT, BUS_ZH1, [QP_KZH] + _e4(m["ZNX"])),
        (SEND, QP_NQI, BUS_QI, [QP_QN, m["IDX"]])])


def query_rows(sh):
    return [(p, q, h) for p in range(sh.NP) for q in range(sh.Q) for h in sh.hs]


def query_pre(sh, log_rows):
    t = np.zeros((1 << log_rows, Q_PRE), dtype=np.uint32)
    for i, (p, q, h) in enumerate(query_rows(sh)):
        r = t[i]
        r[QP_ACT], r[QP_KEY], r[QP_KAH], r[QP_KYH], r[QP_KZH] = 1, p * sh.NTREES + (sh.H - h), ah_key(sh, p, q, h), p * 32 + h, p * 32 + h
        r[QP_TOP], r[QP_QNS], r[QP_LOW], r[QP_NQ] = int(h == sh.H), p * sh.Q + q, int(h != sh.H), int(h != sh.H)
        r[QP_QN], r[QP_NQI] = (p * sh.Q + q) * 32 + h, sum(1 for tr in sh.trees if sh.tree_hs[tr][0] == h)
    return t


def query_main(sh, ws, scs, ahs, yhs, log_rows):
    m = query_cols()
    t = np.zeros((1 << log_rows, Q_MAIN), dtype=np.uint64)
    for i, (p, q, h) in enumerate(query_rows(sh)):
        w, r = ws[p], t[i]
        qv = w["queries"][q]
        ic = qv["index"] >> (sh.H - h)
        xq = pow(two_adic_generator(h), pyref.bitrev(ic, h), P)
        x = [GEN * xq % P, 0, 0, 0]
        c0 = sh.lh.index(h)
        zeta, znx = scs[p][c0]["ZETA"], scs[p][c0]["ZNX"]
        az, an = ahs[p][(q, h)]
        yz, yn = yhs[p][h]
        i1, i2 = pyref.ext_inv(e_sub(x, zeta)), pyref.ext_inv(e_sub(x, znx))
        p1, p2 = ext_mul(e_sub(az, yz), i1), ext_mul(e_sub(an, yn), i2)
        ro = e_add(p1, p2)
        assert ro == list(qv["roh"][h]), "a height's reduced opening is not the verifier's"
        r[m["IDX"] - Q_PRE], r[m["XQ"] - Q_PRE], r[m["IDX0"] - Q_PRE] = ic, xq, qv["index"]
        for name, val in (("RO", ro), ("AZ", az), ("AN", an), ("YZ", yz), ("YN", yn), ("ZETA", zeta), ("ZNX", znx), ("I1", i1), ("I2", i2), ("P1", p1), ("P2", p2)):
            r[m[name] - Q_PRE:m[name] - Q_PRE + 4] = val
    return t.astype(np.uint32)


# ---------------------------------------------------------------------------------------------------------------- FOLD
def inject_layers(sh):
    """the layers at whose row a shorter height's reduced opening joins: the fold chain reaches 2^h entries at layer H - h"""
    return sorted(sh.H - h for h in sh.hs if h != sh.H)


def fold_table(sh):
    INJ, INJF = F.inj_cols(sh.R)
    t = [(SEND, F.ACTIVE, F.BUS_E0, [F.LNX, F.K2] + _e4(F.E0)), (SEND, F.ACTIVE, F.BUS_E1, [F.LNX, F.K2] + _e4(F.E1)),
         (SEND, F.L_REC, F.BUS_Q, [F.PT, F.IDX, F.XS] + _e4(F.OWN)), (SEND, INJF, F.BUS_Q, [F.LNX, INJF + 1, F.IDX, F.XS] + _e4(INJ)),
         (RECV, F.ACTIVE, BUS_BETA, [F.LNX] + _e4(F.BETA)), (SEND, F.L_REC + sh.R - 1, BUS_FIN, [F.PT] + _e4(F.FOLD))]
    return O.interaction_table(t)


def fold_main(sh, ws, log_rows):
    inj = inject_layers(sh)
    t = np.zeros((1 << log_rows, F.width_of(sh.R, rec=True, inject=inj)), dtype=np.uint32)
    t[:, F.T] = 1
    per = sh.Q * sh.R
    for p, w in enumerate(ws):
        view = {"betas": w["betas"], "queries": [(qv["index"], qv["roh"].get(sh.H, [0] * 4), qv["sibs"]) for qv in w["queries"]]}
        injq = [{sh.H - h: list(qv["roh"][h]) for h in sh.hs if h != sh.H} for qv in w["queries"]]
        one, final = F.trace(view, lg(per), wired=True, rec=True, pt=p * sh.NTREES, inject=injq)
        assert list(final) == list(w["final"])
        t[p * per:(p + 1) * per] = one[:per]
    return t


# ---------------------------------------------------------------------------------------------------------------- the machine
CHIPS = ("P2R", "ROWSUM", "FOLD", "TS", "QUERY", "OPENED", "SAMPLES", "SCALARS", "EVAL", "LOGUP")


def heights(sh):
    n = sh.NP
    return {"P2R": lg(n * sh.p2_rows), "ROWSUM": lg(n * len(rowsum_rows(sh))), "FOLD": lg(n * sh.Q * sh.R), "TS": lg(n * sh.NTS), "QUERY": lg(n * sh.Q * len(sh.hs)),
            "OPENED": lg(n * sh.NV // 2), "SAMPLES": lg(n * sh.NS), "SCALARS": lg(n * sh.C), "EVAL": lg(n * len(sh.terms)), "LOGUP": lg(n * len(sh.lrows))}


def order(sh):
    h = heights(sh)
    return sorted(CHIPS, key=lambda c: (-h[c], CHIPS.index(c)))


def programs(sh):
    npub = sh.NP * sh.NPUB
    return {"P2R": p2r_program(sh), "ROWSUM": rowsum_program(sh), "FOLD": F.program(sh.R, wired=True, transcript=True, rec=npub, inject=inject_layers(sh)), "TS": ts_program(sh),
            "QUERY": query_program(sh), "OPENED": opened_program(sh), "SAMPLES": F.samples_program(sh.R, sh.Q, sh.PB, npub), "SCALARS": scalars_program(sh),
            "EVAL": eval_program(sh), "LOGUP": logup_program(sh)}


def tables(sh):
    M0 = F.S_PRE
    s_tab = O.interaction_table([(RECV, F.S_ROW, F.BUS_S0, [F.S_C] + [M0 + F.S_W + j for j in range(4)]), (RECV, F.S_ROW, F.BUS_S1, [F.S_C] + [M0 + F.S_W + j for j in range(4, 8)])]
                                + [(SEND, F.S_ACT + j, F.BUS_I, [F.S_KQ + j, M0 + F.S_IDX + j]) for j in range(8)])
    return {"P2R": p2r_table(), "ROWSUM": rowsum_table(), "FOLD": fold_table(sh), "TS": ts_table(sh), "QUERY": query_table(), "OPENED": opened_table(), "SAMPLES": s_tab,
            "SCALARS": scalars_table(sh), "EVAL": eval_table(), "LOGUP": logup_table()}


def preprocessed(sh):
    from recursion_air import samples_stacked
    h = heights(sh)
    spre, _, _ = samples_stacked(sh, [[[0] * 8] * sh.NS] * sh.NP, h["SAMPLES"])
    return {"P2R": p2r_pre(sh, h["P2R"]), "ROWSUM": rowsum_pre(sh, h["ROWSUM"]), "FOLD": None, "TS": ts_pre(sh, h["TS"]), "QUERY": query_pre(sh, h["QUERY"]),
            "OPENED": opened_pre(sh, h["OPENED"]), "SAMPLES": spre, "SCALARS": scalars_pre(sh, h["SCALARS"]), "EVAL": eval_pre(sh, h["EVAL"]), "LOGUP": logup_pre(sh, h["LOGUP"])}


def main_traces(sh, ws):
    from recursion_air import samples_stacked
    h = heights(sh)
    p2, samples, chals = p2r_main(sh, ws, h["P2R"])
    for w, chal in zip(ws, chals):
        assert chal[sh.TG] == (w["gamma"], w["beta"]) and chal[sh.TA][0] == w["alpha"] and chal[sh.TQ][0] == w["zeta"] and chal[sh.TF][0] == w["fa"], "the sponge rows do not reproduce the challenges"
        assert [chal[sh.TL0 + l][0] for l in range(sh.R)] == w["betas"]
    scs = [scalars_values(sh, w) for w in ws]
    evl, accs = eval_main(sh, ws, scs, h["EVAL"])
    lgu, outs = logup_main(sh, ws, scs, accs, h["LOGUP"])
    for sc, out in zip(scs, outs):
        for c in range(sh.C):
            assert out[c] == sc[c]["ACC"], "chip %d: the constraints do not match the quotient at zeta" % c
    opened, yhs, pws = opened_main(sh, ws, h["OPENED"])
    rs, ahs = rowsum_main(sh, ws, pws, h["ROWSUM"])
    qm = query_main(sh, ws, scs, ahs, yhs, h["QUERY"])
    fold = fold_main(sh, ws, h["FOLD"])
    _, smain, drawn = samples_stacked(sh, samples, h["SAMPLES"])
    assert drawn == [[qv["index"] for qv in w["queries"]] for w in ws], "the query indices are not the ones the transcript draws"
    ts = ts_main(sh, ws, chals, p2, h["TS"])
    return {"P2R": p2, "ROWSUM": rs, "FOLD": fold, "TS": ts, "QUERY": qm, "OPENED": opened, "SAMPLES": smain, "SCALARS": scalars_main(sh, scs, h["SCALARS"]), "EVAL": evl, "LOGUP": lgu}


def machine(chips, key_root, proofs, public_values, n_queries, pow_bits):
    """chips: the inner machine ([{ln, W, Pw, prog, tab}], tallest first); proofs: a LIST of its version-11 proofs (bytes) with a list of public-value lists
    -> (shape, main traces, preprocessed traces, programs, interaction tables, public values), chips tallest first"""
    sh = MShape(chips, key_root, n_queries, pow_bits, len(public_values[0]), len(proofs))
    build_reads(sh)
    progs_in, tabs_in = [c["prog"] for c in chips], [c["tab"] for c in chips]
    ws = [witness(sh, pr, pv_, progs_in, tabs_in, key_root) for pr, pv_ in zip(proofs, public_values)]
    names = order(sh)
    mt, pre, prog, tab = main_traces(sh, ws), preprocessed(sh), programs(sh), tables(sh)
    return sh, [mt[c] for c in names], [pre[c] for c in names], [prog[c] for c in names], [tab[c] for c in names], [int(v) % P for pv_ in public_values for v in pv_]
