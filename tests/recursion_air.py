"""The SHARD VERIFIER as a machine, written a second time (the first is zktls_amd/csrc/recursion.hip): a keyed machine of eight chips that
checks a WHOLE shard proof of this library (version 1: the synthetic AIR, SP1 shape -- blowup 2, fold by 2, constant final value, Poseidon2
width 16) in-circuit: the transcript from the header to the last query index, the AIR identity at zeta, every Merkle opening, the reduced
openings and the FRI folds.  What the reference asks for behind `client.prove(.., Groth16)` (crates/guest-prover-sp1/src/sp1.rs:116: core ->
COMPRESS verifies the shard proofs; RISC Zero: lift -> join, prover.rs:90).

Every structural fact -- which row absorbs what, which path belongs to which query, which words are constants of the shape -- is a
PREPROCESSED column: the key (the commitment to them) depends on the inner proof's SHAPE (log_n, width, queries, proof-of-work bits,
number of public values) and on nothing else, and the verifier of the outer proof is handed the key and the inner proof's PUBLIC VALUES --
no byte of the inner proof.  docs/PROTOCOL.md section 3c describes the chips; tests/pyverify.py is the verifier that is restated.

Chips, tallest first for the shapes of the tests (machine() sorts by height):
  P2R      one Poseidon2 permutation per row.  The sponge rows of the transcript first, then per query the FRI layer paths, then the
           trace opening (the opened row hashed by sponge rows, then its path), then the quotient opening.
  ROWSUM   one row per 8 values of an opened row: hands them to the P2R sponge rows and accumulates sum_j fa^j row[j] (Horner).
  FOLD     tests/fri_air.py's fold chip in its `rec` form.
  TS       the transcript as a table: one row per absorbing sponge row -- the observed words, the challenge sampled behind it.
  QUERY    one row per query: index, point, the reduced opening from ROWSUM's sums and the opened values' sums.
  OPENED   one row per column group (a, b, c, d) of the synthetic AIR: the opened values at zeta and zeta g, their fa-weighted sums, the
           AIR's constraints folded with alpha.
  SAMPLES  tests/fri_air.py's chip: the bits of the sampled words (proof of work, query indices).
  SCALARS  the verifier's scalars: zeta^N, the selectors, powers of fa, the quotient recombination, the AIR identity."""
import numpy as np

import fri_air as F
import oracle_lib as O
import poseidon2_air as P2
import pyref
from pyref import P, ext_mul, two_adic_generator

V = O.air_var
GEN, EXT_W = 31, 11
SEND, RECV = O.SEND, O.RECEIVE
# buses (F.BUS_E0 / E1 / R0 / R1 / Q / S0 / S1 / I keep their meaning from tests/fri_air.py)
BUS_IN0, BUS_IN1, BUS_TC, BUS_BETA, BUS_SC, BUS_QI, BUS_AT, BUS_AQ = 61, 62, 63, 64, 65, 66, 67, 68
BUS_K0, BUS_KFA, BUS_KO0, BUS_OY, BUS_OA = 70, 77, 78, 83, 85               # K0 .. K0 + 6: the QUERY chip's seven constants; KO0 .. KO0 + 4: the OPENED chip's five; OY, OY + 1
BUS_FIN = F.BUS_FIN
BUS_VAL, BUS_EA = 86, 87                                                        # air mode: (key, value at zeta) to the EVAL chip's factor slots; the proof's alpha


# ---------------------------------------------------------------------------------------------------------------- polynomials over columns
# a polynomial = a list of (coefficient, [variables]); an extension expression = four of them (x^4 = 11).  No merging of like terms: the order
# in which terms are produced IS the program (csrc/recursion.hip produces them in the same order; the words are compared).
def pc(c):
    return [(c % P, [])] if c % P else []


def pv(col, nxt=False):
    return [(1, [V(col, nxt)])]


def padd(*ps):
    out = []
    for p in ps:
        out += p
    return out


def pscale(p, k):
    return [(c * k % P, vs) for c, vs in p if c * k % P]


def pneg(p):
    return pscale(p, P - 1)


def pmul(a, b):
    return [(ca * cb % P, va + vb) for ca, va in a for cb, vb in b if ca * cb % P]


def ev(col, nxt=False):
    return [pv(col + i, nxt) for i in range(4)]


def ec(c):
    """an extension constant (4 ints), or a base constant"""
    c = list(c) if isinstance(c, (list, tuple)) else [c, 0, 0, 0]
    return [pc(x) for x in c]


def eb(p):
    """a base-field polynomial as an extension expression"""
    return [p, [], [], []]


def eadd(*es):
    return [padd(*[e[i] for e in es]) for i in range(4)]


def esub(a, b):
    return [padd(a[i], pneg(b[i])) for i in range(4)]


def escale(a, k):
    return [pscale(a[i], k) for i in range(4)]


def emul(a, b):
    out = [[], [], [], []]
    for j in range(4):
        for i in range(4):
            for k in range(4):
                if (i + k) % 4 != j:
                    continue
                t = pmul(a[i], b[k])
                out[j] += pscale(t, EXT_W) if i + k >= 4 else t
    return out


def egate(flag, e):
    """flag * e, flag a polynomial"""
    return [pmul(flag, e[i]) for i in range(4)]


class Cons:
    def __init__(self):
        self.c = []

    def add(self, sel, poly):
        self.c.append((sel, [(c % P, list(vs)) for c, vs in poly if c % P]))

    def ext(self, sel, e):
        for i in range(4):
            self.add(sel, e[i])


class Cols:
    """a running column allocator; names -> first column"""
    def __init__(self, start=0):
        self.n = start
        self.at = {}

    def __call__(self, name, width=4):
        self.at[name] = self.n
        self.n += width
        return self.at[name]

    def __getitem__(self, name):
        return self.at[name]


def rup4(n):
    return (n + 3) & ~3


def lg(n, lo=5):
    l = lo
    while (1 << l) < n:
        l += 1
    return l


# ---------------------------------------------------------------------------------------------------------------- the shape of an inner proof
class Shape:
    """everything the machine's structure depends on: (log_n, width, queries, pow_bits, n_public) of a version-1 shard proof, and how many
    such proofs ONE outer proof verifies (n_proofs: the join -- every chip holds the rows of proof 0, then those of proof 1, ...; tags, tree
    numbers and query numbers carry the proof's number)"""
    def __init__(self, log_n, width, n_queries, pow_bits, n_public, n_proofs=1, program=None):
        assert width % 8 == 0 and width >= 8 and 2 <= log_n <= 22 and n_queries >= 1 and 1 <= n_proofs <= 1024
        self.n, self.W, self.Q, self.PB, self.NPUB, self.NP = log_n, width, n_queries, pow_bits, n_public, n_proofs
        self.R, self.H, self.G, self.WB = log_n, log_n + 1, width // 4, width // 8
        self.head = [log_n, width, 1, n_queries, pow_bits, n_public]           # the header words the transcript observes
        # AIR MODE (program given): the inner proofs are version-7 proofs of that constraint program.  Their transcript starts from 18 words
        # (+ logup_pairs 0, fold 2^1, constant final value, Poseidon2 width 16, the program's 8-word digest), and the program is flattened
        # into TERMS of exactly three factor keys (0 = the constant one; a constraint's selector is one more factor) for the EVAL chip
        self.air = program is not None
        self.HL = 6
        if self.air:
            prog = [int(x) for x in program]
            assert prog[2] == width and prog[4] == n_public
            self.head += [0, 1, 0, 16] + [int(x) for x in O.air_digest(np.array(prog, dtype=np.uint32))]
            self.HL = 18
            self.KSPAN = 1 + 2 * width + n_public + 3
            self.mult = [0] * self.KSPAN
            self.terms = []                                                     # (coefficient, [three keys], first term of its constraint)
            at = 6
            for _ in range(prog[3]):
                sel, nt = prog[at], prog[at + 1]
                at += 2
                for t in range(nt):
                    coeff, d = prog[at], prog[at + 1]
                    at += 2
                    keys = []
                    for v in prog[at:at + d]:
                        kind, idx = v >> 30, v & 0xffff
                        keys.append(self.key_local(idx) if kind == 0 else (self.key_next(idx) if kind == 1 else self.key_pub(idx)))
                    at += d
                    if sel:
                        keys.append(self.key_sel(sel - 1))                      # program selectors: 1 first row, 2 last row, 3 transition
                    assert len(keys) <= 3
                    keys += [0] * (3 - len(keys))
                    for k in keys:
                        self.mult[k] += 1
                    self.terms.append((coeff, keys, 1 if t == 0 else 0))
        n0 = self.HL + 8 + n_public
        self.f0, self.r0 = n0 // 8, n0 % 8
        self.TA = self.f0 if self.r0 else self.f0 - 1                          # the sponge row alpha is sampled behind
        self.TQ = self.TA + 1                                                  # absorbs the quotient root; zeta
        self.TO0 = self.TQ + 1                                                 # first block of the opened values (W / 2 blocks at zeta, W / 2 at zeta g, 4 quotient)
        self.TF = self.TO0 + width + 3                                         # the last of them; fa
        self.TL0 = self.TF + 1                                                 # layer roots; beta_l
        self.TP = self.TL0 + self.R                                            # final value + witness; the proof-of-work word and 7 index words
        self.NS = F.sample_rows(n_queries)
        self.NT = self.TP + self.NS                                            # sponge rows of the transcript
        self.NTS = self.TP + 1                                                 # rows of the TS table (the absorbing ones)
        self.pub_rows = sorted({(self.HL + 8 + i) // 8 for i in range(n_public)})       # TS rows that hold public values
        # P2R row layout
        self.fri_rows = self.R + self.R * (self.R + 1) // 2                    # per query: a leaf row + the path, every layer
        self.p2_fri0 = self.NT
        self.p2_tr0 = self.p2_fri0 + self.Q * self.fri_rows                    # trace openings: WB sponge rows + H path rows per query
        self.p2_q0 = self.p2_tr0 + self.Q * (self.WB + self.H)                 # quotient openings: 1 sponge row + H path rows
        self.p2_rows = self.p2_q0 + self.Q * (1 + self.H)
        self.tag0 = self.NT                                                    # ROWSUM tags follow the transcript's
        self.TAGSPAN = self.NT + self.Q * (self.WB + 1)                        # tags of one proof
        self.TREES = self.R + 2                                                # trees of one proof: the FRI layers, the trace, the quotient

    def ttag(self, p, T):
        return p * self.TAGSPAN + T

    # keys of the values the EVAL chip's factor slots read, inside one proof's key space (proof p: + p KSPAN)
    def key_local(self, c):
        return 1 + c

    def key_next(self, c):
        return 1 + self.W + c

    def key_pub(self, i):
        return 1 + 2 * self.W + i

    def key_sel(self, which):
        return 1 + 2 * self.W + self.NPUB + which                              # 0 first row, 1 last row, 2 transition

    def row_tag(self, p, q, b):
        return p * self.TAGSPAN + self.tag0 + q * (self.WB + 1) + b            # b = WB: the quotient row


def absorbed(sh, T):
    """how many rate words sponge row T absorbs (8: all; the others keep the previous output there)"""
    if T < sh.f0 or sh.TQ <= T < sh.TP:
        return 8
    if T == sh.f0 and sh.r0:
        return sh.r0
    if T == sh.TP:
        return 5
    return 0


# ---------------------------------------------------------------------------------------------------------------- P2R
P2_PRE = 24
PP_SS, PP_SPG, PP_CH, PP_END, PP_K, PP_RIN, PP_TAG, PP_SROOT, PP_TREE, PP_SCH, PP_SSMP, PP_QIDX, PP_QN, PP_RPAIR = 0, 1, 2, 3, 4, 12, 13, 14, 15, 16, 17, 18, 19, 20
P2_MAIN = 360
M_KP = 352                                                                      # main columns: P2.IN .. P2.SP, P2.D (343), P2.BIT (351), KP (352)


def p2r_program(sh):
    M0 = P2_PRE
    cons = Cons()
    for sel, terms in P2.permutation_constraints():
        cons.add(sel, [(c, [v + M0 if (v >> 30) == 0 else v for v in vs]) for c, vs in terms])
    IN, OUT, D, BIT, KP = M0 + P2.IN, M0 + P2.OUTE(7), M0 + P2.D, M0 + P2.BIT, M0 + M_KP
    for j in range(8):
        cons.add(O.SEL_ALL, padd(pv(D + j), pneg(pv(IN + j)), pmul(pv(BIT), pv(IN + j)), pneg(pmul(pv(BIT), pv(IN + 8 + j)))))
    cons.add(O.SEL_ALL, padd(pmul(pv(BIT), pv(BIT)), pneg(pv(BIT))))
    cons.add(O.SEL_ALL, pmul(padd(pv(PP_SS), pv(PP_SPG)), pv(BIT)))             # leaf and transcript rows: no direction
    for j in range(8):
        cons.add(O.SEL_ALL, pmul(pv(PP_SS), pv(IN + 8 + j)))                     # a sponge starts with the zero capacity
    for j in range(8):
        cons.add(O.SEL_TRANSITION, pmul(pv(PP_SPG, True), padd(pv(IN + 8 + j, True), pneg(pv(OUT + 8 + j)))))
    for j in range(8):
        cons.add(O.SEL_TRANSITION, pmul(pv(PP_CH, True), padd(pv(D + j, True), pneg(pv(OUT + j)))))
    for j in range(8):
        cons.add(O.SEL_TRANSITION, pmul(pv(PP_K + j, True), padd(pv(IN + j, True), pneg(pv(OUT + j)))))
    cons.add(O.SEL_TRANSITION, pmul(pv(PP_CH, True), padd(pv(KP), pscale(pv(KP, True), P - 2), pneg(pv(BIT)))))
    cons.add(O.SEL_ALL, pmul(pv(PP_END), padd(pv(KP), pneg(pv(BIT)))))
    return O.air_program(P2_PRE + P2_MAIN, sh.NP * sh.NPUB, cons.c)


def p2r_table():
    M0, o = P2_PRE, P2_PRE + P2.OUTE(7)
    IN, KP = M0 + P2.IN, M0 + M_KP
    return O.interaction_table([
        (RECV, PP_RIN, BUS_IN0, [PP_TAG, IN, IN + 1, IN + 2, IN + 3]), (RECV, PP_RIN, BUS_IN1, [PP_TAG, IN + 4, IN + 5, IN + 6, IN + 7]),
        (RECV, PP_RPAIR, F.BUS_E0, [PP_TREE, KP, IN, IN + 1, IN + 2, IN + 3]), (RECV, PP_RPAIR, F.BUS_E1, [PP_TREE, KP, IN + 4, IN + 5, IN + 6, IN + 7]),
        (SEND, PP_SROOT, F.BUS_R0, [PP_TREE, o, o + 1, o + 2, o + 3]), (SEND, PP_SROOT, F.BUS_R1, [PP_TREE, o + 4, o + 5, o + 6, o + 7]),
        (SEND, PP_SCH, BUS_TC, [PP_TAG, o + 7, o + 6, o + 5, o + 4]),
        (SEND, PP_SSMP, F.BUS_S0, [PP_TAG, o + 7, o + 6, o + 5, o + 4]), (SEND, PP_SSMP, F.BUS_S1, [PP_TAG, o + 3, o + 2, o + 1, o]),
        (RECV, PP_QIDX, BUS_QI, [PP_QN, KP])])


def p2r_pre(sh, log_rows):
    """the structure of the chip: fixed by the shape"""
    t = np.zeros((1 << log_rows, P2_PRE), dtype=np.uint32)
    for p in range(sh.NP):
        _p2r_pre_one(sh, t[p * sh.p2_rows:(p + 1) * sh.p2_rows], p)
    return t


def _p2r_pre_one(sh, t, p):
    for T in range(sh.NT):
        r = t[T]
        k = absorbed(sh, T)
        if T == 0:
            r[PP_SS] = 1
        else:
            r[PP_SPG] = 1
            for j in range(k, 8):
                r[PP_K + j] = 1
        r[PP_TAG] = sh.ttag(p, T)
        if k:
            r[PP_RIN] = 1
        if T in (sh.TA, sh.TQ, sh.TF) or sh.TL0 <= T < sh.TP:
            r[PP_SCH] = 1
        if T >= sh.TP:
            r[PP_SSMP] = 1
    row = sh.p2_fri0
    for q in range(sh.Q):
        for l in range(sh.R):
            t[row, PP_SS], t[row, PP_RPAIR], t[row, PP_TREE] = 1, 1, p * sh.TREES + l
            row += 1
            depth = sh.H - (l + 1)
            for lvl in range(depth):
                t[row, PP_CH], t[row, PP_TREE] = 1, p * sh.TREES + l
                if lvl == depth - 1:
                    t[row, PP_END] = t[row, PP_SROOT] = 1
                row += 1
    assert row == sh.p2_tr0
    for tree, blocks in ((sh.R, sh.WB), (sh.R + 1, 1)):
        for q in range(sh.Q):
            for b in range(blocks):
                t[row, PP_SS if b == 0 else PP_SPG] = 1
                t[row, PP_RIN], t[row, PP_TAG] = 1, sh.row_tag(p, q, b if tree == sh.R else sh.WB)
                row += 1
            for lvl in range(sh.H):
                t[row, PP_CH], t[row, PP_TREE] = 1, p * sh.TREES + tree
                if lvl == 0:
                    t[row, PP_QIDX], t[row, PP_QN] = 1, p * sh.Q + q
                if lvl == sh.H - 1:
                    t[row, PP_END] = t[row, PP_SROOT] = 1
                row += 1
    assert row == sh.p2_rows


def _p2row(state, bit=0, kp=0):
    r, out = P2.row(state, bit)
    return r[:M_KP] + [kp % P] + [0] * 7, out                                    # (the old chip's flag columns behind BIT are not this chip's)


def p2r_main(sh, ws, log_rows):
    """ws = the witnesses, one per inner proof (see witness()) -> (main trace, per proof: sampled words [NS][8], challenges {T: out[7..4]})"""
    rows, samples, chals = [], [], []
    for w in ws:
        r, s, c = _p2r_main_one(sh, w)
        rows += r
        samples.append(s)
        chals.append(c)
    pad, _ = _p2row([0] * 16)
    rows += [pad] * ((1 << log_rows) - len(rows))
    return np.array(rows, dtype=np.uint64).astype(np.uint32), samples, chals


def _p2r_main_one(sh, w):
    rows, chal, samples = [], {}, []
    state = [0] * 16
    for T in range(sh.NT):
        k = absorbed(sh, T)
        blk = w["blocks"].get(T, [])
        state = [blk[j] if j < k else state[j] for j in range(8)] + (state[8:] if T else [0] * 8)
        r, out = _p2row(state)
        rows.append(r)
        chal[T] = [out[7], out[6], out[5], out[4]]
        if T >= sh.TP:
            samples.append([out[7 - j] for j in range(8)])
        state = list(out)
    for q in range(sh.Q):
        for l in range(sh.R):
            k, pair, sibs = w["fri"][q][l]
            r, out = _p2row(list(pair) + [0] * 8, 0, 2 * k)
            rows.append(r)
            digest = out[:8]
            for lvl, sib in enumerate(sibs):
                bit = (k >> lvl) & 1
                r, out = _p2row(list(sib) + digest if bit else digest + list(sib), bit, k >> lvl)
                rows.append(r)
                digest = out[:8]
            assert digest == w["layer_roots"][l], "a FRI layer path does not end in the layer's root"
    for key_row, key_path, root in (("trow", "tpath", w["trace_root"]), ("qrow", "qpath", w["quot_root"])):
        for q in range(sh.Q):
            op = w["openings"][q]
            vals, cap, index = op[key_row], [0] * 8, op["index"]
            nb = len(vals) // 8
            for b in range(nb):
                r, out = _p2row(list(vals[8 * b:8 * b + 8]) + cap, 0, 2 * index if b == nb - 1 else 0)
                rows.append(r)
                cap = out[8:]
            digest = out[:8]
            for lvl, sib in enumerate(op[key_path]):
                bit = (index >> lvl) & 1
                r, out = _p2row(list(sib) + digest if bit else digest + list(sib), bit, index >> lvl)
                rows.append(r)
                digest = out[:8]
            assert digest == root, "an opening does not end in its root"
    assert len(rows) == sh.p2_rows
    return rows, samples, chal


# ---------------------------------------------------------------------------------------------------------------- TS
def ts_cols(sh):
    c = Cols()
    for name, w in (("T", 1), ("ACT", 1), ("NSEND", 1), ("CF", 8), ("CV", 8), ("IND0", 1), ("IP", sh.NP * len(sh.pub_rows)), ("NROOT", 1), ("NTR", 1), ("TREE", 1),
                    ("HASCH", 1), ("NBETA", 1), ("NSC", 1), ("KIND", 1), ("NFIN", 1), ("PT", 1)):
        c(name, w)
    if sh.air:                                                                  # the public values go to the EVAL chip word by word: key and multiplicity per word, a zero column
        c("PK", 8), c("PM", 8), c("Z", 1)
    pre = rup4(c.n)
    m = Cols(pre)
    m("W", 8), m("TR", 8), m("CH", 4)
    return c, m, pre


TS_MAIN = 20


def ts_program(sh):
    c, m, pre = ts_cols(sh)
    cons = Cons()
    W, TR = m["W"], m["TR"]
    for j in range(8):
        cons.add(O.SEL_ALL, pmul(pv(c["CF"] + j), padd(pv(W + j), pneg(pv(c["CV"] + j)))))
    npr = len(sh.pub_rows)
    for p in range(sh.NP):                                                      # the outer proof's public values: those of proof 0, then those of proof 1, ...
        for i in range(sh.NPUB):
            pos = sh.HL + 8 + i
            cons.add(O.SEL_ALL, pmul(pv(c["IP"] + p * npr + sh.pub_rows.index(pos // 8)), padd(pv(W + pos % 8), [(P - 1, [V(p * sh.NPUB + i, public=True)])])))
    o = sh.HL % 8                                                               # the trace root: words HL .. HL + 7 of the transcript
    for j in range(8 - o):
        cons.add(O.SEL_ALL, pmul(pv(c["IND0"]), padd(pv(W + o + j), pneg(pv(TR + j)))))
    for j in range(o):
        cons.add(O.SEL_TRANSITION, pmul(pv(c["IND0"]), padd(pv(W + j, True), pneg(pv(TR + 8 - o + j)))))
    return O.air_program(pre + TS_MAIN, sh.NP * sh.NPUB, cons.c)


def ts_table(sh):
    c, m, _ = ts_cols(sh)
    W, TR, CH = m["W"], m["TR"], m["CH"]
    rows = [
        (SEND, c["NSEND"], BUS_IN0, [c["T"], W, W + 1, W + 2, W + 3]), (SEND, c["NSEND"], BUS_IN1, [c["T"], W + 4, W + 5, W + 6, W + 7]),
        (RECV, c["HASCH"], BUS_TC, [c["T"], CH, CH + 1, CH + 2, CH + 3]),
        (SEND, c["NBETA"], BUS_BETA, [c["TREE"], CH, CH + 1, CH + 2, CH + 3]),
        (SEND, c["NSC"], BUS_SC, [c["KIND"], CH, CH + 1, CH + 2, CH + 3]),
        (RECV, c["NROOT"], F.BUS_R0, [c["TREE"], W, W + 1, W + 2, W + 3]), (RECV, c["NROOT"], F.BUS_R1, [c["TREE"], W + 4, W + 5, W + 6, W + 7]),
        (RECV, c["NTR"], F.BUS_R0, [c["TREE"], TR, TR + 1, TR + 2, TR + 3]), (RECV, c["NTR"], F.BUS_R1, [c["TREE"], TR + 4, TR + 5, TR + 6, TR + 7]),
        (RECV, c["NFIN"], BUS_FIN, [c["PT"], W, W + 1, W + 2, W + 3])]
    if sh.air:
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
            for j in range(8):
                if 8 * T + j < sh.HL:
                    r[c["CF"] + j], r[c["CV"] + j] = 1, sh.head[8 * T + j]
            if T in sh.pub_rows:
                r[c["IP"] + p * npr + sh.pub_rows.index(T)] = 1
            if sh.air:
                for j in range(8):
                    i = 8 * T + j - (sh.HL + 8)
                    if 0 <= i < sh.NPUB:
                        r[c["PK"] + j], r[c["PM"] + j] = p * sh.KSPAN + sh.key_pub(i), sh.mult[sh.key_pub(i)]
            if T == sh.HL // 8:
                r[c["IND0"]], r[c["NTR"]], r[c["TREE"]] = 1, sh.Q, p * sh.TREES + sh.R
            if T == sh.TQ:
                r[c["NROOT"]], r[c["TREE"]] = sh.Q, p * sh.TREES + sh.R + 1
            if sh.TL0 <= T < sh.TP:
                r[c["NROOT"]], r[c["TREE"]], r[c["NBETA"]] = sh.Q, p * sh.TREES + T - sh.TL0, sh.Q
            for kind, Tk in enumerate((sh.TA, sh.TQ, sh.TF)):
                if T == Tk:
                    r[c["NSC"]], r[c["KIND"]] = 1, 3 * p + kind
            if T in (sh.TA, sh.TQ, sh.TF) or sh.TL0 <= T < sh.TP:
                r[c["HASCH"]] = 1
            if T == sh.TP:
                r[c["NFIN"]], r[c["PT"]] = sh.Q, p * sh.TREES
    return t


def ts_main(sh, ws, chals, p2_main, log_rows):
    t = np.zeros((1 << log_rows, TS_MAIN), dtype=np.uint32)
    for p, (w, chal) in enumerate(zip(ws, chals)):
        for T in range(sh.NTS):
            r = t[p * sh.NTS + T]
            r[0:8] = p2_main[p * sh.p2_rows + T, P2.IN:P2.IN + 8]               # the absorbed words, and whatever the kept ones are
            if T in (sh.TA, sh.TQ, sh.TF) or sh.TL0 <= T < sh.TP:
                r[16:20] = chal[T]
        t[p * sh.NTS + sh.HL // 8, 8:16] = w["trace_root"]
    return t


# ---------------------------------------------------------------------------------------------------------------- ROWSUM
RP_TAG, RP_ACT, RP_NOTFIRST, RP_LAST0, RP_LAST1, RP_QN, RP_FIRST, RP_PID = 0, 1, 2, 3, 4, 5, 6, 7
RS_PRE = 12
RP_NFC = 8                                                                       # the row is active and not the first of its proof: the constants stay
RS_V, RS_ACCIN, RS_T, RS_FA, RS_MAIN = 0, 8, 12, 44, 48


def rowsum_program(sh):
    M0 = RS_PRE
    cons = Cons()
    fa = ev(M0 + RS_FA)
    cons.ext(O.SEL_TRANSITION, egate(pv(RP_NFC, True), esub(ev(M0 + RS_FA, True), fa)))
    prev = ev(M0 + RS_ACCIN)
    for s in range(7, -1, -1):
        cur = ev(M0 + RS_T + 4 * s)
        cons.ext(O.SEL_ALL, esub(cur, eadd(emul(prev, fa), eb(pv(M0 + RS_V + s)))))
        prev = cur
    cons.ext(O.SEL_TRANSITION, egate(pv(RP_NOTFIRST, True), esub(ev(M0 + RS_ACCIN, True), ev(M0 + RS_T))))
    cons.ext(O.SEL_ALL, egate(padd(pv(RP_ACT), pneg(pv(RP_NOTFIRST))), ev(M0 + RS_ACCIN)))
    return O.air_program(RS_PRE + RS_MAIN, sh.NP * sh.NPUB, cons.c)


def rowsum_table():
    M0 = RS_PRE
    v, t0, fa = M0 + RS_V, M0 + RS_T, M0 + RS_FA
    return O.interaction_table([
        (SEND, RP_ACT, BUS_IN0, [RP_TAG, v, v + 1, v + 2, v + 3]), (SEND, RP_ACT, BUS_IN1, [RP_TAG, v + 4, v + 5, v + 6, v + 7]),
        (SEND, RP_LAST0, BUS_AT, [RP_QN, t0, t0 + 1, t0 + 2, t0 + 3]), (SEND, RP_LAST1, BUS_AQ, [RP_QN, t0, t0 + 1, t0 + 2, t0 + 3]),
        (RECV, RP_FIRST, BUS_KFA, [RP_PID, fa, fa + 1, fa + 2, fa + 3])])


def rowsum_rows(sh):
    """(q, block) in trace order: a query's trace blocks from the last to the first, then its quotient block"""
    return [(p, q, b) for p in range(sh.NP) for q in range(sh.Q) for b in list(range(sh.WB - 1, -1, -1)) + [sh.WB]]


def rowsum_pre(sh, log_rows):
    t = np.zeros((1 << log_rows, RS_PRE), dtype=np.uint32)
    per = sh.Q * (sh.WB + 1)
    for r, (p, q, b) in enumerate(rowsum_rows(sh)):
        t[r, RP_TAG], t[r, RP_ACT], t[r, RP_QN], t[r, RP_PID] = sh.row_tag(p, q, b), 1, p * sh.Q + q, p
        t[r, RP_NOTFIRST] = 0 if b in (sh.WB - 1, sh.WB) else 1
        t[r, RP_LAST0], t[r, RP_LAST1] = (1 if b == 0 else 0), (1 if b == sh.WB else 0)
        t[r, RP_FIRST], t[r, RP_NFC] = (1, 0) if r % per == 0 else (0, 1)
    return t


def _horner8(acc, vals, fa):
    steps = [None] * 8
    for s in range(7, -1, -1):
        acc = ext_mul(acc, fa)
        acc = [(acc[0] + vals[s]) % P] + acc[1:]
        steps[s] = acc
    return steps


def rowsum_main(sh, ws, fas, log_rows):
    """-> (trace, at[(p, q)], aq[(p, q)]); padding rows keep the last proof's fa (no constraint reads it there)"""
    t = np.zeros((1 << log_rows, RS_MAIN), dtype=np.uint64)
    t[:, RS_FA:RS_FA + 4] = fas[-1]
    at, aq, acc = {}, {}, [0, 0, 0, 0]
    for r, (p, q, b) in enumerate(rowsum_rows(sh)):
        op, fa = ws[p]["openings"][q], fas[p]
        t[r, RS_FA:RS_FA + 4] = fa
        vals = op["qrow"] if b == sh.WB else op["trow"][8 * b:8 * b + 8]
        if b in (sh.WB - 1, sh.WB):
            acc = [0, 0, 0, 0]
        t[r, RS_V:RS_V + 8], t[r, RS_ACCIN:RS_ACCIN + 4] = vals, acc
        steps = _horner8(acc, vals, fa)
        for s in range(8):
            t[r, RS_T + 4 * s:RS_T + 4 * s + 4] = steps[s]
        acc = steps[0]
        if b == 0:
            at[(p, q)] = acc
        if b == sh.WB:
            aq[(p, q)] = acc
    return t.astype(np.uint32), at, aq


# ---------------------------------------------------------------------------------------------------------------- QUERY
Q_PRE = 8
QP_QN, QP_ACT, QP_ACT2, QP_FIRST, QP_PID, QP_NFC, QP_PT = 0, 1, 2, 3, 4, 5, 6


def query_cols():
    m = Cols(Q_PRE)
    m("IDX", 1), m("XQ", 1)
    for name in ("RO", "AT", "AQ", "I1", "I2", "P1", "P2", "P2O", "P3", "P3O", "ZETA", "ZNX", "YL", "YN", "YQ", "OFFN", "OFFQ"):
        m(name)
    return m


QUERY_CONSTS = ("ZETA", "ZNX", "YL", "YN", "YQ", "OFFN", "OFFQ")
Q_MAIN = rup4(query_cols().n - Q_PRE)


def query_program(sh):
    m = query_cols()
    cons = Cons()
    for name in QUERY_CONSTS:
        cons.ext(O.SEL_TRANSITION, egate(pv(QP_NFC, True), esub(ev(m[name], True), ev(m[name]))))
    x = eb(pscale(pv(m["XQ"]), GEN))
    act = pv(QP_ACT)
    cons.ext(O.SEL_ALL, egate(act, esub(emul(esub(x, ev(m["ZETA"])), ev(m["I1"])), ec(1))))
    cons.ext(O.SEL_ALL, egate(act, esub(emul(esub(x, ev(m["ZNX"])), ev(m["I2"])), ec(1))))
    cons.ext(O.SEL_ALL, esub(ev(m["P1"]), emul(esub(ev(m["AT"]), ev(m["YL"])), ev(m["I1"]))))
    cons.ext(O.SEL_ALL, esub(ev(m["P2"]), emul(esub(ev(m["AT"]), ev(m["YN"])), ev(m["I2"]))))
    cons.ext(O.SEL_ALL, esub(ev(m["P2O"]), emul(ev(m["OFFN"]), ev(m["P2"]))))
    cons.ext(O.SEL_ALL, esub(ev(m["P3"]), emul(esub(ev(m["AQ"]), ev(m["YQ"])), ev(m["I1"]))))
    cons.ext(O.SEL_ALL, esub(ev(m["P3O"]), emul(ev(m["OFFQ"]), ev(m["P3"]))))
    cons.ext(O.SEL_ALL, esub(ev(m["RO"]), eadd(ev(m["P1"]), ev(m["P2O"]), ev(m["P3O"]))))
    return O.air_program(Q_PRE + Q_MAIN, sh.NP * sh.NPUB, cons.c)


def _e4(c):
    return [c, c + 1, c + 2, c + 3]


def query_table():
    m = query_cols()
    return O.interaction_table([
        (RECV, QP_ACT, F.BUS_I, [QP_QN, m["IDX"]]),
        (RECV, QP_ACT, F.BUS_Q, [QP_PT, m["IDX"], m["XQ"]] + _e4(m["RO"])),
        (RECV, QP_ACT, BUS_AT, [QP_QN] + _e4(m["AT"])), (RECV, QP_ACT, BUS_AQ, [QP_QN] + _e4(m["AQ"])),
        (SEND, QP_ACT2, BUS_QI, [QP_QN, m["IDX"]]),
        ] + [(RECV, QP_FIRST, BUS_K0 + i, [QP_PID] + _e4(m[name])) for i, name in enumerate(QUERY_CONSTS)])


def query_pre(sh, log_rows):
    t = np.zeros((1 << log_rows, Q_PRE), dtype=np.uint32)
    for p in range(sh.NP):
        for q in range(sh.Q):
            t[p * sh.Q + q] = [p * sh.Q + q, 1, 2, 1 if q == 0 else 0, p, 0 if q == 0 else 1, p * sh.TREES, 0]
    return t


def e_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def e_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def query_main(sh, ws, scs, at, aq, log_rows):
    m = query_cols()
    t = np.zeros((1 << log_rows, Q_MAIN), dtype=np.uint64)
    for name in QUERY_CONSTS:
        t[:, m[name] - Q_PRE:m[name] - Q_PRE + 4] = scs[-1][name]
    ros = []
    wM = two_adic_generator(sh.H)
    for pq in range(sh.NP * sh.Q):
        p, q = divmod(pq, sh.Q)
        w, sc, key = ws[p], scs[p], (p, q)
        for name in QUERY_CONSTS:
            t[pq, m[name] - Q_PRE:m[name] - Q_PRE + 4] = sc[name]
        index = w["openings"][q]["index"]
        xq = pow(wM, pyref.bitrev(index, sh.H), P)
        x = [GEN * xq % P, 0, 0, 0]
        i1, i2 = pyref.ext_inv(e_sub(x, sc["ZETA"])), pyref.ext_inv(e_sub(x, sc["ZNX"]))
        p1 = ext_mul(e_sub(at[key], sc["YL"]), i1)
        p2 = ext_mul(e_sub(at[key], sc["YN"]), i2)
        p2o = ext_mul(sc["OFFN"], p2)
        p3 = ext_mul(e_sub(aq[key], sc["YQ"]), i1)
        p3o = ext_mul(sc["OFFQ"], p3)
        ro = e_add(e_add(p1, p2o), p3o)
        ros.append(ro)
        r = t[pq]
        r[m["IDX"] - Q_PRE], r[m["XQ"] - Q_PRE] = index, xq
        for name, val in (("RO", ro), ("AT", at[key]), ("AQ", aq[key]), ("I1", i1), ("I2", i2), ("P1", p1), ("P2", p2), ("P2O", p2o), ("P3", p3), ("P3O", p3o)):
            r[m[name] - Q_PRE:m[name] - Q_PRE + 4] = val
    return t.astype(np.uint32), ros                                            # (padding rows: everything but the constants zero -- every product has a zero factor)


# ---------------------------------------------------------------------------------------------------------------- OPENED
OP_PRE = 12
OP_ACT, OP_FIRST, OP_LASTG, OP_NOTFIRST, OP_K1, OP_K2, OP_K3, OP_TL0, OP_TL1, OP_TN0, OP_TN1, OP_PID = range(12)


OP_PRE_AIR, OP_KEY0, OP_MUL0 = 28, 12, 20                                       # air mode: (key, multiplicity) of the row's eight opened values on the EVAL chip's bus


def op_pre(sh):
    return OP_PRE_AIR if sh.air else OP_PRE


def opened_cols(base=OP_PRE):
    m = Cols(base)
    for name in ("A", "B", "C", "D", "AN", "BN", "CN", "DN", "FA", "FA4", "ALPHA", "SELT", "SELF", "PW", "PWN", "H2", "H1", "IL", "G2", "G1", "INX",
                 "YLIN", "YLO", "YNIN", "YNO", "A2", "AB", "ACCIN", "U1", "U2", "ACCO"):
        m(name)
    return m


OPENED_CONSTS = ("FA", "FA4", "ALPHA", "SELT", "SELF")
OP_MAIN = opened_cols().n - OP_PRE


def opened_program(sh):
    m = opened_cols(op_pre(sh))
    cons = Cons()
    e = lambda name, nxt=False: ev(m[name], nxt)
    first, nf = pv(OP_FIRST), pv(OP_NOTFIRST, True)
    for name in OPENED_CONSTS:
        cons.ext(O.SEL_TRANSITION, egate(nf, esub(e(name, True), e(name))))
    cons.ext(O.SEL_ALL, egate(first, esub(e("PW"), ec(1))))
    cons.ext(O.SEL_ALL, esub(e("PWN"), emul(e("PW"), e("FA4"))))
    cons.ext(O.SEL_TRANSITION, egate(nf, esub(e("PW", True), e("PWN"))))
    fa = e("FA")
    for h2, h1, il, (a, b, c, d) in (("H2", "H1", "IL", ("A", "B", "C", "D")), ("G2", "G1", "INX", ("AN", "BN", "CN", "DN"))):
        cons.ext(O.SEL_ALL, esub(e(h2), eadd(e(c), emul(fa, e(d)))))
        cons.ext(O.SEL_ALL, esub(e(h1), eadd(e(b), emul(fa, e(h2)))))
        cons.ext(O.SEL_ALL, esub(e(il), eadd(e(a), emul(fa, e(h1)))))
    for yin, yo, il in (("YLIN", "YLO", "IL"), ("YNIN", "YNO", "INX")):
        cons.ext(O.SEL_ALL, egate(first, e(yin)))
        cons.ext(O.SEL_ALL, esub(e(yo), eadd(e(yin), emul(e("PW"), e(il)))))
        cons.ext(O.SEL_TRANSITION, egate(nf, esub(e(yin, True), e(yo))))
    if sh.air:                                                                  # the AIR's fold is the EVAL chip's: the columns behind YNO stay zero
        return O.air_program(op_pre(sh) + OP_MAIN, sh.NP * sh.NPUB, cons.c)
    # the synthetic AIR on the opened values (docs/PROTOCOL.md section 3): C1 = c - a^2 b - (g + 1), C2 = sel_transition (d' - a b - c - (2 g + 3)),
    # C3 = sel_first (d - (5 g + 7)), folded acc = acc alpha + C in this order, group after group
    cons.ext(O.SEL_ALL, esub(e("A2"), emul(e("A"), e("A"))))
    cons.ext(O.SEL_ALL, esub(e("AB"), emul(e("A"), e("B"))))
    al = e("ALPHA")
    cons.ext(O.SEL_ALL, egate(first, e("ACCIN")))
    cons.ext(O.SEL_ALL, esub(e("U1"), eadd(emul(e("ACCIN"), al), esub(esub(e("C"), emul(e("A2"), e("B"))), eb(pv(OP_K1))))))
    cons.ext(O.SEL_ALL, esub(e("U2"), eadd(emul(e("U1"), al), emul(e("SELT"), esub(esub(esub(e("DN"), e("AB")), e("C")), eb(pv(OP_K2)))))))
    cons.ext(O.SEL_ALL, esub(e("ACCO"), eadd(emul(e("U2"), al), emul(e("SELF"), esub(e("D"), eb(pv(OP_K3)))))))
    cons.ext(O.SEL_TRANSITION, egate(nf, esub(e("ACCIN", True), e("ACCO"))))
    return O.air_program(OP_PRE + OP_MAIN, sh.NP * sh.NPUB, cons.c)


def opened_table(sh):
    m = opened_cols(op_pre(sh))
    it = []
    for tag, lo, hi in ((OP_TL0, "A", "B"), (OP_TL1, "C", "D"), (OP_TN0, "AN", "BN"), (OP_TN1, "CN", "DN")):
        it += [(RECV, OP_ACT, BUS_IN0, [tag] + _e4(m[lo])), (RECV, OP_ACT, BUS_IN1, [tag] + _e4(m[hi]))]
    it += [(SEND, OP_LASTG, BUS_OY, [OP_PID] + _e4(m["YLO"])), (SEND, OP_LASTG, BUS_OY + 1, [OP_PID] + _e4(m["YNO"]))]
    if not sh.air:
        it += [(SEND, OP_LASTG, BUS_OA, [OP_PID] + _e4(m["ACCO"]))]
    it += [(RECV, OP_FIRST, BUS_KO0 + i, [OP_PID] + _e4(m[name])) for i, name in enumerate(OPENED_CONSTS)]
    if sh.air:                                                                  # A .. D at zeta, AN .. DN at zeta g
        it += [(SEND, OP_MUL0 + i, BUS_VAL, [OP_KEY0 + i] + _e4(m[name])) for i, name in enumerate(("A", "B", "C", "D", "AN", "BN", "CN", "DN"))]
    return O.interaction_table(it)


def opened_pre(sh, log_rows):
    t = np.zeros((1 << log_rows, op_pre(sh)), dtype=np.uint32)
    for p in range(sh.NP):
        for g in range(sh.G):
            r = t[p * sh.G + g]
            if sh.air:
                for i in range(4):
                    kl, kn = sh.key_local(4 * g + i), sh.key_next(4 * g + i)
                    r[OP_KEY0 + i], r[OP_MUL0 + i] = p * sh.KSPAN + kl, sh.mult[kl]
                    r[OP_KEY0 + 4 + i], r[OP_MUL0 + 4 + i] = p * sh.KSPAN + kn, sh.mult[kn]
            r[OP_ACT], r[OP_NOTFIRST], r[OP_PID] = 1, 1 if g else 0, p
            r[OP_K1], r[OP_K2], r[OP_K3] = g + 1, 2 * g + 3, 5 * g + 7
            r[OP_TL0], r[OP_TL1] = sh.ttag(p, sh.TO0 + 2 * g), sh.ttag(p, sh.TO0 + 2 * g + 1)
            r[OP_TN0], r[OP_TN1] = sh.ttag(p, sh.TO0 + sh.W // 2 + 2 * g), sh.ttag(p, sh.TO0 + sh.W // 2 + 2 * g + 1)
        t[p * sh.G, OP_FIRST], t[p * sh.G + sh.G - 1, OP_LASTG] = 1, 1
    return t


def opened_main(sh, ws, scs, log_rows):
    """-> (trace, per proof (y_loc, y_nxt, acc)); padding rows carry the last proof's constants"""
    m = opened_cols()
    t = np.zeros((1 << log_rows, OP_MAIN), dtype=np.uint64)
    put = lambda r, name, val: r.__setitem__(slice(m[name] - OP_PRE, m[name] - OP_PRE + 4), val)
    results = []
    for row in range(1 << log_rows):
        p, g = divmod(row, sh.G)
        active = p < sh.NP
        w, sc = (ws[p], scs[p]) if active else (None, scs[-1])
        fa, fa4, al, selt, self_ = sc["FA"], sc["FA4"], sc["ALPHA"], sc["SELT"], sc["SELF"]
        if g == 0:
            pw, yl, yn, acc = [1, 0, 0, 0], [0] * 4, [0] * 4, [0] * 4
        r = t[row]
        for name, val in zip(OPENED_CONSTS, (fa, fa4, al, selt, self_)):
            put(r, name, val)
        if active:
            a, b, c, d = w["loc"][4 * g:4 * g + 4]
            an, bn, cn, dn = w["nxt"][4 * g:4 * g + 4]
        else:
            a = b = c = d = an = bn = cn = dn = [0] * 4
            pw, yl, yn, acc = [0] * 4, [0] * 4, [0] * 4, [0] * 4
        for name, val in zip(("A", "B", "C", "D", "AN", "BN", "CN", "DN"), (a, b, c, d, an, bn, cn, dn)):
            put(r, name, val)
        put(r, "PW", pw)
        pwn = ext_mul(pw, fa4)
        put(r, "PWN", pwn)
        outs = []
        for h2n, h1n, iln, (x0, x1, x2, x3) in (("H2", "H1", "IL", (a, b, c, d)), ("G2", "G1", "INX", (an, bn, cn, dn))):
            h2 = e_add(x2, ext_mul(fa, x3))
            h1 = e_add(x1, ext_mul(fa, h2))
            il = e_add(x0, ext_mul(fa, h1))
            put(r, h2n, h2), put(r, h1n, h1), put(r, iln, il)
            outs.append(il)
        put(r, "YLIN", yl), put(r, "YNIN", yn)
        yl, yn = e_add(yl, ext_mul(pw, outs[0])), e_add(yn, ext_mul(pw, outs[1]))
        put(r, "YLO", yl), put(r, "YNO", yn)
        if not sh.air:
            a2, ab = ext_mul(a, a), ext_mul(a, b)
            put(r, "A2", a2), put(r, "AB", ab), put(r, "ACCIN", acc)
            k1, k2, k3 = (g + 1, 2 * g + 3, 5 * g + 7) if active else (0, 0, 0)
            u1 = e_add(ext_mul(acc, al), e_sub(e_sub(c, ext_mul(a2, b)), [k1, 0, 0, 0]))
            u2 = e_add(ext_mul(u1, al), ext_mul(selt, e_sub(e_sub(e_sub(dn, ab), c), [k2, 0, 0, 0])))
            acc = e_add(ext_mul(u2, al), ext_mul(self_, e_sub(d, [k3, 0, 0, 0])))
            put(r, "U1", u1), put(r, "U2", u2), put(r, "ACCO", acc)
        pw = pwn
        if active and g == sh.G - 1:
            results.append((yl, yn, acc))
    return t.astype(np.uint32), results


# ---------------------------------------------------------------------------------------------------------------- SCALARS
SC_PRE = 12
SP_FIRST, SP_KA, SP_KZ, SP_KF, SP_TQZ, SP_PID = 0, 1, 2, 3, 4, 8                 # FIRST: the row is a proof's row (rows behind the proofs repeat row 0: every constraint holds there too)


SC_PRE_AIR, SP_KSEL, SP_KONE, SP_MSEL, SP_MONE, SP_Z = 20, 9, 12, 13, 16, 17      # air mode: the selectors and the constant one go to the EVAL chip


def sc_pre(sh):
    return SC_PRE_AIR if sh.air else SC_PRE


def scalars_cols(sh):
    m = Cols(sc_pre(sh))
    for name in ("ALPHA", "ZETA", "FA"):
        m(name)
    for i in range(1, sh.n + 1):
        m("ZP%d" % i)
    for name in ("INVF", "SELF", "SELT", "ZNX"):
        m(name)
    mb = sh.W.bit_length() - 1
    for i in range(1, mb + 1):
        m("FP%d" % i)
    bits = [i for i in range(mb + 1) if (sh.W >> i) & 1]
    for k in range(1, len(bits)):
        m("PR%d" % k)
    m("OFFN"), m("OFFQ")
    for j in range(8):
        m("QZ%d" % j)
    for j in range(7):
        m("HQ%d" % j)
    for name in ("QK0", "QK1", "QUO", "YL", "YN", "ACC"):
        m(name)
    if sh.air:                                                                  # the last-row selector Z_H(zeta) / (zeta - w^-1): a program may use it
        m("INVT"), m("SELL")
    return m


def zps_consts(sh):
    """zps_k(zeta) = a_k zeta^N + b_k for the two quotient chunks (tests/pyverify.py, the recombination)"""
    N = 1 << sh.n
    wq = two_adic_generator(sh.n + 1)
    sN = [pow(GEN * pow(wq, k, P) % P, N, P) for k in range(2)]
    out = []
    for k in range(2):
        j = 1 - k
        sj_inv = pow(sN[j], -1, P)
        den_inv = pow((sN[k] * sj_inv - 1) % P, -1, P)
        out.append((sj_inv * den_inv % P, (P - den_inv) % P))
    return out


def scalars_program(sh):
    m = scalars_cols(sh)
    cons = Cons()
    e = lambda name: ev(m[name])
    prev = e("ZETA")
    for i in range(1, sh.n + 1):
        cons.ext(O.SEL_ALL, esub(e("ZP%d" % i), emul(prev, prev)))
        prev = e("ZP%d" % i)
    znn = prev
    wni = pow(two_adic_generator(sh.n), -1, P)
    cons.ext(O.SEL_ALL, esub(emul(esub(e("ZETA"), ec(1)), e("INVF")), ec(1)))
    cons.ext(O.SEL_ALL, esub(e("SELF"), emul(esub(znn, ec(1)), e("INVF"))))
    cons.ext(O.SEL_ALL, esub(e("SELT"), esub(e("ZETA"), ec(wni))))
    cons.ext(O.SEL_ALL, esub(e("ZNX"), escale(e("ZETA"), two_adic_generator(sh.n))))
    mb = sh.W.bit_length() - 1
    prev = e("FA")
    fp = [prev]
    for i in range(1, mb + 1):
        cons.ext(O.SEL_ALL, esub(e("FP%d" % i), emul(prev, prev)))
        prev = e("FP%d" % i)
        fp.append(prev)
    bits = [i for i in range(mb + 1) if (sh.W >> i) & 1]
    acc = fp[bits[0]]
    for k in range(1, len(bits)):
        cons.ext(O.SEL_ALL, esub(e("PR%d" % k), emul(acc, fp[bits[k]])))
        acc = e("PR%d" % k)
    cons.ext(O.SEL_ALL, esub(e("OFFN"), acc))
    cons.ext(O.SEL_ALL, esub(e("OFFQ"), emul(e("OFFN"), e("OFFN"))))
    prev = e("QZ7")
    for j in range(6, -1, -1):
        cons.ext(O.SEL_ALL, esub(e("HQ%d" % j), eadd(e("QZ%d" % j), emul(e("FA"), prev))))
        prev = e("HQ%d" % j)
    for k in range(2):
        q = ec(0)
        for t in range(4):
            basis = [0, 0, 0, 0]
            basis[t] = 1
            q = eadd(q, emul(ec(basis), e("QZ%d" % (4 * k + t))))
        cons.ext(O.SEL_ALL, esub(e("QK%d" % k), q))
    (a0, b0), (a1, b1) = zps_consts(sh)
    z0, z1 = eadd(escale(znn, a0), ec(b0)), eadd(escale(znn, a1), ec(b1))
    cons.ext(O.SEL_ALL, esub(e("QUO"), eadd(emul(z0, e("QK0")), emul(z1, e("QK1")))))
    cons.ext(O.SEL_ALL, esub(e("ACC"), emul(e("QUO"), esub(znn, ec(1)))))
    if sh.air:
        cons.ext(O.SEL_ALL, esub(emul(e("SELT"), e("INVT")), ec(1)))
        cons.ext(O.SEL_ALL, esub(e("SELL"), emul(esub(znn, ec(1)), e("INVT"))))
    return O.air_program(sc_pre(sh) + rup4(m.n - sc_pre(sh)), sh.NP * sh.NPUB, cons.c)


def scalars_table(sh):
    m = scalars_cols(sh)
    it = [(RECV, SP_FIRST, BUS_SC, [SP_KA] + _e4(m["ALPHA"])), (RECV, SP_FIRST, BUS_SC, [SP_KZ] + _e4(m["ZETA"])), (RECV, SP_FIRST, BUS_SC, [SP_KF] + _e4(m["FA"]))]
    for i in range(4):
        it += [(RECV, SP_FIRST, BUS_IN0, [SP_TQZ + i] + _e4(m["QZ%d" % (2 * i)])), (RECV, SP_FIRST, BUS_IN1, [SP_TQZ + i] + _e4(m["QZ%d" % (2 * i + 1)]))]
    it += [(RECV, SP_FIRST, BUS_OY, [SP_PID] + _e4(m["YL"])), (RECV, SP_FIRST, BUS_OY + 1, [SP_PID] + _e4(m["YN"])), (RECV, SP_FIRST, BUS_OA, [SP_PID] + _e4(m["ACC"]))]
    it += [(SEND, SP_FIRST, BUS_K0 + i, [SP_PID] + _e4(m[name])) for i, name in enumerate(("ZETA", "ZNX", "YL", "YN", "HQ0", "OFFN", "OFFQ"))]
    it += [(SEND, SP_FIRST, BUS_KFA, [SP_PID] + _e4(m["FA"]))]
    it += [(SEND, SP_FIRST, BUS_KO0 + i, [SP_PID] + _e4(m[name])) for i, name in enumerate(("FA", "FP2", "ALPHA", "SELT", "SELF"))]
    if sh.air:
        it += [(SEND, SP_MSEL + i, BUS_VAL, [SP_KSEL + i] + _e4(m[name])) for i, name in enumerate(("SELF", "SELL", "SELT"))]
        it += [(SEND, SP_MONE, BUS_VAL, [SP_KONE, SP_FIRST, SP_Z, SP_Z, SP_Z]), (SEND, SP_FIRST, BUS_EA, [SP_PID] + _e4(m["ALPHA"]))]
    return O.interaction_table(it)


def scalars_pre(sh, log_rows):
    t = np.zeros((1 << log_rows, sc_pre(sh)), dtype=np.uint32)
    for p in range(sh.NP):
        t[p, SP_FIRST], t[p, SP_PID] = 1, p
        if sh.air:
            for i in range(3):
                t[p, SP_KSEL + i], t[p, SP_MSEL + i] = p * sh.KSPAN + sh.key_sel(i), sh.mult[sh.key_sel(i)]
            t[p, SP_KONE], t[p, SP_MONE] = p * sh.KSPAN, sh.mult[0]
        t[p, SP_KA], t[p, SP_KZ], t[p, SP_KF] = 3 * p, 3 * p + 1, 3 * p + 2
        for i in range(4):
            t[p, SP_TQZ + i] = sh.ttag(p, sh.TO0 + sh.W + i)
    return t


def scalars_values(sh, w, chal):
    """the verifier's scalars up to what the OPENED chip needs (everything but YL, YN, ACC)"""
    m = {}
    m["ALPHA"], m["ZETA"], m["FA"] = chal[sh.TA], chal[sh.TQ], chal[sh.TF]
    prev = m["ZETA"]
    for i in range(1, sh.n + 1):
        prev = m["ZP%d" % i] = ext_mul(prev, prev)
    znn = prev
    wn = two_adic_generator(sh.n)
    wni = pow(wn, -1, P)
    m["INVF"] = pyref.ext_inv(e_sub(m["ZETA"], [1, 0, 0, 0]))
    m["SELF"] = ext_mul(e_sub(znn, [1, 0, 0, 0]), m["INVF"])
    m["SELT"] = e_sub(m["ZETA"], [wni, 0, 0, 0])
    m["ZNX"] = [x * wn % P for x in m["ZETA"]]
    mb = sh.W.bit_length() - 1
    prev = m["FA"]
    fp = [prev]
    for i in range(1, mb + 1):
        prev = m["FP%d" % i] = ext_mul(prev, prev)
        fp.append(prev)
    bits = [i for i in range(mb + 1) if (sh.W >> i) & 1]
    acc = fp[bits[0]]
    for k in range(1, len(bits)):
        acc = m["PR%d" % k] = ext_mul(acc, fp[bits[k]])
    m["OFFN"] = acc
    m["OFFQ"] = ext_mul(acc, acc)
    for j in range(8):
        m["QZ%d" % j] = list(w["qz"][j])
    prev = m["QZ7"]
    for j in range(6, -1, -1):
        prev = m["HQ%d" % j] = e_add(m["QZ%d" % j], ext_mul(m["FA"], prev))
    for k in range(2):
        q = [0] * 4
        for t in range(4):
            basis = [0, 0, 0, 0]
            basis[t] = 1
            q = e_add(q, ext_mul(basis, m["QZ%d" % (4 * k + t)]))
        m["QK%d" % k] = q
    (a0, b0), (a1, b1) = zps_consts(sh)
    z0 = e_add([x * a0 % P for x in znn], [b0, 0, 0, 0])
    z1 = e_add([x * a1 % P for x in znn], [b1, 0, 0, 0])
    m["QUO"] = e_add(ext_mul(z0, m["QK0"]), ext_mul(z1, m["QK1"]))
    m["FA4"], m["YQ"] = m["FP2"], m["HQ0"]
    m["ZNN"] = znn
    if sh.air:
        m["INVT"] = pyref.ext_inv(m["SELT"])
        m["SELL"] = ext_mul(e_sub(znn, [1, 0, 0, 0]), m["INVT"])
    return m


def scalars_main(sh, scs, log_rows):
    m = scalars_cols(sh)
    base = sc_pre(sh)
    width = rup4(m.n - base)
    t = np.zeros((1 << log_rows, width), dtype=np.uint64)
    for r in range(1 << log_rows):
        sc = scs[r] if r < sh.NP else scs[0]
        for name, col in m.at.items():
            t[r, col - base:col - base + 4] = sc[name]
    return t.astype(np.uint32)


# ---------------------------------------------------------------------------------------------------------------- EVAL (air mode)
# One row per TERM of the inner program: coeff x F1 x F2 x F3, the three factor values received over BUS_VAL by their preprocessed keys (an opened
# value at zeta / zeta g from OPENED, a public value from TS, a selector or the constant one from SCALARS); the fold of the constraints with alpha
# runs down the rows: where a constraint's first term stands, ACC is multiplied by alpha first.
EV_PRE = 12
EP_COEF, EP_K0, EP_FIRSTC, EP_ACT, EP_LAST, EP_PID, EP_NFC, EP_PFIRST = 0, 1, 4, 5, 6, 7, 8, 9
EV_F0, EV_M, EV_TV, EV_ACCIN, EV_ACCO, EV_ALPHA, EV_MAIN = 0, 12, 16, 20, 24, 28, 32


def eval_program(sh):
    M0 = EV_PRE
    cons = Cons()
    f0, f1, f2, mm, tv, ai, ao, al = (ev(M0 + c) for c in (EV_F0, EV_F0 + 4, EV_F0 + 8, EV_M, EV_TV, EV_ACCIN, EV_ACCO, EV_ALPHA))
    cons.ext(O.SEL_ALL, esub(mm, emul(f0, f1)))
    cons.ext(O.SEL_ALL, esub(tv, egate(pv(EP_COEF), emul(mm, f2))))
    cons.ext(O.SEL_ALL, esub(ao, eadd(ai, egate(pv(EP_FIRSTC), esub(emul(ai, al), ai)), tv)))
    cons.ext(O.SEL_TRANSITION, egate(pv(EP_NFC, True), esub(ev(M0 + EV_ACCIN, True), ao)))
    cons.ext(O.SEL_ALL, egate(pv(EP_PFIRST), ai))
    cons.ext(O.SEL_TRANSITION, egate(pv(EP_NFC, True), esub(ev(M0 + EV_ALPHA, True), al)))
    return O.air_program(EV_PRE + EV_MAIN, sh.NP * sh.NPUB, cons.c)


def eval_table():
    M0 = EV_PRE
    it = [(RECV, EP_ACT, BUS_VAL, [EP_K0 + j] + _e4(M0 + EV_F0 + 4 * j)) for j in range(3)]
    it += [(RECV, EP_PFIRST, BUS_EA, [EP_PID] + _e4(M0 + EV_ALPHA)), (SEND, EP_LAST, BUS_OA, [EP_PID] + _e4(M0 + EV_ACCO))]
    return O.interaction_table(it)


def eval_pre(sh, log_rows):
    t = np.zeros((1 << log_rows, EV_PRE), dtype=np.uint32)
    nt = len(sh.terms)
    for p in range(sh.NP):
        for i, (coeff, keys, first) in enumerate(sh.terms):
            r = t[p * nt + i]
            r[EP_COEF], r[EP_FIRSTC], r[EP_ACT], r[EP_LAST], r[EP_PID] = coeff, first, 1, int(i + 1 == nt), p
            r[EP_K0:EP_K0 + 3] = [p * sh.KSPAN + k for k in keys]
            r[EP_NFC], r[EP_PFIRST] = int(i > 0), int(i == 0)
    return t


def eval_main(sh, ws, scs, public_values, log_rows):
    """-> (trace, per proof the program folded with alpha at zeta)"""
    t = np.zeros((1 << log_rows, EV_MAIN), dtype=np.uint64)
    nt, W = len(sh.terms), sh.W
    accs = []
    for p, (w, sc, pubs) in enumerate(zip(ws, scs, public_values)):
        def value(key):
            if key == 0:
                return [1, 0, 0, 0]
            if key <= W:
                return list(w["loc"][key - 1])
            if key <= 2 * W:
                return list(w["nxt"][key - 1 - W])
            if key <= 2 * W + sh.NPUB:
                return [int(pubs[key - 1 - 2 * W]) % P, 0, 0, 0]
            return [sc["SELF"], sc["SELL"], sc["SELT"]][key - 1 - 2 * W - sh.NPUB]
        run = [0] * 4
        for i, (coeff, keys, first) in enumerate(sh.terms):
            r = t[p * nt + i]
            f0, f1, f2 = (value(k) for k in keys)
            mm = ext_mul(f0, f1)
            tv = [x * coeff % P for x in ext_mul(mm, f2)]
            r[EV_F0:EV_F0 + 4], r[EV_F0 + 4:EV_F0 + 8], r[EV_F0 + 8:EV_F0 + 12], r[EV_M:EV_M + 4], r[EV_TV:EV_TV + 4] = f0, f1, f2, mm, tv
            r[EV_ACCIN:EV_ACCIN + 4] = run
            run = e_add(ext_mul(run, sc["ALPHA"]) if first else run, tv)
            r[EV_ACCO:EV_ACCO + 4], r[EV_ALPHA:EV_ALPHA + 4] = run, sc["ALPHA"]
        accs.append(run)
    return t.astype(np.uint32), accs


# ---------------------------------------------------------------------------------------------------------------- FOLD (tests/fri_air.py, rec form)
def fold_table(layers):
    t = [(SEND, F.ACTIVE, F.BUS_E0, [F.LNX, F.K2, F.E0, F.E0 + 1, F.E0 + 2, F.E0 + 3]), (SEND, F.ACTIVE, F.BUS_E1, [F.LNX, F.K2, F.E1, F.E1 + 1, F.E1 + 2, F.E1 + 3]),
         (SEND, F.L_REC, F.BUS_Q, [F.PT, F.IDX, F.XS, F.OWN, F.OWN + 1, F.OWN + 2, F.OWN + 3]),
         (RECV, F.ACTIVE, BUS_BETA, [F.LNX, F.BETA, F.BETA + 1, F.BETA + 2, F.BETA + 3]),
         (SEND, F.L_REC + layers - 1, BUS_FIN, [F.PT, F.FOLD, F.FOLD + 1, F.FOLD + 2, F.FOLD + 3])]
    return O.interaction_table(t)


# ---------------------------------------------------------------------------------------------------------------- the witness and the machine
def witness(proof, log_n, width, public_values, n_queries, pow_bits, sh=None, program=None):
    """everything the machine's main columns hold for ONE inner proof, taken from it by the Python verifier (which must accept it)"""
    import pyverify
    view = {}
    pyverify.verify(proof, log_n, width, public_values, num_queries=n_queries, pow_bits=pow_bits, view=view, air=program)
    sh = sh or Shape(log_n, width, n_queries, pow_bits, len(public_values), program=program)
    w = {"trace_root": view["trace_root"], "quot_root": view["quot_root"], "layer_roots": view["roots"], "final": view["final"], "witness": view["witness"],
         "loc": view["loc"], "nxt": view["nxt"], "qz": view["qz"], "openings": view["openings"], "betas": view["betas"], "view": view}
    # the blocks the transcript absorbs, by sponge row
    seq0 = sh.head + list(view["trace_root"]) + [int(v) % P for v in public_values]
    blocks = {}
    for T in range(sh.f0 + (1 if sh.r0 else 0)):
        blocks[T] = seq0[8 * T:8 * T + 8]
    blocks[sh.TQ] = list(view["quot_root"])
    opened = [c for e in view["loc"] for c in e] + [c for e in view["nxt"] for c in e] + [c for e in view["qz"] for c in e]
    for i in range(width + 4):
        blocks[sh.TO0 + i] = opened[8 * i:8 * i + 8]
    for l in range(sh.R):
        blocks[sh.TL0 + l] = list(view["roots"][l])
    blocks[sh.TP] = list(view["final"]) + [view["witness"]]
    w["blocks"] = blocks
    # the FRI pairs of every (query, layer): (pair index, the two entries, the path)
    fri = []
    for q, (index, value, sibs) in enumerate(view["queries"]):
        idx, own, per = index, list(value), []
        for l in range(sh.R):
            bit, k = idx & 1, idx >> 1
            e0, e1 = (sibs[l], own) if bit else (own, sibs[l])
            per.append((k, list(e0) + list(e1), view["paths"][q][l]))
            own, idx = F.fold_pair(k, sh.H - (l + 1), view["betas"][l], e0, e1)[0], k
        fri.append(per)
    w["fri"] = fri
    return sh, w


CHIPS = ("P2R", "ROWSUM", "FOLD", "TS", "QUERY", "OPENED", "SAMPLES", "SCALARS", "EVAL")       # (EVAL: air mode only)


def chips(sh):
    return CHIPS if sh.air else CHIPS[:-1]


def heights(sh):
    n = sh.NP
    h = {"P2R": lg(n * sh.p2_rows), "ROWSUM": lg(n * sh.Q * (sh.WB + 1)), "FOLD": lg(n * sh.Q * sh.R), "TS": lg(n * sh.NTS), "QUERY": lg(n * sh.Q), "OPENED": lg(n * sh.G),
         "SAMPLES": lg(n * sh.NS), "SCALARS": lg(n)}
    if sh.air:
        h["EVAL"] = lg(n * len(sh.terms))
    return h


def order(sh):
    """tallest first; equal heights in the order of CHIPS"""
    h = heights(sh)
    return sorted(chips(sh), key=lambda c: (-h[c], CHIPS.index(c)))


def programs(sh):
    npub = sh.NP * sh.NPUB
    d = {"P2R": p2r_program(sh), "ROWSUM": rowsum_program(sh), "FOLD": F.program(sh.R, wired=True, transcript=True, rec=npub), "TS": ts_program(sh),
         "QUERY": query_program(sh), "OPENED": opened_program(sh), "SAMPLES": F.samples_program(sh.R, sh.Q, sh.PB, npub), "SCALARS": scalars_program(sh)}
    if sh.air:
        d["EVAL"] = eval_program(sh)
    return d


def tables(sh):
    M0 = F.S_PRE
    s_tab = O.interaction_table([(RECV, F.S_ROW, F.BUS_S0, [F.S_C] + [M0 + F.S_W + j for j in range(4)]), (RECV, F.S_ROW, F.BUS_S1, [F.S_C] + [M0 + F.S_W + j for j in range(4, 8)])]
                                + [(SEND, F.S_ACT + j, F.BUS_I, [F.S_KQ + j, M0 + F.S_IDX + j]) for j in range(8)])
    d = {"P2R": p2r_table(), "ROWSUM": rowsum_table(), "FOLD": fold_table(sh.R), "TS": ts_table(sh), "QUERY": query_table(), "OPENED": opened_table(sh),
         "SAMPLES": s_tab, "SCALARS": scalars_table(sh)}
    if sh.air:
        d["EVAL"] = eval_table()
    return d


def samples_stacked(sh, words_per_proof, log_rows):
    """the SAMPLES chip for several proofs: proof p's rows behind proof p - 1's; its sponge rows are numbered from its own tags, its queries from p Q"""
    pre = np.zeros((1 << log_rows, F.S_PRE), dtype=np.uint32)
    main = np.zeros((1 << log_rows, F.S_MAIN), dtype=np.uint32)
    drawn = []
    for p, words in enumerate(words_per_proof):
        a, b, d = F.samples_tables(sh.R, sh.Q, words, lg(sh.NS), base=sh.ttag(p, sh.TP))
        for j in range(8):
            a[:, F.S_KQ + j] += (p * sh.Q) * a[:, F.S_ACT + j]
        pre[p * sh.NS:(p + 1) * sh.NS], main[p * sh.NS:(p + 1) * sh.NS] = a[:sh.NS], b[:sh.NS]
        drawn.append(d)
    return pre, main, drawn


def preprocessed(sh):
    """the key material: every chip's preprocessed trace (None: the chip has none) -- a function of the shape"""
    h = heights(sh)
    spre, _, _ = samples_stacked(sh, [[[0] * 8] * sh.NS] * sh.NP, h["SAMPLES"])
    d = {"P2R": p2r_pre(sh, h["P2R"]), "ROWSUM": rowsum_pre(sh, h["ROWSUM"]), "FOLD": None, "TS": ts_pre(sh, h["TS"]), "QUERY": query_pre(sh, h["QUERY"]),
         "OPENED": opened_pre(sh, h["OPENED"]), "SAMPLES": spre, "SCALARS": scalars_pre(sh, h["SCALARS"])}
    if sh.air:
        d["EVAL"] = eval_pre(sh, h["EVAL"])
    return d


def fold_stacked(sh, ws, log_rows):
    t = np.zeros((1 << log_rows, F.width_of(sh.R, rec=True)), dtype=np.uint32)
    t[:, F.T] = 1
    per = sh.Q * sh.R
    for p, w in enumerate(ws):
        one, final = F.trace(w["view"], lg(per), wired=True, rec=True, pt=p * sh.TREES)
        assert list(final) == list(w["final"])
        t[p * per:(p + 1) * per] = one[:per]
    return t


def main_traces(sh, ws, public_values=None):
    h = heights(sh)
    p2, samples, chals = p2r_main(sh, ws, h["P2R"])
    scs = []
    for w, chal in zip(ws, chals):
        assert chal[sh.TA] == w["view"]["alpha"] and chal[sh.TQ] == w["view"]["zeta"] and chal[sh.TF] == w["view"]["fa"], "the sponge rows do not reproduce the verifier's challenges"
        assert [chal[sh.TL0 + l] for l in range(sh.R)] == w["betas"]
        scs.append(scalars_values(sh, w, chal))
    opened, results = opened_main(sh, ws, scs, h["OPENED"])
    evl = None
    if sh.air:                                                                  # the fold of the AIR at zeta is the EVAL chip's
        evl, accs = eval_main(sh, ws, scs, public_values, h["EVAL"])
        results = [(yl, yn, acc) for (yl, yn, _), acc in zip(results, accs)]
    for sc, (yl, yn, acc) in zip(scs, results):
        sc["YL"], sc["YN"], sc["ACC"] = yl, yn, acc
        assert acc == ext_mul(sc["QUO"], e_sub(sc["ZNN"], [1, 0, 0, 0])), "the AIR identity at zeta does not hold"
    rs, at, aq = rowsum_main(sh, ws, [sc["FA"] for sc in scs], h["ROWSUM"])
    qm, ros = query_main(sh, ws, scs, at, aq, h["QUERY"])
    assert ros == [list(v) for w in ws for _, v, _ in w["view"]["queries"]], "the reduced openings are not the verifier's"
    fold = fold_stacked(sh, ws, h["FOLD"])
    _, smain, drawn = samples_stacked(sh, samples, h["SAMPLES"])
    assert drawn == [[op["index"] for op in w["openings"]] for w in ws], "the query indices are not the ones the transcript draws"
    ts = ts_main(sh, ws, chals, p2, h["TS"])
    d = {"P2R": p2, "ROWSUM": rs, "FOLD": fold, "TS": ts, "QUERY": qm, "OPENED": opened, "SAMPLES": smain, "SCALARS": scalars_main(sh, scs, h["SCALARS"])}
    if sh.air:
        d["EVAL"] = evl
    return d


def machine(proofs, log_n, width, public_values, n_queries, pow_bits, program=None):
    """proofs: ONE inner proof (bytes) with its public values, or a LIST of proofs with a list of public-value lists (the join: one outer proof for all);
    program: the constraint program of version-7 inner proofs (air mode), None for version-1 proofs of the synthetic AIR
    -> (shape, main traces, preprocessed traces, programs, interaction tables, public values), chips tallest first"""
    if isinstance(proofs, (bytes, bytearray)):
        proofs, public_values = [proofs], [public_values]
    sh = Shape(log_n, width, n_queries, pow_bits, len(public_values[0]), len(proofs), program=program)
    ws = [witness(pr, log_n, width, pv, n_queries, pow_bits, sh, program)[1] for pr, pv in zip(proofs, public_values)]
    names = order(sh)
    mt, pre, prog, tab = main_traces(sh, ws, public_values), preprocessed(sh), programs(sh), tables(sh)
    return sh, [mt[c] for c in names], [pre[c] for c in names], [prog[c] for c in names], [tab[c] for c in names], [int(v) % P for pv in public_values for v in pv]


# ---------------------------------------------------------------------------------------------------------------- a small inner program for the tests
def counter_program(width=8):
    """a small program that uses everything a program can: the three selectors, public values, a next-row variable, a degree-three term"""
    v = O.air_var

    def pub(i):
        return O.air_var(i, public=True)
    cons = [(O.SEL_FIRST, [(1, [v(0)]), (P - 1, [pub(0)])]),                                      # first row: c0 = pub0
            (O.SEL_TRANSITION, [(1, [v(0, True)]), (P - 1, [v(0)]), (P - 1, [])]),              # c0' = c0 + 1
            (O.SEL_LAST, [(1, [v(0)]), (P - 1, [pub(1)])]),                                       # last row: c0 = pub1
            (O.SEL_ALL, [(1, [v(1)]), (P - 1, [v(0), v(0)])]),                                    # c1 = c0^2
            (O.SEL_ALL, [(1, [v(2)]), (P - 3, [v(0), v(1), v(0)]), (P - 5, [pub(2)])]),         # c2 = 3 c0^2 c1 + 5 pub2
            (O.SEL_TRANSITION, [(1, [v(3, True)]), (P - 1, [v(3), v(width - 1)])])]               # c3' = c3 c7
    return O.air_program(width, 3, cons)


def counter_trace(log_n, width, start, pub2, seed=3):
    rng = np.random.default_rng(seed)
    t = rng.integers(0, P, size=(1 << log_n, width), dtype=np.uint64)
    for r in range(1 << log_n):
        c0 = (start + r) % P
        t[r, 0], t[r, 1] = c0, c0 * c0 % P
        t[r, 2] = (3 * c0 * c0 % P * int(t[r, 1]) + 5 * pub2) % P
        if r:
            t[r, 3] = int(t[r - 1, 3]) * int(t[r - 1, width - 1]) % P
    return t.astype(np.uint32), [start % P, (start + (1 << log_n) - 1) % P, pub2]



# ---------------------------------------------------------------------------------------------------------------- checks in plain integers
def bus_events(mains, pres, tabs):
    """every active interaction of every row: -> {bus: [(sign, chip, interaction index, row, multiplicity, tuple, preprocessed?-per-position)]}"""
    out = {}
    for ci, (main, pre, tab) in enumerate(zip(mains, pres, tabs)):
        if tab is None:
            continue
        pw = 0 if pre is None else pre.shape[1]
        rows = (main if pre is None else np.concatenate([pre, main], axis=1)).astype(np.int64)
        t = [int(x) for x in tab]
        p = 3
        for k in range(t[1]):
            sign, mult, bus, nv = t[p:p + 4]
            cols = t[p + 4:p + 4 + nv]
            p += 4 + nv
            mus = np.ones(rows.shape[0], dtype=np.int64) if mult == 0xFFFFFFFF else rows[:, mult]
            for r in np.nonzero(mus)[0]:
                out.setdefault(bus, []).append((sign, ci, k, int(r), int(mus[r]), tuple(int(rows[r, c]) for c in cols), tuple(c < pw for c in cols)))
    return out


def bus_ambiguity(mains, pres, tabs, extra_identity=None, function_tables=()):
    """A bus is a MULTISET: two sends whose tuples agree on everything that NAMES their receiver are exchangeable without any cell looking wrong
    (the class of 8ca4614: the fold chain's tuple for a lower height lacked the query's index, so two queries could take each other's reduced openings;
    a flipped-cell test cannot see it).  The audit, over an honest trace:
      * the positions of a bus's tuple that NAME the receiver: those that are PREPROCESSED columns on the receiving side in every receive interaction of
        the bus (the key fixes them per row), or preprocessed on the sending side in every send interaction; plus `extra_identity[bus]` -- main columns
        that a constraint pins to the receiver's identity (declared with their reason where the machine is defined);
      * two active RECEIVES on different rows whose tuples agree on the naming positions but differ elsewhere are a finding: a sender cannot tell
        the two receivers apart, so what it hands one of them can be handed to the other.  (Receivers that agree on the WHOLE tuple are one
        receiver with a multiplicity; buses in `function_tables` are lookups into a relation -- the receiver's whole tuple is fixed by its own
        constraints, e.g. a Poseidon2 row's (input, output) -- where any row may serve any sender by design.)
    -> [(bus, naming positions, (chip, row, tuple), (chip, row, tuple))], empty when every receiver is named"""
    extra_identity = extra_identity or {}
    findings = []
    for bus, ev in sorted(bus_events(mains, pres, tabs).items()):
        if bus in function_tables:
            continue
        recv, send = [e for e in ev if e[0] == RECV], [e for e in ev if e[0] == SEND]
        if not recv:
            continue
        width = len(recv[0][5])
        if any(len(e[5]) != width for e in ev):                # (interactions of different lengths on one bus never meet: audit them by length)
            groups = {}
            for e in ev:
                groups.setdefault(len(e[5]), []).append(e)
        else:
            groups = {width: ev}
        for width, evs in sorted(groups.items()):
            recv, send = [e for e in evs if e[0] == RECV], [e for e in evs if e[0] == SEND]
            if not recv:
                continue
            named = [k for k in range(width) if all(e[6][k] for e in recv) or (send and all(e[6][k] for e in send))]
            named = sorted(set(named) | set(extra_identity.get((bus, width), extra_identity.get(bus, []))))
            seen = {}
            for e in recv:
                key = tuple(e[5][k] for k in named)
                if key in seen and seen[key][5] != e[5] and (seen[key][1], seen[key][3]) != (e[1], e[3]):
                    findings.append((bus, tuple(named), (seen[key][1], seen[key][3], seen[key][5]), (e[1], e[3], e[5])))
                    break                                      # one finding per bus and length is enough to fail
                seen.setdefault(key, e)
    return findings


def bus_balance(mains, pres, tabs):
    """every tuple sent on a bus is received with the same total multiplicity -> the list of (bus, tuple) that do not balance"""
    tot = {}
    for main, pre, tab in zip(mains, pres, tabs):
        rows = main if pre is None else np.concatenate([pre, main], axis=1)
        rows = rows.astype(np.int64)
        t = [int(x) for x in tab]
        p = 3
        for _ in range(t[1]):
            sign, mult, bus, nv = t[p:p + 4]
            cols = t[p + 4:p + 4 + nv]
            p += 4 + nv
            for r in rows:
                mu = 1 if mult == 0xFFFFFFFF else int(r[mult])
                if mu:
                    key = (bus, tuple(int(r[c]) for c in cols))
                    tot[key] = (tot.get(key, 0) + (mu if sign == SEND else -mu)) % P
    return [k for k, v in tot.items() if v]
