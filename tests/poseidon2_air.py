"""The width-16 Poseidon2 permutation as a constraint program, with Merkle-path chaining (test-side restatement; the product's generator
is csrc/poseidon2_chip.hip).  This is the workhorse of a recursion machine: a STARK verifier inside a STARK spends its rows on Poseidon2
(Merkle paths of the FRI queries, the transcript) -- sp1-recursion's Poseidon2 chips, reference Cargo.lock:6172 ff.

One row = one permutation `out = poseidon2(in)` of tests/pyref.py (this repo's parameter set).  360 columns, every constraint of degree
<= 3 including its selector (log_quotient_degree 1):
  IN   16   the input state
  S0   16   the state after the initial external layer
  X3E[r] 16, OUTE[r] 16 for the eight external rounds r: (y + rc)^3 of the round's input y, and the state after the round
            (S-box x^7 = x^3 * x^3 * x, then the external matrix): inputs are S0 (r = 0), OUTE[r-1] (r = 1..3, 5..7), SP (r = 4)
  S0P[r], X3P[r], SBP[r] for the thirteen internal rounds: element 0 before the S-box, its cube, its seventh power; the other elements
            stay LINEAR FORMS over OUTE[3] and the SBP columns so far (no columns needed)
  SP   16   the state after the internal rounds
  D    8    the digest-carrying half of the input: D[j] = IN[j] (1 - BIT) + IN[8 + j] BIT
  BIT, CH, END  booleans: the node of this row is a RIGHT child; the row continues the path of the previous row (D = the previous
            row's digest OUTE[7][0..8]); the row ends a path (its digest is the public root)
  CNT       running count of END rows; the last row's CNT is the public count
  SPG, SS   booleans for LEAF HASHING (the overwrite-mode sponge over a row of 8 k values): SS = the row absorbs the first 8 values of a leaf
            (its capacity half IN[8..16] is zero); SPG = the row absorbs the next 8 values (its capacity half is the previous row's
            OUTE[7][8..16]).  The row after a leaf's last sponge row starts the path with CH = 1: its D is the leaf digest.
Public values: root[8], count.  What a proof says: "I know `count` openings -- rows or leaf digests, siblings and positions are the
prover's -- that end in `root`" (sponge + truncated-permutation compression, as the commitments of this repo and of p3-merkle-tree).
"""
import numpy as np

import oracle_lib as O
import pyref

P = O.P
V = O.air_var
PARAMS = pyref.PARAMS

IN, S0 = 0, 16
SP, D = 327, 343
BIT, CH, END, CNT, SPG, SS = 351, 352, 353, 354, 355, 356
WIDTH = 360
N_PUBLIC = 9
LNP, KP, M = 357, 358, 359          # the FRI-layers variant's use of the three spare columns (program(fri_layers=True))
TRS, WIDTH_T = 360, 364             # the transcript variant (program(fri_layers=True, transcript=cap_pub)): one more flag column
QP, QF = 361, 362                   # ... and its query-phase rows (program(..., queries=final_pub)): a sponge row whose outputs are sampled; the first of them


def X3E(r):
    return 32 + 32 * r


def OUTE(r):
    return 48 + 32 * r


def S0P(r):
    return 288 + 3 * r


def X3P(r):
    return 289 + 3 * r


def SBP(r):
    return 290 + 3 * r


def ext_input(r):
    return S0 if r == 0 else (SP if r == 4 else OUTE(r - 1))


def _term(coeff, vs):
    return (coeff % P, list(vs))


def _cube_def(x3, c, k):
    """x3 - (c + k)^3 = 0, zero coefficients omitted"""
    t = [_term(1, [V(x3)]), _term(P - 1, [V(c)] * 3), _term(P - 3 * k, [V(c)] * 2), _term(P - 3 * k * k, [V(c)]), _term(P - pow(k, 3, P), [])]
    return [x for x in t if x[0]]


def _linear_def(col, form):
    """col - sum coeff * column = 0; form: {column: coeff}, ascending column order"""
    return [_term(1, [V(col)])] + [_term(P - form[c], [V(c)]) for c in sorted(form) if form[c] % P]


def program(fri_layers=False, n_public=N_PUBLIC, transcript=None, queries=None):
    """fri_layers: the variant wired to the FRI-fold chip (tests/fri_air.py): paths of different depths, no public root (END rows send
    their digest on a bus instead), LNP / KP / M in the spare columns.
    transcript = index of the first of 8 public values (the duplex challenger's capacity as the FRI commit phase finds it): the trace
    then STARTS with transcript rows (TRS = 1, LNP = 0, 1, 2, ...), a sponge chain over the layer roots -- row l absorbs root_l into the
    rate half (sent to the ROOTS table like a path's digest), keeps the capacity of row l - 1 (row 0: the public one), and sends
    (l, out[7], out[6], out[5], out[4]) = the challenge beta_l on a bus of its own.
    queries = index of the first of 4 public values (the final value): the QUERY PHASE of the transcript follows the chain -- row R (QF)
    absorbs (final value [4], proof-of-work witness) over the rate words 0..4 and keeps the rest of the previous row's output (duplex
    challenger: inputs overwrite the front of the rate); every further QP row permutes the previous row's whole output.  The rate
    outputs of the QP rows are the sampled words (out[7] first): the proof-of-work word, then one word per query index -- sent to the
    SAMPLES chip (tests/fri_air.py), which takes their bits"""
    cons = permutation_constraints()
    return _program_flags(cons, fri_layers, n_public, transcript, queries)


def permutation_constraints():
    """the constraints that make a row ONE permutation out = poseidon2(in): columns IN .. SP (343 columns), shared by every variant of the
    chip (and by the recursion machine's chip, tests/recursion_air.py)"""
    ME, rc_e, rc_i, diag = pyref.ME, PARAMS["external_rc"], PARAMS["internal_rc"], PARAMS["internal_diag"]
    cons = []
    for i in range(16):
        cons.append((O.SEL_ALL, _linear_def(S0 + i, {IN + j: ME[i][j] for j in range(16)})))

    def external_round(r):
        c0 = ext_input(r)
        for i in range(16):
            cons.append((O.SEL_ALL, _cube_def(X3E(r) + i, c0 + i, rc_e[r][i])))
        for i in range(16):
            t = [_term(1, [V(OUTE(r) + i)])]
            for j in range(16):
                x3, c, k = V(X3E(r) + j), V(c0 + j), rc_e[r][j]
                t.append(_term(P - ME[i][j], [x3, x3, c]))
                if ME[i][j] * k % P:
                    t.append(_term(P - ME[i][j] * k, [x3, x3]))
            cons.append((O.SEL_ALL, t))
    for r in range(4):
        external_round(r)
    lin = [{OUTE(3) + i: 1} for i in range(16)]
    for r in range(13):
        k = rc_i[r]
        cons.append((O.SEL_ALL, _linear_def(S0P(r), lin[0])))
        cons.append((O.SEL_ALL, _cube_def(X3P(r), S0P(r), k)))
        t = [_term(1, [V(SBP(r))]), _term(P - 1, [V(X3P(r)), V(X3P(r)), V(S0P(r))])]
        if k % P:
            t.append(_term(P - k, [V(X3P(r)), V(X3P(r))]))
        cons.append((O.SEL_ALL, t))
        lin[0] = {SBP(r): 1}
        total = {}
        for f in lin:
            for c, v in f.items():
                total[c] = (total.get(c, 0) + v) % P
        lin = [{c: (diag[i] * lin[i].get(c, 0) + total.get(c, 0)) % P for c in set(lin[i]) | set(total)} for i in range(16)]
    for i in range(16):
        cons.append((O.SEL_ALL, _linear_def(SP + i, lin[i])))
    for r in range(4, 8):
        external_round(r)
    return cons


def _program_flags(cons, fri_layers, n_public, transcript, queries):
    for j in range(8):
        cons.append((O.SEL_ALL, [_term(1, [V(D + j)]), _term(P - 1, [V(IN + j)]), _term(1, [V(BIT), V(IN + j)]), _term(P - 1, [V(BIT), V(IN + 8 + j)])]))
    for b in (BIT, CH, END, SPG, SS):
        cons.append((O.SEL_ALL, [_term(1, [V(b), V(b)]), _term(P - 1, [V(b)])]))
    cons.append((O.SEL_FIRST, [_term(1, [V(CH)])]))
    cons.append((O.SEL_FIRST, [_term(1, [V(SPG)])]))
    for j in range(8):
        cons.append((O.SEL_TRANSITION, [_term(1, [V(SPG, True), V(IN + 8 + j, True)]), _term(P - 1, [V(SPG, True), V(OUTE(7) + 8 + j)])]))
    for j in range(8):
        cons.append((O.SEL_ALL, [_term(1, [V(SS), V(IN + 8 + j)])]))
    for j in range(8):
        cons.append((O.SEL_TRANSITION, [_term(1, [V(CH, True), V(D + j, True)]), _term(P - 1, [V(CH, True), V(OUTE(7) + j)])]))
    if not fri_layers:
        for j in range(8):
            cons.append((O.SEL_ALL, [_term(1, [V(END), V(OUTE(7) + j)]), _term(P - 1, [V(END), V(j, public=True)])]))
    cons.append((O.SEL_FIRST, [_term(1, [V(CNT)]), _term(P - 1, [V(END)])]))
    cons.append((O.SEL_TRANSITION, [_term(1, [V(CNT, True)]), _term(P - 1, [V(CNT)]), _term(P - 1, [V(END, True)])]))
    if not fri_layers:
        cons.append((O.SEL_LAST, [_term(1, [V(CNT)]), _term(P - 1, [V(8, public=True)])]))
    else:
        cons.append((O.SEL_TRANSITION, [_term(1, [V(SS)]), _term(P - 1, [V(SS), V(CH, True)]), _term(1, [V(CH)]), _term(P - 1, [V(CH), V(CH, True)]),
                                        _term(P - 1, [V(END)]), _term(1, [V(END), V(CH, True)])]))
        cons.append((O.SEL_LAST, [_term(1, [V(SS)]), _term(1, [V(CH)]), _term(P - 1, [V(END)])]))
        cons.append((O.SEL_ALL, [_term(1, [V(END)]), _term(P - 1, [V(END), V(CH)])]))
        cons.append((O.SEL_ALL, [_term(1, [V(SS), V(BIT)])]))
        cons.append((O.SEL_ALL, [_term(1, [V(SS), V(CH)])]))
        cons.append((O.SEL_TRANSITION, [_term(1, [V(CH, True), V(LNP, True)]), _term(P - 1, [V(CH, True), V(LNP)])]))
        cons.append((O.SEL_TRANSITION, [_term(1, [V(CH, True), V(KP)]), _term(P - 2, [V(CH, True), V(KP, True)]), _term(P - 1, [V(CH, True), V(BIT)])]))
        cons.append((O.SEL_ALL, [_term(1, [V(END), V(KP)]), _term(P - 1, [V(END), V(BIT)])]))
        cons.append((O.SEL_ALL, [_term(1, [V(M)]), _term(P - 1, [V(M), V(SS)])]))
    if transcript is not None:
        assert fri_layers
        cons.append((O.SEL_ALL, [_term(1, [V(TRS), V(TRS)]), _term(P - 1, [V(TRS)])]))
        cons.append((O.SEL_FIRST, [_term(1, [V(TRS)]), _term(P - 1, [])]))                                   # the trace starts with the transcript
        cons.append((O.SEL_FIRST, [_term(1, [V(LNP)])]))
        for j in range(8):
            cons.append((O.SEL_FIRST, [_term(1, [V(IN + 8 + j)]), _term(P - 1, [V(transcript + j, public=True)])]))
        cons.append((O.SEL_TRANSITION, [_term(1, [V(TRS, True)]), _term(P - 1, [V(TRS), V(TRS, True)])]))      # transcript rows are a prefix
        if queries is None:
            cons.append((O.SEL_TRANSITION, [_term(1, [V(TRS, True), V(LNP, True)]), _term(P - 1, [V(TRS, True), V(LNP)]), _term(P - 1, [V(TRS, True)])]))
        cons.append((O.SEL_TRANSITION, [_term(1, [V(TRS, True)]), _term(P - 1, [V(TRS, True), V(SPG, True)])]))   # ... chained through the capacity
        if queries is None:
            cons.append((O.SEL_ALL, [_term(1, [V(SPG)]), _term(P - 1, [V(SPG), V(TRS)])]))                     # and nothing else is
        for f in (CH, END, SS, BIT, M):
            cons.append((O.SEL_ALL, [_term(1, [V(TRS), V(f)])]))
        if queries is not None:
            o7 = OUTE(7)
            for f in (QP, QF):
                cons.append((O.SEL_ALL, [_term(1, [V(f), V(f)]), _term(P - 1, [V(f)])]))
            cons.append((O.SEL_ALL, [_term(1, [V(QF)]), _term(P - 1, [V(QF), V(QP)])]))                          # the first query row is one
            cons.append((O.SEL_ALL, [_term(1, [V(QP), V(TRS)])]))
            cons.append((O.SEL_FIRST, [_term(1, [V(QF)])]))
            cons.append((O.SEL_TRANSITION, [_term(1, [V(QF, True)]), _term(P - 1, [V(TRS)]), _term(1, [V(TRS), V(TRS, True)])]))      # QF' = TRS (1 - TRS'): right behind the chain
            cons.append((O.SEL_TRANSITION, [_term(1, [V(QP, True)]), _term(P - 1, [V(QP, True), V(QP)]), _term(P - 1, [V(QF, True)])]))  # a query row is the first or follows one
            # the row counter runs on through both kinds of rows
            cons.append((O.SEL_TRANSITION, [_term(1, [V(TRS, True), V(LNP, True)]), _term(P - 1, [V(TRS, True), V(LNP)]), _term(P - 1, [V(TRS, True)]),
                                            _term(1, [V(QP, True), V(LNP, True)]), _term(P - 1, [V(QP, True), V(LNP)]), _term(P - 1, [V(QP, True)])]))
            cons.append((O.SEL_TRANSITION, [_term(1, [V(QP, True)]), _term(P - 1, [V(QP, True), V(SPG, True)])]))  # the capacity is kept
            cons.append((O.SEL_ALL, [_term(1, [V(SPG)]), _term(P - 1, [V(SPG), V(TRS)]), _term(P - 1, [V(SPG), V(QP)])]))
            for j in range(5, 8):                                                                                # rate words the inputs do not reach
                cons.append((O.SEL_TRANSITION, [_term(1, [V(QP, True), V(IN + j, True)]), _term(P - 1, [V(QP, True), V(o7 + j)])]))
            for j in range(5):                                                                                   # later rows: nothing absorbed
                cons.append((O.SEL_TRANSITION, [_term(1, [V(QP, True), V(IN + j, True)]), _term(P - 1, [V(QP, True), V(o7 + j)]),
                                                _term(P - 1, [V(QF, True), V(IN + j, True)]), _term(1, [V(QF, True), V(o7 + j)])]))
            for j in range(4):                                                                                   # the first absorbs the final value (and a free witness)
                cons.append((O.SEL_ALL, [_term(1, [V(QF), V(IN + j)]), _term(P - 1, [V(QF), V(queries + j, public=True)])]))
            for f in (CH, END, SS, BIT, M):
                cons.append((O.SEL_ALL, [_term(1, [V(QP), V(f)])]))
        return O.air_program(WIDTH_T, n_public, cons)
    return O.air_program(WIDTH, n_public, cons)


def layer_paths_trace(paths, log_n, transcript=None, queries=None):
    """the FRI-layers variant's trace: paths = [(layer, leaf index, pair [8 values], siblings [[8] x depth], multiplicity)], one leaf row
    (the sponge over the pair) + depth compression rows each -> (trace [2^log_n][WIDTH], roots).
    transcript = (capacity [8], layer roots [[8] x R]): the transcript variant -- R sponge rows over the roots come first, the rows are
    WIDTH_T wide -> (trace, roots, betas)
    queries = (final value [4], witness, rows): `rows` query-phase sponge rows follow the chain -> (trace, roots, betas, samples), samples
    = [rows][8] in the order the challenger hands them out (out[7] first)"""
    rows, roots, cnt = [], [], 0
    betas = []
    if transcript is not None:
        cap, layer_roots = [int(v) % P for v in transcript[0]], transcript[1]
        for l, root in enumerate(layer_roots):
            r, out = row([int(v) % P for v in root] + cap, 0, 0, 0, 0, 1 if l else 0, 0)
            r[LNP] = l
            rows.append(r + [1, 0, 0, 0])
            betas.append([out[7], out[6], out[5], out[4]])
            cap = out[8:]
        samples = []
        if queries is not None:
            final, witness, n_rows = queries
            state = [int(v) % P for v in final] + [int(witness) % P] + out[5:8] + cap
            for i in range(n_rows):
                r, out = row(state, 0, 0, 0, 0, 1, 0)
                r[LNP] = len(layer_roots) + i
                rows.append(r + [0, 1, 1 if i == 0 else 0, 0])
                samples.append([out[7 - j] for j in range(8)])
                state = list(out)
    for layer, index, pair, sibs, mult in paths:
        r, out = row([int(v) % P for v in pair] + [0] * 8, 0, 0, 0, cnt, 0, 1)
        r[LNP], r[KP], r[M] = layer, 2 * index % P, mult
        rows.append(r)
        digest = out[:8]
        depth = len(sibs)
        for lvl in range(depth):
            bit = (index >> lvl) & 1
            sib = [int(v) % P for v in sibs[lvl]]
            end = 1 if lvl == depth - 1 else 0
            cnt += end
            r, out = row(sib + digest if bit else digest + sib, bit, 1, end, cnt)
            r[LNP], r[KP] = layer, index >> lvl
            rows.append(r)
            digest = out[:8]
        roots.append(digest)
    pad, _ = row([0] * 16, 0, 0, 0, cnt)
    if transcript is not None:
        rows = [r if len(r) == WIDTH_T else r + [0, 0, 0, 0] for r in rows]
        pad = pad + [0, 0, 0, 0]
    assert len(rows) <= 1 << log_n
    rows += [pad] * ((1 << log_n) - len(rows))
    if queries is not None:
        return np.array(rows, dtype=np.uint64).astype(np.uint32), roots, betas, samples
    if transcript is not None:
        return np.array(rows, dtype=np.uint64).astype(np.uint32), roots, betas
    return np.array(rows, dtype=np.uint64).astype(np.uint32), roots


def row(state_in, bit=0, ch=0, end=0, cnt=0, spg=0, ss=0):
    """one trace row: every intermediate of poseidon2(state_in) -> (row, output state)"""
    ME, MI, rc_e, rc_i = pyref.ME, pyref.MI, PARAMS["external_rc"], PARAMS["internal_rc"]
    t = [0] * WIDTH
    s = [x % P for x in state_in]
    t[IN:IN + 16] = s
    s = pyref._matvec(ME, s)
    t[S0:S0 + 16] = s

    def external_round(r, s):
        y = [(s[i] + rc_e[r][i]) % P for i in range(16)]
        x3 = [pow(v, 3, P) for v in y]
        t[X3E(r):X3E(r) + 16] = x3
        s = pyref._matvec(ME, [x3[i] * x3[i] % P * y[i] % P for i in range(16)])
        t[OUTE(r):OUTE(r) + 16] = s
        return s
    for r in range(4):
        s = external_round(r, s)
    for r in range(13):
        t[S0P(r)] = s[0]
        y = (s[0] + rc_i[r]) % P
        t[X3P(r)] = pow(y, 3, P)
        s[0] = t[SBP(r)] = pow(y, 7, P)
        s = pyref._matvec(MI, s)
    t[SP:SP + 16] = s
    for r in range(4, 8):
        s = external_round(r, s)
    for j in range(8):
        t[D + j] = state_in[8 + j] % P if bit else state_in[j] % P
    t[BIT], t[CH], t[END], t[CNT], t[SPG], t[SS] = bit, ch, end, cnt % P, spg, ss
    return t, s


def merkle_trace(leaves, siblings, indices, log_n=None, hashed_rows=False):
    """paths p: leaf digest leaves[p] (8 values) -- or, with hashed_rows, the opened ROW leaves[p] (8 k values), hashed by k sponge rows
    first --, siblings[p][level] (8 values each), indices[p] (bit `level`: the node is a right child)
    -> (trace [2^log_n][WIDTH], roots [n_paths][8]); rows after the paths are permutations of the zero state with no flags"""
    n_paths, depth = len(leaves), len(siblings[0])
    rows, roots, cnt = [], [], 0
    for p in range(n_paths):
        digest = [int(v) % P for v in leaves[p]]
        if hashed_rows:
            vals, cap = digest, [0] * 8
            assert len(vals) % 8 == 0 and vals
            for k in range(0, len(vals), 8):
                r, out = row(vals[k:k + 8] + cap, 0, 0, 0, cnt, 1 if k else 0, 0 if k else 1)
                rows.append(r)
                cap = out[8:]
            digest = out[:8]
        for lvl in range(depth):
            bit = (int(indices[p]) >> lvl) & 1
            sib = [int(v) % P for v in siblings[p][lvl]]
            end = 1 if lvl == depth - 1 else 0
            cnt += end
            r, out = row(sib + digest if bit else digest + sib, bit, 1 if (lvl or hashed_rows) else 0, end, cnt)
            rows.append(r)
            digest = out[:8]
        roots.append(digest)
    need = max(len(rows), 32)
    if log_n is None:
        log_n = max(5, (need - 1).bit_length())
    pad, _ = row([0] * 16, 0, 0, 0, cnt)
    rows += [pad] * ((1 << log_n) - len(rows))
    return np.array(rows, dtype=np.uint64).astype(np.uint32), roots


def tree_paths(n_leaves_log, n_paths, seed=1):
    """a random tree of 2^n_leaves_log leaf digests and n_paths openings of it -> (leaves, siblings, indices, root)"""
    rng = np.random.default_rng(seed)
    level = [[int(v) for v in rng.integers(0, P, 8)] for _ in range(1 << n_leaves_log)]
    levels = [level]
    while len(level) > 1:
        level = [pyref.compress(level[2 * i], level[2 * i + 1]) for i in range(len(level) // 2)]
        levels.append(level)
    idx = [int(v) for v in rng.integers(0, 1 << n_leaves_log, n_paths)]
    leaves = [levels[0][i] for i in idx]
    sibs = [[levels[l][(i >> l) ^ 1] for l in range(n_leaves_log)] for i in idx]
    return leaves, sibs, idx, levels[-1][0]
