"""The FRI-fold chip and its machine, written a second time (the first is zktls_amd/csrc/fri_chip.hip): the constraint program, the
trace of a view, the OPENINGS table and the two-chip keyed machine -- in plain Python on tests/pyref.py's field arithmetic.  The tests
require the program words, the trace and the table to be EQUAL to the library's, and the oracle's generic keyed-machine prover on
these arrays to produce the library's proof bytes.

A view = what the FRI check of a shard proof reads (fold by 2, constant final value): betas[R] (extension elements), final value,
per query (index, reduced opening, R siblings).  tests/pyverify.py hands one out while verifying (view=...), and so does the
library (zkhip_fri_view_shard): the two must agree before anything is built on them."""
import numpy as np

import oracle_lib as O
from pyref import P, bitrev, ext_mul, two_adic_generator

V = O.air_var
E0, E1, BIT, K, X, XI, S, T, B, BETA, FOLD, ACTIVE, LN, G, GS, GT, OWN, L = 0, 4, 8, 9, 10, 11, 12, 13, 14, 15, 19, 23, 24, 25, 26, 27, 28, 32
BUS_E0, BUS_E1, BUS_R0, BUS_R1, BUS_Q = 40, 41, 42, 43, 44
BUS_B, BUS_BF = 45, 46                         # the transcript machine: (layer, beta) from the Poseidon2 chip's transcript rows to the ROOTS table, and from there to the fold rows
ROOTS_MAIN_T = 8                               # ... whose MAIN row is then (paths + 1, beta[4], queries, 0, 0); preprocessed (layer, root[8], 1, 0, 0)
N_PUBLIC_T = 12                                # final value, the challenger's capacity
BUS_S0, BUS_S1, BUS_I = 47, 48, 49             # the query-phase machine: the sampled words of a sponge row (two halves) to the SAMPLES chip, (query, index) from there to QUERIES
# SAMPLES chip (one row per query-phase sponge row): preprocessed C (the sponge row's number), ROW, ACT[8] (word j is a query index), POW (row 0: word 0
# is the proof-of-work sample), KQ[8] (the query's number); main W[8] (the words), IDX[8] (their low bits), H1 H2 HH [8] (canonical-form helpers), bits [8][31]
S_PRE, S_C, S_ROW, S_ACT, S_POW, S_KQ = 20, 0, 1, 2, 10, 11
S_W, S_IDX, S_H1, S_H2, S_HH, S_BITS, S_MAIN = 0, 8, 16, 24, 32, 40, 288
K2, IDX, L_WIRED = 32, 33, 34                  # the wired form's extra columns, its layer selectors start two columns later
XS, PT, LNX, L_REC = 34, 35, 36, 37            # the recursion machine's form (tests/recursion_air.py): XS = X (1 - 2 BIT), the query's point / g on its first row;
#                                                PT = (number of the inner proof) x (its trees), constant along a chain; LNX = PT + LN names the layer's tree on the buses
BUS_FIN = 60                                   # ... whose END rows send their folded value (the final value is an observed word of the transcript, no public value)
QUERIES_PRE, ROOTS_PRE = 8, 12
OPEN_PRE, OPEN_MAIN = 12, 4
INV2 = (P + 1) // 2
EXT_W = 11


def width_of(layers, wired=False, rec=False, inject=None):
    w = ((L_REC if rec else L_WIRED if wired else L) + layers + 3) & ~3
    return w + 8 if inject is not None else w          # (machine mode: INJ [4], INJF behind the recursion form's columns)


def inj_cols(layers):
    """machine mode (tests/recursion_machine.py): INJ = what joins the folded value at this row (the reduced opening of the height just reached), INJF = the row has one;
    IDX0 (at INJF + 1) = the query's index, constant along its chain: what joins is named by the query it joins"""
    w = width_of(layers, rec=True)
    return w, w + 4


def n_public_of(layers):
    return 4 * layers + 4


def root_const(l):
    """c_l = w_{2^(l+1)}: what bit l of a query index contributes to its evaluation point"""
    return two_adic_generator(l + 1)


def program(layers, wired=False, transcript=False, rec=None, inject=None):
    """transcript: the fold chip of the TRANSCRIPT machine -- the challenges are no public values any more (the rows receive
    (layer, beta) on a bus from the ROOTS table, which has them from the Poseidon2 chip's transcript rows); public values: the final
    value, then the challenger's capacity (8 words, read by the Poseidon2 chip)"""
    R = layers
    L = L_REC if rec is not None else L_WIRED if wired else globals()["L"]
    END = L + R - 1
    cons = []
    assert rec is None or (wired and transcript)          # rec = the number of public values of the recursion machine's programs

    def add(sel, terms):
        cons.append((sel, [(c % P, list(vs)) for c, vs in terms if c % P]))

    def gated(terms):
        """G * terms, G = ACTIVE - END: an active row that is not the last of its query (a column: a selector counts one degree)"""
        return [(c, [V(G)] + list(vs)) for c, vs in terms]
    add(O.SEL_ALL, [(1, [V(ACTIVE)])] + [(P - 1, [V(L + l)]) for l in range(R)])
    add(O.SEL_ALL, [(1, [V(LN)])] + [(P - l, [V(L + l)]) for l in range(R)])
    add(O.SEL_ALL, [(1, [V(ACTIVE), V(ACTIVE)]), (P - 1, [V(ACTIVE)])])
    add(O.SEL_ALL, [(1, [V(BIT), V(BIT)]), (P - 1, [V(BIT)])])
    for l in range(R):
        add(O.SEL_ALL, [(1, [V(L + l), V(L + l)]), (P - 1, [V(L + l)])])
    if not transcript:
        for j in range(4):
            add(O.SEL_ALL, [(1, [V(BETA + j)])] + [(P - 1, [V(L + l), V(4 * l + j, public=True)]) for l in range(R)])
    add(O.SEL_ALL, [(1, [V(G)]), (P - 1, [V(ACTIVE)]), (1, [V(END)])])
    add(O.SEL_ALL, [(1, [V(S)]), (P - 1, [V(X), V(X)])])
    add(O.SEL_ALL, [(1, [V(GS)]), (P - 1, [V(G), V(S)])])
    add(O.SEL_ALL, [(1, [V(GT)]), (P - 1, [V(G), V(T)])])
    for j in range(4):
        add(O.SEL_ALL, [(1, [V(OWN + j)]), (P - 1, [V(E0 + j)]), (1, [V(BIT), V(E0 + j)]), (P - 1, [V(BIT), V(E1 + j)])])
    add(O.SEL_ALL, [(1, [V(ACTIVE), V(X), V(XI)]), (P - 1, [V(ACTIVE)])])
    add(O.SEL_ALL, [(1, [V(T)]), (P - 1, []), (1, [V(BIT)])] + [(P - root_const(l), [V(BIT), V(L + l)]) for l in range(R)])
    for j in range(4):
        t = [(1, [V(FOLD + j)]), (P - INV2, [V(E0 + j)]), (P - INV2, [V(E1 + j)])]
        for a in range(4):
            for d in range(4):
                if (a + d) % 4 != j:
                    continue
                w = INV2 * EXT_W % P if a + d >= 4 else INV2
                t.append((P - w, [V(BETA + a), V(E0 + d), V(XI)]))
                t.append((w, [V(BETA + a), V(E1 + d), V(XI)]))
        add(O.SEL_ALL, t)
    for l in range(R - 1):
        add(O.SEL_TRANSITION, [(1, [V(L + l + 1, True)]), (P - 1, [V(L + l)])])
    add(O.SEL_TRANSITION, gated([(1, [V(K)]), (P - 2, [V(K, True)]), (P - 1, [V(BIT, True)])]))
    add(O.SEL_ALL, [(1, [V(END), V(K), V(K)]), (P - 1, [V(END), V(K)])])
    add(O.SEL_TRANSITION, [(1, [V(G), V(X, True)]), (P - 1, [V(GS)]), (2, [V(GS), V(BIT, True)])])
    add(O.SEL_TRANSITION, [(1, [V(G), V(B)]), (P - 1, [V(GT), V(B, True)])])
    add(O.SEL_ALL, [(1, [V(END), V(B)]), (P - 1, [V(END), V(T)]), (P - (root_const(R) - 1), [V(END), V(T), V(K)])])
    add(O.SEL_TRANSITION, [(1, [V(L), V(X)]), (P - 1, [V(L), V(B, True)])])
    if inject is None:
        for j in range(4):
            add(O.SEL_TRANSITION, gated([(1, [V(FOLD + j)]), (P - 1, [V(OWN + j, True)])]))
    else:                                              # inject: the layers (>= 1) at whose row the height just reached adds its reduced opening
        assert rec is not None and all(1 <= l < R for l in inject)
        INJ, INJF = inj_cols(R)
        for j in range(4):
            add(O.SEL_TRANSITION, gated([(1, [V(FOLD + j)]), (1, [V(INJ + j, True)]), (P - 1, [V(OWN + j, True)])]))
        add(O.SEL_ALL, [(1, [V(INJF)])] + [(P - 1, [V(L + l)]) for l in inject])
        for j in range(4):
            add(O.SEL_ALL, [(1, [V(INJ + j)]), (P - 1, [V(INJF), V(INJ + j)])])
        IDX0 = INJF + 1
        add(O.SEL_ALL, [(1, [V(L), V(IDX0)]), (P - 1, [V(L), V(IDX)])])                 # the chain's first row: the query's index
        add(O.SEL_TRANSITION, gated([(1, [V(IDX0)]), (P - 1, [V(IDX0, True)])]))
    if rec is None:
        for j in range(4):
            add(O.SEL_ALL, [(1, [V(END), V(FOLD + j)]), (P - 1, [V(END), V((0 if transcript else 4 * R) + j, public=True)])])
    if wired:
        add(O.SEL_ALL, [(1, [V(K2)]), (P - 2, [V(K)])])
        add(O.SEL_ALL, [(1, [V(IDX)]), (P - 1, [V(K2)]), (P - 1, [V(BIT)])])
    if rec is not None:
        add(O.SEL_ALL, [(1, [V(XS)]), (P - 1, [V(X)]), (2, [V(X), V(BIT)])])
        add(O.SEL_ALL, [(1, [V(LNX)]), (P - 1, [V(PT)]), (P - 1, [V(LN)])])
        add(O.SEL_TRANSITION, gated([(1, [V(PT)]), (P - 1, [V(PT, True)])]))
        return O.air_program(width_of(R, rec=True, inject=inject), rec, cons)
    return O.air_program(width_of(R, wired), N_PUBLIC_T if transcript else n_public_of(R), cons)


def fold_pair(k, lh, beta, e0, e1):
    """(e0 + e1)/2 + beta (e0 - e1)/(2 x), x = w_{2^(lh+1)}^bitrev_lh(k); -> (folded, x)"""
    x = pow(two_adic_generator(lh + 1), bitrev(k, lh), P)
    xi = pow(x, P - 2, P)
    even = [(a + b) * INV2 % P for a, b in zip(e0, e1)]
    odd = [(a - b) * INV2 % P * xi % P for a, b in zip(e0, e1)]
    return [(a + b) % P for a, b in zip(even, ext_mul(beta, odd))], x, xi


def log_rows_of(layers, n_queries):
    lr = 5
    while (1 << lr) < n_queries * layers:
        lr += 1
    return lr


def trace(view, log_rows=None, wired=False, rec=False, pt=0, inject=None):
    """-> (trace [2^log_rows][width] canonical, final value): one row per (query, layer), padding rows zero with T = 1.
    inject: per query {layer: value that joins the folded value at that layer's row} (machine mode)"""
    betas, queries = view["betas"], view["queries"]
    R = len(betas)
    H = R + 1
    W = width_of(R, wired, rec, inject=inject)
    L = L_REC if rec else L_WIRED if wired else globals()["L"]
    lr = log_rows if log_rows is not None else log_rows_of(R, len(queries))
    t = np.zeros((1 << lr, W), dtype=np.uint64)
    t[:, T] = 1
    final = None
    for q, (index, value, sibs) in enumerate(queries):
        idx, own = index, list(value)
        tcol = []
        for l in range(R):
            row = t[q * R + l]
            if inject is not None:
                row[inj_cols(R)[1] + 1] = index
            if inject is not None and l in inject[q]:
                INJ, INJF = inj_cols(R)
                row[INJ:INJ + 4], row[INJF] = inject[q][l], 1
                own = [(a + b) % P for a, b in zip(own, inject[q][l])]
            bit, k = idx & 1, idx >> 1
            e0, e1 = (sibs[l], own) if bit else (own, sibs[l])
            fold, x, xi = fold_pair(k, H - (l + 1), betas[l], e0, e1)
            row[E0:E0 + 4], row[E1:E1 + 4], row[BETA:BETA + 4], row[FOLD:FOLD + 4] = e0, e1, betas[l], fold
            row[BIT], row[K], row[X], row[XI], row[S] = bit, k, x, xi, x * x % P
            tcol.append(root_const(l) if bit else 1)
            row[T], row[ACTIVE], row[LN], row[L + l] = tcol[-1], 1, l, 1
            if l + 1 < R:
                row[G], row[GS], row[GT] = 1, x * x % P, tcol[-1]
            row[OWN:OWN + 4] = own
            if wired:
                row[K2], row[IDX] = 2 * k, 2 * k + bit
            if rec:
                row[XS] = (P - x) % P if bit else x
                row[PT], row[LNX] = pt, pt + l
            own, idx = fold, k
        acc = root_const(R) if idx else 1
        for l in reversed(range(R)):
            acc = acc * tcol[l] % P
            t[q * R + l, B] = acc
        assert final is None or final == own, "the chains do not end in one value"
        final = own
    return t.astype(np.uint32), final


def openings(view, log_rows):
    """the preprocessed table: one row per distinct (layer, pair), ascending, (ln, k, e0[4], e1[4], multiplicity, 0)"""
    betas, queries = view["betas"], view["queries"]
    R = len(betas)
    H = R + 1
    rows = {}
    for index, value, sibs in queries:
        idx, own = index, list(value)
        for l in range(R):
            bit, k = idx & 1, idx >> 1
            e0, e1 = (sibs[l], own) if bit else (own, sibs[l])
            key = (l, k)
            if key in rows:
                assert rows[key][0] == list(e0) + list(e1)
                rows[key][1] += 1
            else:
                rows[key] = [list(e0) + list(e1), 1]
            own, idx = fold_pair(k, H - (l + 1), betas[l], e0, e1)[0], k
    t = np.zeros((1 << log_rows, OPEN_PRE), dtype=np.uint32)
    for r, key in enumerate(sorted(rows)):
        t[r, 0], t[r, 1], t[r, 2:10], t[r, 10] = key[0], key[1], rows[key][0], rows[key][1]
    return t


def machine(view):
    """-> (main traces, preprocessed traces, programs, interaction tables, public values), the FRI chip first"""
    R = len(view["betas"])
    lr = log_rows_of(R, len(view["queries"]))
    tr, final = trace(view, lr)
    pre = openings(view, lr)
    open_prog = O.air_program(OPEN_PRE + OPEN_MAIN, n_public_of(R), [(O.SEL_FIRST, [(1, [V(OPEN_PRE + 3)])])])
    fri_tab = O.interaction_table([(O.SEND, ACTIVE, BUS_E0, [LN, K, E0, E0 + 1, E0 + 2, E0 + 3]), (O.SEND, ACTIVE, BUS_E1, [LN, K, E1, E1 + 1, E1 + 2, E1 + 3])])
    open_tab = O.interaction_table([(O.RECEIVE, 10, BUS_E0, [0, 1, 2, 3, 4, 5]), (O.RECEIVE, 10, BUS_E1, [0, 1, 6, 7, 8, 9])])
    pub = [c for b in view["betas"] for c in b] + list(final)
    return [tr, np.zeros((1 << lr, OPEN_MAIN), dtype=np.uint32)], [None, pre], [program(R), open_prog], [fri_tab, open_tab], pub


def random_view(layers, n_queries, seed=1):
    """a consistent view that belongs to no proof: random layer vectors folded honestly (queries that meet share their pairs)"""
    rng = np.random.default_rng(seed)
    H = layers + 1
    betas = [[int(v) for v in rng.integers(0, P, 4)] for _ in range(layers)]
    vec = [[int(v) for v in rng.integers(0, P, 4)] for _ in range(1 << H)]
    layers_vec = [vec]
    for l in range(layers):
        cur = layers_vec[-1]
        lh = H - (l + 1)
        layers_vec.append([fold_pair(k, lh, betas[l], cur[2 * k], cur[2 * k + 1])[0] for k in range(len(cur) // 2)])
    # a constant final value needs a low-degree start; instead of constructing one, fold to the top and require nothing of the final
    # layer beyond what the chip checks per query: the chains of all queries must END IN ONE VALUE, so pick queries in one top half
    top = int(rng.integers(0, 2))
    queries = []
    for _ in range(n_queries):
        index = (top << layers) | int(rng.integers(0, 1 << layers))
        idx, sibs = index, []
        for l in range(layers):
            sibs.append(layers_vec[l][idx ^ 1])
            idx >>= 1
        queries.append((index, layers_vec[0][index], sibs))
    return {"betas": betas, "queries": queries, "final": layers_vec[layers][top]}


def sample_rows(n_queries):
    """sponge rows of the query phase: one proof-of-work word, then one word per query, 8 words per row"""
    return (1 + n_queries + 7) // 8


def samples_program(layers, n_queries, pow_bits, n_public):
    """every word = sum of 31 bits, canonical (bits 27..30 all set -> bits 0..26 clear: P = 2^31 - 2^27 + 1), IDX = the low layers + 1 bits; the
    proof-of-work word's low pow_bits bits are zero.  Columns: the preprocessed ones first"""
    cons = []

    def add(sel, terms):
        cons.append((sel, [(c % P, list(vs)) for c, vs in terms if c % P]))
    M0 = S_PRE
    for j in range(8):
        bits = [M0 + S_BITS + 31 * j + i for i in range(31)]
        for b in bits:
            add(O.SEL_ALL, [(1, [V(b), V(b)]), (P - 1, [V(b)])])
        add(O.SEL_ALL, [(1, [V(M0 + S_W + j)])] + [(P - (1 << i), [V(bits[i])]) for i in range(31)])
        add(O.SEL_ALL, [(1, [V(M0 + S_H1 + j)]), (P - 1, [V(bits[30]), V(bits[29])])])
        add(O.SEL_ALL, [(1, [V(M0 + S_H2 + j)]), (P - 1, [V(bits[28]), V(bits[27])])])
        add(O.SEL_ALL, [(1, [V(M0 + S_HH + j)]), (P - 1, [V(M0 + S_H1 + j), V(M0 + S_H2 + j)])])
        add(O.SEL_ALL, [(1, [V(M0 + S_HH + j), V(bits[i])]) for i in range(27)])
        add(O.SEL_ALL, [(1, [V(M0 + S_IDX + j)])] + [(P - (1 << i), [V(bits[i])]) for i in range(layers + 1)])
    if pow_bits:
        add(O.SEL_ALL, [(1, [V(S_POW), V(M0 + S_BITS + i)]) for i in range(pow_bits)])
    return O.air_program(S_PRE + S_MAIN, n_public, cons)


def samples_tables(layers, n_queries, words, log_rows, base=None):
    """-> (preprocessed, main) of the SAMPLES chip; words = [rows][8] sampled words, canonical; base: the number of the first query-phase
    sponge row (the machines of this file: `layers`)"""
    n_rows = sample_rows(n_queries)
    pre = np.zeros((1 << log_rows, S_PRE), dtype=np.uint32)
    main = np.zeros((1 << log_rows, S_MAIN), dtype=np.uint32)
    indices = []
    for r in range(n_rows):
        pre[r, S_C], pre[r, S_ROW] = (layers if base is None else base) + r, 1
        for j in range(8):
            slot = 8 * r + j
            w = int(words[r][j])
            assert 0 <= w < P
            if slot == 0:
                pre[r, S_POW] = 1
            elif slot <= n_queries:
                pre[r, S_ACT + j], pre[r, S_KQ + j] = 1, slot - 1
                indices.append(w & ((1 << (layers + 1)) - 1))
            main[r, S_W + j], main[r, S_IDX + j] = w, w & ((1 << (layers + 1)) - 1)
            b = [(w >> i) & 1 for i in range(31)]
            main[r, S_BITS + 31 * j:S_BITS + 31 * j + 31] = b
            main[r, S_H1 + j], main[r, S_H2 + j], main[r, S_HH + j] = b[30] & b[29], b[28] & b[27], b[30] & b[29] & b[28] & b[27]
    return pre, main, indices


def machine_layers(view, capacity=None, query_phase=None):
    """the wired machine (zkhip_prove_fri_layers): the Poseidon2 chip's FRI-layers variant (one Merkle path per (query, layer)), the fold
    chip in its wired form, and the preprocessed QUERIES / ROOTS tables -> (main traces, preprocessed traces, programs, tables, public values).
    capacity (8 words: the duplex challenger's capacity as the commit phase finds it): the TRANSCRIPT machine (zkhip_prove_fri_transcript)
    -- the Poseidon2 chip's trace starts with a sponge chain over the layer roots whose outputs must be the betas of the ROOTS table.
    query_phase = (proof-of-work witness, pow_bits): the QUERY-PHASE machine (zkhip_prove_fri_indices) -- the chain goes on through the final value
    and the witness, a fifth chip (SAMPLES) takes the bits of the sampled words, and the QUERIES table's index column is the prover's, tied to them:
    the key holds no index any more"""
    import poseidon2_air as P2
    betas, queries, roots, paths = view["betas"], view["queries"], view["roots"], view["paths"]
    R, Q = len(betas), len(queries)
    H = R + 1
    T = capacity is not None
    NP = N_PUBLIC_T if T else n_public_of(R)

    def lg(n, lo=5):
        l = lo
        while (1 << l) < n:
            l += 1
        return l
    # the pairs of every (query, layer) and their paths
    plist = []
    for q, (index, value, sibs) in enumerate(queries):
        idx, own = index, list(value)
        for l in range(R):
            bit, k = idx & 1, idx >> 1
            e0, e1 = (sibs[l], own) if bit else (own, sibs[l])
            plist.append((l, k, list(e0) + list(e1), paths[q][l], 1))
            own, idx = fold_pair(k, H - (l + 1), betas[l], e0, e1)[0], k
    QM = query_phase is not None
    assert T or not QM
    lr_p2 = lg(Q * (R + R * (R + 1) // 2) + (R if T else 0) + (sample_rows(Q) if QM else 0))
    if QM:
        p2_trace, p2_roots, chain, words = P2.layer_paths_trace(plist, lr_p2, transcript=(capacity, roots), queries=(view["final"], query_phase[0], sample_rows(Q)))
        assert chain == [list(b) for b in betas], "the challenges are not the sponge chain over the roots"
        lr_s = lg(sample_rows(Q))
        spre, smain, drawn = samples_tables(R, Q, words, lr_s)
        assert drawn == [index for index, _, _ in view["queries"]], "the query indices are not the ones the transcript draws"
        assert query_phase[1] == 0 or words[0][0] & ((1 << query_phase[1]) - 1) == 0, "the witness does not satisfy the proof of work"
    elif T:
        p2_trace, p2_roots, chain = P2.layer_paths_trace(plist, lr_p2, transcript=(capacity, roots))
        assert chain == [list(b) for b in betas], "the challenges are not the sponge chain over the roots"
    else:
        p2_trace, p2_roots = P2.layer_paths_trace(plist, lr_p2)
    for i, (l, k, pair, sibs, mult) in enumerate(plist):
        assert p2_roots[i] == list(roots[l]), "a path does not end in its layer's root"
    lr_fri = lg(Q * R)
    fri_trace, final = trace(view, lr_fri, wired=True)
    lr_r = max(lg(R), lg(sample_rows(Q))) if QM else lg(R)       # tallest first: the SAMPLES chip comes last
    lr_q = max(lg(Q), lr_r)
    qpre = np.zeros((1 << lr_q, QUERIES_PRE), dtype=np.uint32)
    qmain = np.zeros((1 << lr_q, 4), dtype=np.uint32)
    for q, (index, value, _) in enumerate(view["queries"]):
        qpre[q, 0], qpre[q, 1:5], qpre[q, 5] = (q if QM else index), value, 1
        if QM:
            qmain[q, 0] = index
    RP = ROOTS_PRE
    rpre = np.zeros((1 << lr_r, RP), dtype=np.uint32)
    rmain = np.zeros((1 << lr_r, ROOTS_MAIN_T if T else 4), dtype=np.uint32)
    for l in range(R):
        rpre[l, 0], rpre[l, 1:9] = l, roots[l]
        rmain[l, 0] = Q + (1 if T else 0)              # the paths that end in the root, and the transcript row that absorbs it
        if T:                                          # the challenge is the PROVER's (main columns): the buses tie it to the transcript rows
            rpre[l, 9], rmain[l, 1:5], rmain[l, 5] = 1, betas[l], Q
    o7 = P2.OUTE(7)
    p2_inter = [(O.RECEIVE, P2.M, BUS_E0, [P2.LNP, P2.KP, P2.IN, P2.IN + 1, P2.IN + 2, P2.IN + 3]),
                (O.RECEIVE, P2.M, BUS_E1, [P2.LNP, P2.KP, P2.IN + 4, P2.IN + 5, P2.IN + 6, P2.IN + 7]),
                (O.SEND, P2.END, BUS_R0, [P2.LNP, o7, o7 + 1, o7 + 2, o7 + 3]),
                (O.SEND, P2.END, BUS_R1, [P2.LNP, o7 + 4, o7 + 5, o7 + 6, o7 + 7])]
    if T:
        p2_inter += [(O.SEND, P2.TRS, BUS_R0, [P2.LNP, P2.IN, P2.IN + 1, P2.IN + 2, P2.IN + 3]),
                     (O.SEND, P2.TRS, BUS_R1, [P2.LNP, P2.IN + 4, P2.IN + 5, P2.IN + 6, P2.IN + 7]),
                     (O.SEND, P2.TRS, BUS_B, [P2.LNP, o7 + 7, o7 + 6, o7 + 5, o7 + 4])]
    if QM:
        p2_inter += [(O.SEND, P2.QP, BUS_S0, [P2.LNP, o7 + 7, o7 + 6, o7 + 5, o7 + 4]), (O.SEND, P2.QP, BUS_S1, [P2.LNP, o7 + 3, o7 + 2, o7 + 1, o7])]
    p2_tab = O.interaction_table(p2_inter)
    fri_inter = [(O.SEND, ACTIVE, BUS_E0, [LN, K2, E0, E0 + 1, E0 + 2, E0 + 3]), (O.SEND, ACTIVE, BUS_E1, [LN, K2, E1, E1 + 1, E1 + 2, E1 + 3]),
                 (O.SEND, L_WIRED, BUS_Q, [IDX, OWN, OWN + 1, OWN + 2, OWN + 3])]
    if T:
        fri_inter.append((O.RECEIVE, ACTIVE, BUS_BF, [LN, BETA, BETA + 1, BETA + 2, BETA + 3]))
    fri_tab = O.interaction_table(fri_inter)
    q_tab = O.interaction_table([(O.RECEIVE, 5, BUS_Q, [QUERIES_PRE, 1, 2, 3, 4]), (O.RECEIVE, 5, BUS_I, [0, QUERIES_PRE])] if QM else [(O.RECEIVE, 5, BUS_Q, [0, 1, 2, 3, 4])])
    r_inter = [(O.RECEIVE, RP, BUS_R0, [0, 1, 2, 3, 4]), (O.RECEIVE, RP, BUS_R1, [0, 5, 6, 7, 8])]
    if T:
        r_inter += [(O.RECEIVE, 9, BUS_B, [0, RP + 1, RP + 2, RP + 3, RP + 4]), (O.SEND, RP + 5, BUS_BF, [0, RP + 1, RP + 2, RP + 3, RP + 4])]
    r_tab = O.interaction_table(r_inter)

    def table_prog(pre_width, main_width=4):
        return O.air_program(pre_width + main_width, NP, [(O.SEL_FIRST, [(1, [V(pre_width + main_width - 1)])])])
    pub = (list(final) + [int(v) for v in capacity]) if T else ([c for b in betas for c in b] + list(final))
    if QM:
        M0 = S_PRE
        s_tab = O.interaction_table([(O.RECEIVE, S_ROW, BUS_S0, [S_C] + [M0 + S_W + j for j in range(4)]), (O.RECEIVE, S_ROW, BUS_S1, [S_C] + [M0 + S_W + j for j in range(4, 8)])]
                                    + [(O.SEND, S_ACT + j, BUS_I, [S_KQ + j, M0 + S_IDX + j]) for j in range(8)])
        return ([p2_trace, fri_trace, qmain, rmain, smain], [None, None, qpre, rpre, spre],
                [P2.program(fri_layers=True, n_public=NP, transcript=4, queries=0), program(R, wired=True, transcript=True),
                 table_prog(QUERIES_PRE), table_prog(RP, ROOTS_MAIN_T), samples_program(R, Q, query_phase[1], NP)],
                [p2_tab, fri_tab, q_tab, r_tab, s_tab], pub)
    return ([p2_trace, fri_trace, np.zeros((1 << lr_q, 4), dtype=np.uint32), rmain], [None, None, qpre, rpre],
            [P2.program(fri_layers=True, n_public=NP, transcript=4 if T else None), program(R, wired=True, transcript=T),
             table_prog(QUERIES_PRE), table_prog(RP, ROOTS_MAIN_T if T else 4)],
            [p2_tab, fri_tab, q_tab, r_tab], pub)
