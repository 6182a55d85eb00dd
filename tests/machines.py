"""Small machines for the tests: tables with their own constraint programs that look each other up (interaction tables)."""
import numpy as np

import oracle_lib as O

P = O.P
V = O.air_var
BUS_RANGE, BUS_PAIR = 7, 9


def range_machine(log_table=6, log_users=7, seed=1):
    """a range-check machine over three public values (p0, p1, p2 unused by most):
         USER   2^log_users x 4: (x, y, x y, 0), x and y < 2^log_table; sends (1, [x]) and (1, [y]) on the range bus, and on every row
                (1, [x, y]) on the pair bus;
         PICK   2^(log_users+1) x 4: (a, b, m, 0): a superset of USER's (x, y) rows; receives (m, [a, b]) on the pair bus, m in {0, 1};
         TABLE  2^log_table x 4: (v, m, 0, 0), v = row index (first row 0, step 1), receives (m, [v]) on the range bus.
       -> (traces, programs, tables, public values), tallest first"""
    rng = np.random.default_rng(seed)
    nu, nt = 1 << log_users, 1 << log_table
    x, y = rng.integers(0, nt, nu), rng.integers(0, nt, nu)
    user = np.zeros((nu, 4), dtype=np.uint64)
    user[:, 0], user[:, 1], user[:, 2] = x, y, x * y % P
    user_prog = O.air_program(4, 3, [(O.SEL_ALL, [(1, [V(2)]), (P - 1, [V(0), V(1)])])])
    user_tab = O.interaction_table([(O.SEND, None, BUS_RANGE, [0]), (O.SEND, None, BUS_RANGE, [1]), (O.SEND, None, BUS_PAIR, [0, 1])])
    # PICK: USER's pairs at pseudo-random rows of a table twice as tall, the other rows hold other pairs with multiplicity 0
    npk = 2 * nu
    pick = np.zeros((npk, 4), dtype=np.uint64)
    pick[:, 0], pick[:, 1] = rng.integers(0, P, npk), rng.integers(0, P, npk)
    rows = rng.permutation(npk)[:nu]
    pick[rows, 0], pick[rows, 1], pick[rows, 2] = x, y, 1
    pick_prog = O.air_program(4, 3, [(O.SEL_ALL, [(1, [V(2), V(2)]), (P - 1, [V(2)])])])          # m is a bit
    pick_tab = O.interaction_table([(O.RECEIVE, 2, BUS_PAIR, [0, 1])])
    table = np.zeros((nt, 4), dtype=np.uint64)
    table[:, 0] = np.arange(nt)
    table[:, 1] = np.bincount(np.concatenate([x, y]), minlength=nt)
    table_prog = O.air_program(4, 3, [(O.SEL_FIRST, [(1, [V(0)])]),
                                      (O.SEL_TRANSITION, [(1, [V(0, True)]), (P - 1, [V(0)]), (P - 1, [])])])
    table_tab = O.interaction_table([(O.RECEIVE, 1, BUS_RANGE, [0])])
    traces = [pick.astype(np.uint32), user.astype(np.uint32), table.astype(np.uint32)]
    return traces, [pick_prog, user_prog, table_prog], [pick_tab, user_tab, table_tab], [11, 22, 33]


def random_machine(seed):
    """A pseudo-random machine: 2..5 tables of random heights and widths; every bus carries tuples of a random length 1..8 from
    one SENDER table (every row sends one tuple of a pool, some rows send a second one with a 0 / 1 multiplicity column) to one
    RECEIVER table (one row per pool tuple, with the total multiplicity in a column; its other rows hold junk with multiplicity 0).
    Each table's program: its last column is a bit.  -> (traces, programs, tables, public values), tallest first"""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 6))
    heights = sorted((int(h) for h in rng.integers(5, 10, n)), reverse=True)
    while max(heights.count(h) for h in heights) > 4:
        heights = sorted((int(h) for h in rng.integers(5, 10, n)), reverse=True)
    widths = [4 * int(rng.integers(6, 12)) for _ in range(n)]                   # room for several tuples of up to 8 values
    traces = [rng.integers(0, P, (1 << h, w)).astype(np.uint64) for h, w in zip(heights, widths)]
    for t in traces:
        t[:, -1] = rng.integers(0, 2, t.shape[0])
    inter = [[] for _ in range(n)]
    used = [0] * n                                                               # next free column per table
    for bus in range(int(rng.integers(1, 4))):
        a, b = (int(x) for x in rng.choice(n, 2, replace=False))
        nv = int(rng.integers(1, 9))
        if used[a] + nv + 1 >= widths[a] - 1 or used[b] + nv + 1 >= widths[b] - 1:
            continue
        ra, rb = traces[a].shape[0], traces[b].shape[0]
        pool_n = int(rng.integers(1, min(rb, 40) + 1))
        pool = rng.integers(0, P, (pool_n, nv)).astype(np.uint64)
        pick = rng.integers(0, pool_n, ra)
        ca, cb = used[a], used[b]
        traces[a][:, ca:ca + nv] = pool[pick]
        flag = rng.integers(0, 2, ra).astype(np.uint64)                         # a multiplicity column on the sending side too
        traces[a][:, ca + nv] = flag
        counts = np.bincount(pick, weights=flag.astype(np.float64), minlength=pool_n).astype(np.uint64)
        inter[a].append((O.SEND, ca + nv, 100 + bus, list(range(ca, ca + nv))))
        if rng.random() < 0.5:                                                   # and once more with the constant multiplicity
            inter[a].append((O.SEND, None, 100 + bus, list(range(ca, ca + nv))))
            counts = counts + np.bincount(pick, minlength=pool_n).astype(np.uint64)
        rows = rng.permutation(rb)[:pool_n]
        traces[b][:, cb + nv] = 0
        traces[b][rows, cb:cb + nv] = pool
        traces[b][rows, cb + nv] = counts % P
        inter[b].append((O.RECEIVE, cb + nv, 100 + bus, list(range(cb, cb + nv))))
        used[a], used[b] = ca + nv + 1, cb + nv + 1
    progs = [O.air_program(w, 2, [(O.SEL_ALL, [(1, [V(w - 1), V(w - 1)]), (P - 1, [V(w - 1)])])]) for w in widths]
    tables = [O.interaction_table(it) if it else None for it in inter]
    return [t.astype(np.uint32) for t in traces], progs, tables, [int(rng.integers(0, P)), int(rng.integers(0, P))]


BUS_BYTE = 21


def byte_machine(log_users=7, bits=4, seed=1):
    """A machine with PREPROCESSED columns, as SP1's byte chip is: (the combined row of a chip is [preprocessed | main])
         USER   2^log_users rows, no preprocessed columns, main (a, b, x, o): claims x = a XOR b and o = a OR b for `bits`-bit a, b by
                sending (1, [a, b, x, o]) on the byte bus; its program: x + 2 (a AND b) = a + b with a AND b = a + b - o, i.e.
                x + 2 (a + b - o) = a + b -- a relation the table's rows satisfy too, so this constraint alone proves nothing new; it is
                there to have a program next to the lookup;
         TABLE  2^(2 bits) rows, preprocessed (a, b, a XOR b, a OR b) for every pair, main (m, 0, 0, 0): receives (m, [pre 0..3]);
                its program: first row's preprocessed a is 0 (a harmless identity: the table's contents are fixed by the KEY, not by
                constraints).
       -> (traces, preprocessed traces, programs, tables, public values), tallest first"""
    rng = np.random.default_rng(seed)
    nu, nb = 1 << log_users, 1 << bits
    a, b = rng.integers(0, nb, nu), rng.integers(0, nb, nu)
    user = np.stack([a, b, a ^ b, a | b], axis=1).astype(np.uint32)
    user_prog = O.air_program(4, 1, [(O.SEL_ALL, [(1, [V(2)]), (1, [V(0)]), (1, [V(1)]), (P - 2, [V(3)])])])
    user_tab = O.interaction_table([(O.SEND, None, BUS_BYTE, [0, 1, 2, 3])])
    aa, bb = np.divmod(np.arange(nb * nb), nb)
    pre = np.stack([aa, bb, aa ^ bb, aa | bb], axis=1).astype(np.uint32)
    main = np.zeros((nb * nb, 4), dtype=np.uint32)
    main[:, 0] = np.bincount(a * nb + b, minlength=nb * nb)
    table_prog = O.air_program(8, 1, [(O.SEL_FIRST, [(1, [V(0)])])])
    table_tab = O.interaction_table([(O.RECEIVE, 4, BUS_BYTE, [0, 1, 2, 3])])
    chips = [(log_users, user, None, user_prog, user_tab), (2 * bits, main, pre, table_prog, table_tab)]
    chips.sort(key=lambda c: -c[0])
    return [c[1] for c in chips], [c[2] for c in chips], [c[3] for c in chips], [c[4] for c in chips], [5]


def random_keyed_machine(seed):
    """random_machine(seed) with the leading columns of some tables declared PREPROCESSED: the combined row [pre | main] is the original
    row, so programs and interaction tables stay as they are.  -> (main traces, preprocessed traces or None, programs, tables, public values)"""
    traces, progs, tables, pub = random_machine(seed)
    rng = np.random.default_rng(seed + 77)
    pre, main = [], []
    for t in traces:
        pw = 4 * int(rng.integers(0, t.shape[1] // 4))                       # 0 .. width - 4
        pre.append(np.ascontiguousarray(t[:, :pw]) if pw else None)
        main.append(np.ascontiguousarray(t[:, pw:]))
    if all(p is None for p in pre):
        pre[0], main[0] = np.ascontiguousarray(traces[0][:, :4]), np.ascontiguousarray(traces[0][:, 4:])
    return main, pre, progs, tables, pub


def sha256_machine(message):
    """the SHA-256 guest as a keyed machine, rebuilt independently of zkhip_prove_sha256_machine: the compression chip (tests/sha256_air.py)
    sends the OUT limbs of d and h to a 2^16-row range table whose values are preprocessed and whose multiplicities are main columns.
    -> (traces, preprocessed traces, programs, tables, public values), tallest first"""
    import sha256_air as S
    sha_t, pub = S.trace(S.pad(message))
    sent = [S.OUT + 6, S.OUT + 7, S.OUT + 14, S.OUT + 15]
    sha_tab = O.interaction_table([(O.SEND, None, 16, [c]) for c in sent])
    values = np.zeros((1 << 16, 4), dtype=np.uint32)
    values[:, 0] = np.arange(1 << 16)
    main = values.copy()
    main[:, 1] = np.bincount(sha_t[:, sent].ravel(), minlength=1 << 16)
    table_prog = O.air_program(8, S.N_PUBLIC, [(O.SEL_FIRST, [(1, [V(0)])])])
    table_tab = O.interaction_table([(O.RECEIVE, 5, 16, [0])])
    chips = [(sha_t, None, S.program(), sha_tab), (main, values, table_prog, table_tab)]
    if sha_t.shape[0] <= 1 << 16:
        chips.reverse()
    return [c[0] for c in chips], [c[1] for c in chips], [c[2] for c in chips], [c[3] for c in chips], list(pub)


BUS_SP1 = 300


def sp1_shaped_spec(spec):
    """spec: [(log_n, width, pairs, partner)] tallest first; partner = index of the chip of EQUAL height whose sender groups this chip's receiver
    groups hold (-1: the pairs are in-table).  -> per chip the interaction list: pair q sends the (a, b) of column group 2q and receives at group
    2q + 1 -- on a bus of the chip's own for an in-table pair, and for a cross pair the sender's bus (chip c sends on BUS_SP1 + 16 c + q, its
    partner receives there), so that the two tables' cumulative sums cancel instead of vanishing one by one (SP1's cross-table lookups)."""
    inter = []
    for c, (ln, w, pairs, partner) in enumerate(spec):
        assert w % 4 == 0 and 8 * pairs <= w and pairs <= 16
        it = []
        for q in range(pairs):
            it.append((O.SEND, None, BUS_SP1 + 16 * c + q, [8 * q, 8 * q + 1]))
            src = c if partner < 0 else partner
            it.append((O.RECEIVE, None, BUS_SP1 + 16 * src + q, [8 * q + 4, 8 * q + 5]))
        if partner >= 0:
            assert spec[partner][0] == ln and spec[partner][3] == c and spec[partner][2] == pairs, "cross pairs come in two tables of one height that look each other up"
        inter.append(it)
    return inter


def sp1_shaped_machine(spec, seed=1, shard=0, pre=(), n_public=3, key_shard=9999, key_seed=None):
    """SP1's shard structure as a KEYED machine (proof version 11): chips of mixed heights, each under the synthetic AIR (as a constraint program)
    on traces of orc_gen_trace_logup / _cross, in-table LogUp pairs and a cross-table bus between two chips of one height, and preprocessed
    leading columns on the chips listed in `pre` ((chip, columns), ...).  The chip streams are seed + 100 shard + chip; the preprocessed
    columns come from the stream of shard `key_shard`, so every shard is proven against ONE key (zktls_amd.device.Sp1ShapedShard does the same
    on the device).  key_seed: the preprocessed columns' stream when it is not the shards' (the host mirror derives it from the guest program alone: setup(elf),
    sp1.rs:113).  -> (main traces, preprocessed traces or None, programs, tables, public values)"""
    inter = sp1_shaped_spec(spec)
    pw = dict(pre)

    def gen(c, sh):
        ln, w, pairs, partner = spec[c]
        sd = key_seed if (sh == key_shard and key_seed is not None) else seed
        if partner < 0:
            return O.gen_trace_logup(sd, 100 * sh + c, ln, w, pairs)
        return O.gen_trace_logup_cross(sd, 100 * sh + c, 100 * sh + partner, ln, w, spec[partner][1], pairs)
    traces = [gen(c, shard) for c in range(len(spec))]
    progs = [O.air_synthetic(w, n_public) for _, w, _, _ in spec]
    tabs = [O.interaction_table(it) if it else None for it in inter]
    pres = [np.ascontiguousarray(gen(c, key_shard)[:, :pw[c]]) if pw.get(c) else None for c in range(len(spec))]
    mains = [np.ascontiguousarray(t[:, pw.get(c, 0):]) for c, t in enumerate(traces)]
    return mains, pres, progs, tabs, [(11 * (i + 1)) % P for i in range(n_public - 1)] + [shard]
