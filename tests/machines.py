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
