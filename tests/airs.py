"""Constraint programs (the AIR as data) used by the tests, and host-side trace generators for them."""
import numpy as np

import oracle_lib as O

P = O.P
V = O.air_var


def fibonacci_program():
    """4 columns (a, b, c, s): a' = b, b' = a + b on transitions; c = a b s and s (s - 1) = 0 on every row;
    a, b of the first row and b of the last row are public values 0, 1, 2"""
    return O.air_program(4, 3, [
        (O.SEL_TRANSITION, [(1, [V(0, True)]), (P - 1, [V(1)])]),
        (O.SEL_TRANSITION, [(1, [V(1, True)]), (P - 1, [V(0)]), (P - 1, [V(1)])]),
        (O.SEL_ALL, [(1, [V(2)]), (P - 1, [V(0), V(1), V(3)])]),
        (O.SEL_ALL, [(1, [V(3), V(3)]), (P - 1, [V(3)])]),
        (O.SEL_FIRST, [(1, [V(0)]), (P - 1, [V(0, public=True)])]),
        (O.SEL_FIRST, [(1, [V(1)]), (P - 1, [V(1, public=True)])]),
        (O.SEL_LAST, [(1, [V(1)]), (P - 1, [V(2, public=True)])]),
    ])


def fibonacci_trace(log_n, a0, b0):
    n = 1 << log_n
    t = np.zeros((n, 4), dtype=np.uint64)
    a, b = a0 % P, b0 % P
    for i in range(n):
        s = i & 1
        t[i] = [a, b, a * b * s % P, s]
        a, b = b, (a + b) % P
    t = t.astype(np.uint32)
    return t, [a0 % P, b0 % P, int(t[-1, 1])]


def counter_program(width):
    """`width` columns (multiple of 4): column 0 counts rows from public value 0 in steps of public value 1, every other column
    j holds (column 0)^2 * j + column (j - 1) -- a wide degree-3 AIR with a running dependency between columns"""
    cons = [(O.SEL_FIRST, [(1, [V(0)]), (P - 1, [V(0, public=True)])]),
            (O.SEL_TRANSITION, [(1, [V(0, True)]), (P - 1, [V(0)]), (P - 1, [V(1, public=True)])])]
    for j in range(1, width):
        cons.append((O.SEL_ALL, [(1, [V(j)]), (P - j, [V(0), V(0)]), (P - 1, [V(j - 1)])]))
    return O.air_program(width, 2, cons)


def counter_trace(log_n, width, start, step):
    n = 1 << log_n
    t = np.zeros((n, width), dtype=np.uint64)
    x = (start + step * np.arange(n, dtype=np.uint64)) % P
    t[:, 0] = x
    for j in range(1, width):
        t[:, j] = (x * x % P * j + t[:, j - 1]) % P
    return t.astype(np.uint32), [start % P, step % P]


def quintic_program():
    """4 columns (x, y, z, w): y = x^5 on every row (degree 5), z = x y y w (degree 4), w (w - 1) = 0, x' = x + 1 on transitions,
    x of the first row = public value 0.  Degree 5: four quotient chunks (log_quotient_degree 2), needs log_blowup >= 2."""
    return O.air_program(4, 1, [
        (O.SEL_ALL, [(1, [V(1)]), (P - 1, [V(0)] * 5)]),
        (O.SEL_TRANSITION, [(1, [V(0, True)]), (P - 1, [V(0)]), (P - 1, [])]),
        (O.SEL_ALL, [(1, [V(2)]), (P - 1, [V(0), V(1), V(1), V(3)])]),
        (O.SEL_ALL, [(1, [V(3), V(3)]), (P - 1, [V(3)])]),
        (O.SEL_FIRST, [(1, [V(0)]), (P - 1, [V(0, public=True)])]),
    ])


def quintic_trace(log_n, x0):
    n = 1 << log_n
    t = np.zeros((n, 4), dtype=np.uint64)
    for i in range(n):
        x = (x0 + i) % P
        y = pow(x, 5, P)
        w = (i >> 1) & 1
        t[i] = [x, y, x * y % P * y % P * w % P, w]
    return t.astype(np.uint32), [x0 % P]


def random_program_and_trace(seed, log_n, width, max_degree):
    """A pseudo-random AIR that its own trace satisfies: column 0 is a counter (first-row and transition constraints against public
    values), every later column j is DEFINED by a random polynomial of total degree <= max_degree in earlier columns of the same row
    and (for some terms) of the NEXT row's column 0 -- one `every row` constraint per column --, plus a last-row constraint on
    column 0.  Returns (program, trace, public values)."""
    rng = np.random.default_rng(seed)
    n = 1 << log_n
    start, step = int(rng.integers(0, P)), int(rng.integers(1, P))
    x = (start + step * np.arange(n + 1, dtype=object)) % P          # one extra row: the "next" value of the last row wraps to row 0 below
    t = np.zeros((n, width), dtype=object)
    t[:, 0] = x[:n]
    nxt0 = np.array([int(t[(i + 1) % n, 0]) for i in range(n)], dtype=object)
    cons = [(O.SEL_FIRST, [(1, [V(0)]), (P - 1, [V(0, public=True)])]),
            (O.SEL_TRANSITION, [(1, [V(0, True)]), (P - 1, [V(0)]), (P - 1, [V(1, public=True)])]),
            (O.SEL_LAST, [(1, [V(0)]), (P - 1, [V(2, public=True)])])]
    for j in range(1, width):
        terms, val = [], np.zeros(n, dtype=object)
        for _ in range(int(rng.integers(1, 4))):
            d = int(rng.integers(0, max_degree + 1))
            coeff = int(rng.integers(1, P))
            vs, prod = [], np.full(n, coeff, dtype=object)
            for _k in range(d):
                if rng.random() < 0.2:
                    vs.append(V(0, True)); prod = prod * nxt0 % P
                else:
                    c = int(rng.integers(0, j)); vs.append(V(c)); prod = prod * t[:, c] % P
            terms.append(((P - coeff) % P, vs))
            val = (val + prod) % P
        t[:, j] = val
        cons.append((O.SEL_ALL, [(1, [V(j)])] + terms))
    pub = [start, step, int(t[n - 1, 0])]
    return O.air_program(width, 3, cons), t.astype(np.uint32), pub
