"""First-principles pure-Python definitions (ints and pow(), no shared code with oracle/
or the HIP path).  Small cases only: used to pin the C oracle from a second, independent
restatement, as SURVEY.md section 8(c) prescribes for a path with no reference vectors.
"""
import json
import os

P = 2**31 - 2**27 + 1
GEN = 31
W = 11  # x^4 = 11

_HERE = os.path.dirname(os.path.abspath(__file__))
PARAMS = json.load(open(os.path.join(_HERE, "golden", "poseidon2_params.json")))


def two_adic_generator(bits):
    return pow(pow(GEN, (P - 1) >> 27, P), 1 << (27 - bits), P)


def dft(vec, inverse=False):
    n = len(vec)
    bits = n.bit_length() - 1
    w = two_adic_generator(bits)
    if inverse:
        w = pow(w, -1, P)
    out = [sum(vec[j] * pow(w, j * k, P) for j in range(n)) % P for k in range(n)]
    if inverse:
        ninv = pow(n, -1, P)
        out = [x * ninv % P for x in out]
    return out


def bitrev(i, bits):
    return int(format(i, "0%db" % bits)[::-1], 2) if bits else 0


def coset_lde_column(vec, log_blowup, shift):
    """evaluate the interpolant of vec (on the subgroup) at shift*w^i, bit-reversed order"""
    n = len(vec)
    coeffs = dft(vec, inverse=True)
    m = n << log_blowup
    bits = m.bit_length() - 1
    w = two_adic_generator(bits)
    out = [0] * m
    for i in range(m):
        x = shift * pow(w, i, P) % P
        out[bitrev(i, bits)] = sum(c * pow(x, j, P) for j, c in enumerate(coeffs)) % P
    return out


# ---- extension field
def ext_mul(a, b):
    t = [0] * 7
    for i in range(4):
        for j in range(4):
            t[i + j] = (t[i + j] + a[i] * b[j]) % P
    return [(t[i] + (W * t[i + 4] if i + 4 < 7 else 0)) % P for i in range(4)]


def ext_pow(a, e):
    r = [1, 0, 0, 0]
    while e:
        if e & 1:
            r = ext_mul(r, a)
        a = ext_mul(a, a)
        e >>= 1
    return r


def ext_inv(a):
    return ext_pow(a, P**4 - 2)


# ---- Poseidon2 from the parameter file, matrices written out explicitly
def _mat_external():
    m4 = PARAMS["m4"]
    M = [[0] * 16 for _ in range(16)]
    for bi in range(4):
        for bj in range(4):
            f = 2 if bi == bj else 1
            for i in range(4):
                for j in range(4):
                    M[4 * bi + i][4 * bj + j] = f * m4[i][j]
    return M


def _mat_internal():
    d = PARAMS["internal_diag"]
    return [[(1 + (d[i] if i == j else 0)) % P for j in range(16)] for i in range(16)]


ME, MI = _mat_external(), _mat_internal()


def _matvec(M, v):
    return [sum(M[i][j] * v[j] for j in range(16)) % P for i in range(16)]


def poseidon2(state):
    s = [x % P for x in state]
    rc_e, rc_i = PARAMS["external_rc"], PARAMS["internal_rc"]
    s = _matvec(ME, s)
    for r in range(4):
        s = [pow((s[i] + rc_e[r][i]) % P, 7, P) for i in range(16)]
        s = _matvec(ME, s)
    for r in range(13):
        s[0] = pow((s[0] + rc_i[r]) % P, 7, P)
        s = _matvec(MI, s)
    for r in range(4, 8):
        s = [pow((s[i] + rc_e[r][i]) % P, 7, P) for i in range(16)]
        s = _matvec(ME, s)
    return s


def sponge_hash(vals):
    st = [0] * 16
    pos = 0
    for v in vals:
        st[pos] = v % P
        pos += 1
        if pos == 8:
            st = poseidon2(st)
            pos = 0
    if pos:
        st = poseidon2(st)
    return st[:8]


def compress(l, r):
    return poseidon2(list(l) + list(r))[:8]


def mix64(z):
    M = (1 << 64) - 1
    z = (z + 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def synth_value(seed, index):
    M = (1 << 64) - 1
    return mix64((seed + index * 0x9E3779B97F4A7C15) & M) % P


# ---- width 24 (RISC Zero's shape), same construction from its own parameter file
PARAMS24 = json.load(open(os.path.join(_HERE, "golden", "poseidon2_24_params.json")))


def poseidon2_24(state):
    m4 = PARAMS24["m4"]
    t = 24
    ME24 = [[(2 if i // 4 == j // 4 else 1) * m4[i % 4][j % 4] for j in range(t)] for i in range(t)]
    d = PARAMS24["internal_diag"]
    MI24 = [[(1 + (d[i] if i == j else 0)) % P for j in range(t)] for i in range(t)]
    mv = lambda M, v: [sum(M[i][j] * v[j] for j in range(t)) % P for i in range(t)]
    rc_e, rc_i = PARAMS24["external_rc"], PARAMS24["internal_rc"]
    s = mv(ME24, [x % P for x in state])
    for r in range(4):
        s = mv(ME24, [pow((s[i] + rc_e[r][i]) % P, 7, P) for i in range(t)])
    for r in range(21):
        s[0] = pow((s[0] + rc_i[r]) % P, 7, P)
        s = mv(MI24, s)
    for r in range(4, 8):
        s = mv(ME24, [pow((s[i] + rc_e[r][i]) % P, 7, P) for i in range(t)])
    return s


def sponge24(vals):
    st = [0] * 24
    pos = 0
    for v in vals:
        st[pos] = v % P
        pos += 1
        if pos == 16:
            st = poseidon2_24(st)
            pos = 0
    if pos:
        st = poseidon2_24(st)
    return st[:8]


def compress24(l, r):
    return poseidon2_24(list(l) + list(r) + [0] * 8)[:8]
