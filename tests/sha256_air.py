"""SHA-256 compression as a constraint program (test-side restatement; the product's generator is csrc/sha256_chip.cpp).

One row per round, 64 rows per 64-byte block, blocks one after the other; 640 columns (612 in use), every constraint of degree <= 3
(log_quotient_degree 1).  What a proof says (round 5): "I know a message of exactly L bytes whose SHA-256 digest is the 16 public
16-bit limbs" -- L public, the FIPS 180-4 padding constrained in-circuit through 75 more public values a verifier derives from L
(padding_publics).  The trace height is a power of two, the block count of a message is not: blocks after the message are INACTIVE
(ACT = 0) and pass the chaining value through unchanged; a block counter pins where ACT drops.

Columns (bit i of a word = column base + i, least significant first; a limb pair = low 16 bits, high 16 bits):
  SEL  64   one-hot round selector s_t
  A B C E F G  32 bits each: working variables a, b, c, e, f, g BEFORE round t
  D HV  2 limbs each: working variables d, h
  S1 CH S0 MJ  32 bits each: Sigma1(e), Ch(e,f,g), Sigma0(a), Maj(a,b,c)
  HC   8 x 2 limbs: the chaining value the block started from
  OUT  8 x 2 limbs: the working variables after the round, plus the chaining value in round 63 (mod 2^32)
  X0 X13  32 bits each: W_t and W_{t+13};  XL  14 x 2 limbs: W_{t+j}, j = 1..12, 14, 15
  SG0 SG1  32 bits each: sigma0(W_t), sigma1(W_{t+13})
  CY   28 carry bits
  ACT  1 while the block belongs to the message;  SKIP = s_63 (1 - ACT): round 63 of an inactive block
  CNT  active blocks left (this one included);  LASTB  the last active block;  L2  the block before it;  SB  the row whose W_t holds the 0x80 byte;
  Z0 = s_0 LASTB;  Z2 = s_0 L2
"""
import hashlib
import struct

import numpy as np

import oracle_lib as O

P = O.P
V = O.air_var

SEL, A, B, C, E, F, G = 0, 64, 96, 128, 160, 192, 224
D, HV = 256, 258
S1, CH, S0, MJ = 260, 292, 324, 356
HC, OUT = 388, 404
X0, X13 = 420, 452
XL = 484
SG0, SG1 = 512, 544
CY = 576
ACT, SKIP = 604, 605
CNT, LASTB, L2, SB, Z0, Z2 = 606, 607, 608, 609, 610, 611
WIDTH = 640          # 612 columns in use: the product pads to whole 32-column tiles (the LDE's fast passes)
# public values: 16 digest limbs (chained: + 16 limbs of the initial chaining value), then the padding's 75:
N_DIGEST = 16
PP_K, PP_FIN, PP_Z13, PP_BWL, PP_BW2, PP_KIND, PP_ZWL, PP_ZW2, PP_LEN, N_PAD = 0, 1, 2, 3, 19, 35, 39, 55, 71, 75
N_PUBLIC = N_DIGEST + N_PAD
N_PUBLIC_CHAINED = 2 * N_DIGEST + N_PAD
# carries: a (3 + 3), e (3 + 3), b c d f g h (1 + 1 each), schedule (2 + 2)
CY_A, CY_E, CY_W6, CY_SCHED = CY, CY + 6, CY + 12, CY + 24

IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


def _round_constants():
    """K_t = first 32 bits of the fractional part of the cube root of the t-th prime (FIPS 180-4, 4.2.2), from integers only"""
    ks, n = [], 2
    while len(ks) < 64:
        if all(n % q for q in range(2, int(n ** 0.5) + 1)):
            lo, hi = 0, 1 << 40                       # floor(cbrt(n) * 2^32) by bisection on integers
            while hi - lo > 1:
                mid = (lo + hi) // 2
                if mid ** 3 <= n << 96:
                    lo = mid
                else:
                    hi = mid
            ks.append(lo & 0xffffffff)
        n += 1
    return ks


K = _round_constants()
assert K[0] == 0x428a2f98 and K[63] == 0xc67178f2


def _xl(j):
    """first column of the limb pair of W_{t+j}, j in 1..12, 14, 15"""
    return XL + 2 * ({14: 12, 15: 13}[j] if j >= 14 else j - 1)


def _limb(kind, base, l, nxt=False, scale=1):
    """terms of `scale` x (limb l of a word); kind 'bits': 32 bit columns from base; kind 'limbs': a limb pair at base"""
    if kind == "bits":
        return [((scale << i) % P, [V(base + 16 * l + i, nxt)]) for i in range(16)]
    return [(scale % P, [V(base + l, nxt)])]


WORDS = [("bits", A), ("bits", B), ("bits", C), ("limbs", D), ("bits", E), ("bits", F), ("bits", G), ("limbs", HV)]     # a .. h


def _x(j):
    return ("bits", X0) if j == 0 else (("bits", X13) if j == 13 else ("limbs", _xl(j)))


def _neg(terms):
    return [((P - c) % P, vs) for c, vs in terms]


def _xor3(x, y, z):
    return [(1, [x]), (1, [y]), (1, [z]), (P - 2, [x, y]), (P - 2, [y, z]), (P - 2, [x, z]), (4, [x, y, z])]


def _xor2(x, y):
    return [(1, [x]), (1, [y]), (P - 2, [x, y])]


def program(chained=False):
    """chained: the chaining value of the first row is PUBLIC too (32 public values: final limbs, then initial limbs) instead of the
    standard initial value -- a shard of a longer message"""
    cons = []
    s63 = V(SEL + 63)
    # round selector: s_0 = 1 on the first row, cyclic shift on transitions
    cons.append((O.SEL_FIRST, [(1, [V(SEL)]), (P - 1, [])]))
    for t in range(1, 64):
        cons.append((O.SEL_FIRST, [(1, [V(SEL + t)])]))
    for t in range(64):
        cons.append((O.SEL_TRANSITION, [(1, [V(SEL + (t + 1) % 64, True)]), (P - 1, [V(SEL + t)])]))
    # bits are bits
    for base in (A, B, C, E, F, G, X0, X13):
        for i in range(32):
            cons.append((O.SEL_ALL, [(1, [V(base + i), V(base + i)]), (P - 1, [V(base + i)])]))
    for i in range(28):
        cons.append((O.SEL_ALL, [(1, [V(CY + i), V(CY + i)]), (P - 1, [V(CY + i)])]))
    # the bitwise functions of the round
    for i in range(32):
        cons.append((O.SEL_ALL, [(1, [V(S1 + i)])] + _neg(_xor3(V(E + (i + 6) % 32), V(E + (i + 11) % 32), V(E + (i + 25) % 32)))))
    for i in range(32):
        cons.append((O.SEL_ALL, [(1, [V(CH + i)]), (P - 1, [V(G + i)]), (P - 1, [V(E + i), V(F + i)]), (1, [V(E + i), V(G + i)])]))
    for i in range(32):
        cons.append((O.SEL_ALL, [(1, [V(S0 + i)])] + _neg(_xor3(V(A + (i + 2) % 32), V(A + (i + 13) % 32), V(A + (i + 22) % 32)))))
    for i in range(32):
        a, b, c = V(A + i), V(B + i), V(C + i)
        cons.append((O.SEL_ALL, [(1, [V(MJ + i)]), (P - 1, [a, b]), (P - 1, [a, c]), (P - 1, [b, c]), (2, [a, b, c])]))
    # the message schedule's bitwise functions: sigma0 = rotr 7 ^ rotr 18 ^ shr 3 of W_t, sigma1 = rotr 17 ^ rotr 19 ^ shr 10 of W_{t+13}
    for out, src, r1, r2, sh in ((SG0, X0, 7, 18, 3), (SG1, X13, 17, 19, 10)):
        for i in range(32):
            x, y = V(src + (i + r1) % 32), V(src + (i + r2) % 32)
            body = _xor3(x, y, V(src + i + sh)) if i + sh < 32 else _xor2(x, y)
            cons.append((O.SEL_ALL, [(1, [V(out + i)])] + _neg(body)))
    # the round: OUT = new working variables (+ chaining value in round 63) mod 2^32, limb by limb with carries
    for l in range(2):
        k_l = [(((K[t] >> (16 * l)) & 0xffff), [V(SEL + t)]) for t in range(64)]
        t1 = _limb("limbs", HV, l) + _limb("bits", S1, l) + _limb("bits", CH, l) + k_l + _limb("bits", X0, l)
        t2 = _limb("bits", S0, l) + _limb("bits", MJ, l)
        for w, (src, cy, ncy) in enumerate([(t1 + t2, CY_A, 3), (WORDS[0:1], CY_W6, 1), (WORDS[1:2], CY_W6 + 2, 1), (WORDS[2:3], CY_W6 + 4, 1),
                                            (_limb("limbs", D, l) + t1, CY_E, 3), (WORDS[4:5], CY_W6 + 6, 1), (WORDS[5:6], CY_W6 + 8, 1),
                                            (WORDS[6:7], CY_W6 + 10, 1)]):
            rhs = src if w in (0, 4) else _limb(src[0][0], src[0][1], l)
            terms = [(1, [V(OUT + 2 * w + l)])] + [((1 << (16 + k)) % P, [V(cy + ncy * l + k)]) for k in range(ncy)]
            terms += _neg(rhs) + [(c, [V(SKIP)] + vs) for c, vs in rhs] + [(P - 1, [s63, V(HC + 2 * w + l)])]
            if l == 1:
                terms += [((P - (1 << k)) % P, [V(cy + k)]) for k in range(ncy)]          # carry out of the low limb
            cons.append((O.SEL_ALL, terms))
    # activity: a bit, 1 on the first row, never back from 0 to 1 (so constant inside a block needs no separate rule: the
    # window / round rows of an inactive block are unconstrained garbage that SKIP drops); SKIP = s_63 (1 - ACT)
    cons.append((O.SEL_ALL, [(1, [V(ACT), V(ACT)]), (P - 1, [V(ACT)])]))
    cons.append((O.SEL_FIRST, [(1, [V(ACT)]), (P - 1, [])]))
    cons.append((O.SEL_TRANSITION, [(1, [V(ACT, True)]), (P - 1, [V(ACT, True), V(ACT)])]))
    cons.append((O.SEL_ALL, [(1, [V(SKIP)]), (P - 1, [s63]), (1, [s63, V(ACT)])]))
    # the next row starts from OUT
    for w, (kind, base) in enumerate(WORDS):
        for l in range(2):
            cons.append((O.SEL_TRANSITION, _limb(kind, base, l, True) + [(P - 1, [V(OUT + 2 * w + l)])]))
    # chaining value: equals the working variables where a block starts, constant inside a block
    for w, (kind, base) in enumerate(WORDS):
        for l in range(2):
            cons.append((O.SEL_ALL, [(1, [V(SEL), V(HC + 2 * w + l)])] + [((P - c) % P, [V(SEL)] + vs) for c, vs in _limb(kind, base, l)]))
    for i in range(16):
        h, hn = V(HC + i), V(HC + i, True)
        cons.append((O.SEL_TRANSITION, [(1, [hn]), (P - 1, [h]), (P - 1, [s63, hn]), (1, [s63, h])]))
    # first row: the IV; last row: the public digest
    for w, (kind, base) in enumerate(WORDS):
        for l in range(2):
            if chained:
                cons.append((O.SEL_FIRST, _limb(kind, base, l) + [(P - 1, [V(16 + 2 * w + l, public=True)])]))
            else:
                cons.append((O.SEL_FIRST, _limb(kind, base, l) + [((P - ((IV[w] >> (16 * l)) & 0xffff)) % P, [])]))
    for i in range(16):
        cons.append((O.SEL_LAST, [(1, [V(OUT + i)]), (P - 1, [V(i, public=True)])]))
    # message schedule window: shifts and the recurrence, both off in round 63 (the next block brings its own 16 words)
    def gated(terms):
        return terms + [((P - c) % P, [s63] + vs) for c, vs in terms]
    for j in range(15):
        (k0, b0), (k1, b1) = _x(j), _x(j + 1)
        for l in range(2):
            cons.append((O.SEL_TRANSITION, gated(_limb(k0, b0, l, True) + _neg(_limb(k1, b1, l)))))
    for l in range(2):
        terms = _limb("limbs", _xl(15), l, True) + [((1 << (16 + k)) % P, [V(CY_SCHED + 2 * l + k)]) for k in range(2)]
        terms += _neg(_limb("bits", SG1, l, True) + _limb("limbs", _xl(8), l, True) + _limb("bits", SG0, l, True) + _limb("bits", X0, l))
        if l == 1:
            terms += [((P - (1 << k)) % P, [V(CY_SCHED + k)]) for k in range(2)]
        cons.append((O.SEL_TRANSITION, gated(terms)))
    # ---- padding (FIPS 180-4 5.1.1), pinned through public values derived from the message length (padding_publics)
    PB = 2 * N_DIGEST if chained else N_DIGEST
    pv = lambda i: V(PB + i, public=True)
    act, actn, cnt, lastb, l2, sb, z0, z2, s0 = V(ACT), V(ACT, True), V(CNT), V(LASTB), V(L2), V(SB), V(Z0), V(Z2), V(SEL)
    # the block count: CNT starts at K, loses one where an active block ends, is zero on inactive rows, equals ACT on the last row
    cons.append((O.SEL_FIRST, [(1, [cnt]), (P - 1, [pv(PP_K)])]))
    cons.append((O.SEL_TRANSITION, [(1, [V(CNT, True)]), (P - 1, [cnt]), (1, [s63, act])]))
    cons.append((O.SEL_ALL, [(1, [cnt]), (P - 1, [act, cnt])]))
    cons.append((O.SEL_LAST, [(1, [cnt]), (P - 1, [act])]))
    # LASTB, L2: bits, constant inside a block; ACT drops exactly behind the LASTB block; the block before it carries L2
    for f in (LASTB, L2):
        cons.append((O.SEL_ALL, [(1, [V(f), V(f)]), (P - 1, [V(f)])]))
        cons.append((O.SEL_TRANSITION, [(1, [V(f, True)]), (P - 1, [V(f)]), (P - 1, [s63, V(f, True)]), (1, [s63, V(f)])]))
    cons.append((O.SEL_TRANSITION, [(1, [s63, act]), (P - 1, [s63, actn]), (P - 1, [s63, lastb])]))
    cons.append((O.SEL_LAST, [(1, [lastb]), (P - 1, [act])]))
    cons.append((O.SEL_TRANSITION, [(1, [s63, V(LASTB, True)]), (P - 1, [s63, l2])]))
    cons.append((O.SEL_LAST, [(1, [l2])]))
    cons.append((O.SEL_ALL, [(1, [z0]), (P - 1, [s0, lastb])]))
    cons.append((O.SEL_ALL, [(1, [z2]), (P - 1, [s0, l2])]))
    # SB marks row t = j of the block with the 0x80 byte, j = the word's index: there X0 holds the word's bits
    terms = [(1, [sb])]
    for j in range(16):
        terms += [(P - 1, [pv(PP_BWL + j), lastb, V(SEL + j)]), (P - 1, [pv(PP_BW2 + j), l2, V(SEL + j)])]
    cons.append((O.SEL_ALL, terms))
    for i in range(32):
        terms = []
        for c in range(4):
            if i <= 31 - 8 * c:
                terms.append((1, [pv(PP_KIND + c), sb, V(X0 + i)]))
            if i == 31 - 8 * c:
                terms.append((P - 1, [pv(PP_KIND + c), sb]))
        cons.append((O.SEL_ALL, terms))
    # zero words and the length field, at row s_0 of their block (all sixteen words of the block are in the window there)
    for j in range(16):
        kind, base = _x(j)
        for l in range(2):
            lj = _limb(kind, base, l)
            terms = [(c, [pv(PP_ZWL + j), z0] + vs) for c, vs in lj] + [(c, [pv(PP_ZW2 + j), z2] + vs) for c, vs in lj]
            if j <= 13:
                terms += [(c, [pv(PP_Z13), z0] + vs) for c, vs in lj]
            else:
                terms += [(c, [pv(PP_FIN), z0] + vs) for c, vs in lj]
                terms.append((P - 1, [pv(PP_FIN), z0, pv(PP_LEN + (0 if j == 15 else 2) + l)]))
            cons.append((O.SEL_ALL, terms))
    return O.air_program(WIDTH, N_PUBLIC_CHAINED if chained else N_PUBLIC, cons)


def padding_publics(message_len, first_block=0, n_active=None):
    """the 75 public values of the padding constraints for a trace that holds blocks [first_block, first_block + n_active) of the padded
    message of message_len bytes (a whole message: all (L + 8) // 64 + 1 blocks) -> (values, (block, row) of the boundary word or None)"""
    L = int(message_len)
    k, q, r = (L + 8) // 64 + 1, L // 64, L % 64
    if n_active is None:
        n_active = k - first_block
    out = [0] * N_PAD
    has_last = first_block <= k - 1 < first_block + n_active
    has_pad = first_block <= q < first_block + n_active
    out[PP_K] = n_active
    out[PP_FIN] = int(has_last)
    out[PP_Z13] = int(has_last and q != k - 1)
    place = None
    if has_pad:
        j, c = r // 4, r % 4
        by_l2 = q != k - 1 and has_last
        out[(PP_BW2 if by_l2 else PP_BWL) + j] = 1
        out[PP_KIND + c] = 1
        for w in range(j + 1, (13 if q == k - 1 else 15) + 1):
            out[(PP_ZW2 if by_l2 else PP_ZWL) + w] = 1
        place = (q - first_block, j)
    if has_last:
        bits = 8 * L
        out[PP_LEN:PP_LEN + 4] = [bits & 0xffff, (bits >> 16) & 0xffff, (bits >> 32) & 0xffff, (bits >> 48) & 0xffff]
    return out, place


def _rotr(x, r):
    return ((x >> r) | (x << (32 - r))) & 0xffffffff


class Padded(bytes):
    """padded blocks that remember the length of the message they came from (a slice forgets it: pass message_len / first_block then)"""
    message_len = None


def pad(message):
    """FIPS 180-4 padding; the block count is then rounded up to a power of two is NOT done here: choose the message length"""
    m = bytes(message) + b"\x80"
    m += b"\x00" * ((56 - len(m)) % 64) + struct.pack(">Q", 8 * len(message))
    m = Padded(m)
    m.message_len = len(message)
    return m


def trace(blocks, total_blocks=None, chain_in=None, message_len=None, first_block=0):
    """blocks: bytes, a multiple of 64 long: blocks [first_block, ...) of the padded message of message_len bytes (default: what pad()
    remembered); total_blocks: a power of two >= their number (default: the next one), the rest are inactive all-zero blocks ->
    (trace [64 total_blocks][640] canonical, public values: 16 limbs of the final chaining value, then the 75 padding values of this slice)"""
    assert len(blocks) % 64 == 0 and len(blocks) > 0
    if message_len is None:
        message_len = getattr(blocks, "message_len", None)
    assert message_len is not None, "trace: the message length is part of the statement (pass message_len for a slice of padded blocks)"
    active = len(blocks) // 64
    pad_pub, place = padding_publics(message_len, first_block, active)
    nb = total_blocks or 1 << (active - 1).bit_length()
    assert nb & (nb - 1) == 0 and nb >= active
    blocks = bytes(blocks) + bytes(64 * (nb - active))
    t = np.zeros((64 * nb, WIDTH), dtype=np.uint32)
    h = list(chain_in) if chain_in is not None else list(IV)

    def bits(row, base, x):
        t[row, base:base + 32] = [(x >> i) & 1 for i in range(32)]

    def limbs(row, base, x):
        t[row, base], t[row, base + 1] = x & 0xffff, x >> 16

    out = None
    for blk in range(nb):
        w = list(struct.unpack(">16I", blocks[64 * blk:64 * blk + 64]))
        for i in range(16, 80):
            s0 = _rotr(w[i - 15], 7) ^ _rotr(w[i - 15], 18) ^ (w[i - 15] >> 3)
            s1 = _rotr(w[i - 2], 17) ^ _rotr(w[i - 2], 19) ^ (w[i - 2] >> 10)
            w.append((w[i - 16] + s0 + w[i - 7] + s1) & 0xffffffff)
        v = list(h)
        for r in range(64):
            row = 64 * blk + r
            a, b, c, d, e, f, g, hh = v
            t[row, SEL + r] = 1
            act = blk < active
            t[row, ACT], t[row, SKIP] = int(act), int(r == 63 and not act)
            t[row, CNT] = active - blk if act else 0
            t[row, LASTB], t[row, L2] = int(blk == active - 1), int(blk == active - 2)
            t[row, SB] = int(place == (blk, r))
            t[row, Z0], t[row, Z2] = int(r == 0 and blk == active - 1), int(r == 0 and blk == active - 2)
            for base, x in ((A, a), (B, b), (C, c), (E, e), (F, f), (G, g)):
                bits(row, base, x)
            limbs(row, D, d); limbs(row, HV, hh)
            s1 = _rotr(e, 6) ^ _rotr(e, 11) ^ _rotr(e, 25)
            ch = (e & f) ^ (~e & g & 0xffffffff)
            s0 = _rotr(a, 2) ^ _rotr(a, 13) ^ _rotr(a, 22)
            mj = (a & b) ^ (a & c) ^ (b & c)
            bits(row, S1, s1); bits(row, CH, ch); bits(row, S0, s0); bits(row, MJ, mj)
            for k in range(8):
                limbs(row, HC + 2 * k, h[k])
            add = h if r == 63 else [0] * 8
            srcs = [[hh, s1, ch, K[r], w[r], s0, mj, add[0]], [a, add[1]], [b, add[2]], [c, add[3]],
                    [d, hh, s1, ch, K[r], w[r], add[4]], [e, add[5]], [f, add[6]], [g, add[7]]]
            if r == 63 and not act:
                srcs = [[x] for x in h]
            cys = [(CY_A, 3), (CY_W6, 1), (CY_W6 + 2, 1), (CY_W6 + 4, 1), (CY_E, 3), (CY_W6 + 6, 1), (CY_W6 + 8, 1), (CY_W6 + 10, 1)]
            out = []
            for k, (src, (cy, ncy)) in enumerate(zip(srcs, cys)):
                lo = sum(x & 0xffff for x in src)
                hi = sum(x >> 16 for x in src) + (lo >> 16)
                t[row, OUT + 2 * k], t[row, OUT + 2 * k + 1] = lo & 0xffff, hi & 0xffff
                for q in range(ncy):
                    t[row, cy + q] = ((lo >> 16) >> q) & 1
                    t[row, cy + ncy + q] = ((hi >> 16) >> q) & 1
                assert (lo >> 16) < (1 << ncy) and (hi >> 16) < (1 << ncy)
                out.append((lo & 0xffff) | ((hi & 0xffff) << 16))
            bits(row, X0, w[r]); bits(row, X13, w[r + 13])
            for j in list(range(1, 13)) + [14, 15]:
                limbs(row, _xl(j), w[r + j])
            bits(row, SG0, _rotr(w[r], 7) ^ _rotr(w[r], 18) ^ (w[r] >> 3))
            bits(row, SG1, _rotr(w[r + 13], 17) ^ _rotr(w[r + 13], 19) ^ (w[r + 13] >> 10))
            if r < 63:
                parts = [_rotr(w[r + 14], 17) ^ _rotr(w[r + 14], 19) ^ (w[r + 14] >> 10), w[r + 9],
                         _rotr(w[r + 1], 7) ^ _rotr(w[r + 1], 18) ^ (w[r + 1] >> 3), w[r]]
                lo = sum(x & 0xffff for x in parts)
                hi = sum(x >> 16 for x in parts) + (lo >> 16)
                for q in range(2):
                    t[row, CY_SCHED + q] = ((lo >> 16) >> q) & 1
                    t[row, CY_SCHED + 2 + q] = ((hi >> 16) >> q) & 1
            t1 = (hh + s1 + ch + K[r] + w[r]) & 0xffffffff
            t2 = (s0 + mj) & 0xffffffff
            v = [(t1 + t2) & 0xffffffff, a, b, c, (d + t1) & 0xffffffff, e, f, g]
            if r == 63:
                v = [(x + y) & 0xffffffff for x, y in zip(v, h)] if act else list(h)
            assert out == v
        h = v
    pub = []
    for x in h:
        pub += [x & 0xffff, x >> 16]
    return t, pub + pad_pub


def chained_publics(out_pub, in_limbs):
    """public values of the CHAINED program from trace()'s: final limbs, the initial chaining value's limbs, the slice's padding values"""
    return list(out_pub[:N_DIGEST]) + list(in_limbs) + list(out_pub[N_DIGEST:])


def digest_bytes(pub):
    return b"".join(struct.pack(">I", pub[2 * k] | (pub[2 * k + 1] << 16)) for k in range(8))


def check_rows(prog, t, pub):
    """every constraint of `prog` on every row of `t` (plain Python integers): returns the list of (constraint, row) that fail"""
    prog = [int(x) for x in prog]
    n = t.shape[0]
    loc = t.astype(object)
    nxt = np.roll(loc, -1, axis=0)
    p, bad = 6, []
    for k in range(prog[3]):
        sel, nt = prog[p], prog[p + 1]
        p += 2
        acc = np.zeros(n, dtype=object)
        for _ in range(nt):
            coeff, d = prog[p], prog[p + 1]
            p += 2
            prod = np.full(n, coeff, dtype=object)
            for _j in range(d):
                v = prog[p]; p += 1
                kind, idx = v >> 30, v & 0xffff
                prod = prod * (loc[:, idx] if kind == 0 else (nxt[:, idx] if kind == 1 else pub[idx])) % P
            acc = (acc + prod) % P
        rows = range(n) if sel == 0 else ([0] if sel == 1 else ([n - 1] if sel == 2 else range(n - 1)))
        bad += [(k, r) for r in rows if acc[r] != 0]
    return bad


if __name__ == "__main__":
    prog = program()
    for msg, total in ((b"abc", None), (bytes(range(150)), None), (b"abc", 2), (b"", None), (bytes(55), None), (bytes(56), None), (bytes(63), 4), (bytes(64), None), (bytes(119), None), (bytes(121), 8)):
        tr, pub = trace(pad(msg), total)
        assert digest_bytes(pub) == hashlib.sha256(msg).digest()
        print("rows", tr.shape[0], "program words", prog.size, "constraints", prog[3], "failing", check_rows(prog, tr, pub)[:5])
