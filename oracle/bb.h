/*
 * oracle/bb.h -- BabyBear field and its quartic extension, CPU restatement.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into or called by
 * the product (zktls_amd/, libzkhip.so).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it, and only as the checker/baseline.
 *
 * PARITY UNPINNED: the reference (the3cloud/zktls) contains no prover arithmetic;
 * the field lives in the un-vendored crate p3-baby-bear 0.2.1-succinct
 * (reference Cargo.lock:3845), pulled in by sp1-sdk 4.1.4 through the call sites
 * crates/guest-prover-sp1/src/sp1.rs:113,116,120.  The reference holds no golden
 * vector for it, so this file restates the PUBLISHED
 * definition: p = 2^31 - 2^27 + 1, multiplicative generator 31, two-adicity 27,
 * extension F_p[x]/(x^4 - 11).  It is pinned only by first-principles KATs
 * (tests/test_oracle_field.py), not by reference outputs.
 *
 * Representation: CANONICAL residues in [0, p) -- deliberately NOT the Montgomery
 * form the HIP path computes in, so the two implementations share no arithmetic.
 */
#ifndef ORACLE_BB_H
#define ORACLE_BB_H

#include <stdint.h>
#include <stddef.h>

#define BB_P 2013265921u          /* 0x78000001 */
#define BB_GEN 31u                /* generator of F_p^*            */
#define BB_TWO_ADICITY 27
#define BB_TWO_ADIC_GEN 440564289u /* 31^15: element of order 2^27  */
#define BB_EXT_W 11u              /* x^4 = 11                       */

typedef uint32_t bb_t;
typedef struct { bb_t c[4]; } bb4_t;

static inline bb_t bb_add(bb_t a, bb_t b) { uint32_t s = a + b; return s >= BB_P ? s - BB_P : s; }
static inline bb_t bb_sub(bb_t a, bb_t b) { return a >= b ? a - b : a + BB_P - b; }
static inline bb_t bb_neg(bb_t a) { return a ? BB_P - a : 0; }
static inline bb_t bb_mul(bb_t a, bb_t b) { return (bb_t)(((uint64_t)a * b) % BB_P); }

static inline bb_t bb_pow(bb_t a, uint64_t e) {
    bb_t r = 1;
    while (e) { if (e & 1) r = bb_mul(r, a); a = bb_mul(a, a); e >>= 1; }
    return r;
}
static inline bb_t bb_inv(bb_t a) { return bb_pow(a, (uint64_t)BB_P - 2); }

/* element of multiplicative order 2^bits */
static inline bb_t bb_two_adic_generator(int bits) {
    bb_t g = BB_TWO_ADIC_GEN;
    for (int i = bits; i < BB_TWO_ADICITY; i++) g = bb_mul(g, g);
    return g;
}

/* Montgomery form conversions (R = 2^32); only used at test boundaries. */
static inline bb_t bb_to_monty(bb_t a) { return (bb_t)((((uint64_t)a) << 32) % BB_P); }
static inline bb_t bb_from_monty(bb_t a) { return bb_mul(a, 943718400u /* 2^-32 mod p */); }

/* ---------------- quartic extension F_p[x]/(x^4 - 11) ---------------- */
static inline bb4_t bb4_zero(void) { bb4_t r = {{0, 0, 0, 0}}; return r; }
static inline bb4_t bb4_one(void) { bb4_t r = {{1, 0, 0, 0}}; return r; }
static inline bb4_t bb4_from_base(bb_t a) { bb4_t r = {{a, 0, 0, 0}}; return r; }
static inline int bb4_eq(bb4_t a, bb4_t b) {
    return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3];
}
static inline bb4_t bb4_add(bb4_t a, bb4_t b) {
    bb4_t r; for (int i = 0; i < 4; i++) r.c[i] = bb_add(a.c[i], b.c[i]); return r;
}
static inline bb4_t bb4_sub(bb4_t a, bb4_t b) {
    bb4_t r; for (int i = 0; i < 4; i++) r.c[i] = bb_sub(a.c[i], b.c[i]); return r;
}
static inline bb4_t bb4_neg(bb4_t a) {
    bb4_t r; for (int i = 0; i < 4; i++) r.c[i] = bb_neg(a.c[i]); return r;
}
static inline bb4_t bb4_mul_base(bb4_t a, bb_t b) {
    bb4_t r; for (int i = 0; i < 4; i++) r.c[i] = bb_mul(a.c[i], b); return r;
}
static inline bb4_t bb4_add_base(bb4_t a, bb_t b) { a.c[0] = bb_add(a.c[0], b); return a; }
static inline bb4_t bb4_sub_base(bb4_t a, bb_t b) { a.c[0] = bb_sub(a.c[0], b); return a; }

/* schoolbook product, reduced with x^4 = 11 */
static inline bb4_t bb4_mul(bb4_t a, bb4_t b) {
    uint64_t t[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            t[i + j] = (t[i + j] + (uint64_t)bb_mul(a.c[i], b.c[j])) % BB_P;
    bb4_t r;
    for (int i = 0; i < 4; i++) {
        uint64_t v = t[i];
        if (i + 4 < 7) v = (v + (uint64_t)BB_EXT_W * t[i + 4]) % BB_P;
        r.c[i] = (bb_t)v;
    }
    return r;
}
static inline bb4_t bb4_pow(bb4_t a, uint64_t e) {
    bb4_t r = bb4_one();
    while (e) { if (e & 1) r = bb4_mul(r, a); a = bb4_mul(a, a); e >>= 1; }
    return r;
}
/* Frobenius x -> x^p: (sum a_i x^i)^p = sum a_i z^i x^i with z = 11^((p-1)/4), because
 * x^p = x * (x^4)^((p-1)/4) and a_i^p = a_i. */
#define BB_FROB_Z 1728404513u
static inline bb4_t bb4_frobenius(bb4_t a) {
    bb_t z2 = bb_mul(BB_FROB_Z, BB_FROB_Z), z3 = bb_mul(z2, BB_FROB_Z);
    bb4_t r = {{a.c[0], bb_mul(a.c[1], BB_FROB_Z), bb_mul(a.c[2], z2), bb_mul(a.c[3], z3)}};
    return r;
}
/* inverse through the norm to F_p: r = 1 + p + p^2 + p^3, a^r lies in F_p,
 * a^-1 = a^(r-1) / a^r with a^(r-1) = a^p a^(p^2) a^(p^3).  (tests/test_oracle.py checks it
 * against the plain Fermat power a^(p^4-2) computed in Python.) */
static inline bb4_t bb4_inv(bb4_t a) {
    bb4_t ap = bb4_frobenius(a);
    bb4_t ap2 = bb4_frobenius(ap);
    bb4_t ap3 = bb4_frobenius(ap2);
    bb4_t conj = bb4_mul(bb4_mul(ap, ap2), ap3);   /* a^(r-1) */
    bb4_t norm = bb4_mul(a, conj);                 /* in F_p: c[1..3] == 0 */
    bb_t ninv = bb_inv(norm.c[0]);
    return bb4_mul_base(conj, ninv);
}

static inline uint32_t bb_reverse_bits(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

#endif /* ORACLE_BB_H */
