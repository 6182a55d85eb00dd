/*
 * oracle/air.c -- constraint programs: an AIR supplied as DATA, CPU restatement.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/bb.h).  PARITY UNPINNED: upstream, constraints are Rust `Air::eval` bodies driven through
 * p3-uni-stark's symbolic / prover / verifier constraint folders (p3-air, p3-uni-stark 0.2.1-succinct, reference
 * Cargo.lock:3835, 4055; sp1-stark :6172), reached from crates/guest-prover-sp1/src/sp1.rs:116.  A folder sees a constraint as
 * a polynomial in the local / next row, the public values and the selectors is_first_row / is_last_row / is_transition,
 * and folds acc = acc * alpha + constraint.  The program below is that polynomial written out in sum-of-products form.
 *
 * Program (u32 words, canonical residues):
 *   [0] 0x50524941 "AIRP"   [1] 1 (format)   [2] width   [3] number of constraints K   [4] n_public   [5] total words
 *   then K constraints:  selector (0 every row, 1 first row, 2 last row, 3 transition), n_terms,
 *                        then n_terms terms:  coefficient, degree d (0..5), d variables
 *   variable: bits 31..30 = kind (0 local row, 1 next row, 2 public value), bits 15..0 = column / index.
 * value of a constraint = selector * sum_t coeff_t * prod_j var_tj; its degree (largest d, + 1 with a selector) is at most 5:
 * degree <= 3 gives two quotient chunks (log_quotient_degree 1), degree 4 or 5 four chunks (log_quotient_degree 2, log_blowup >= 2).
 */
#include <stdlib.h>
#include <string.h>

#include "oracle.h"
#include "stark_internal.h"

#define AIR_MAGIC 0x50524941u

int orc_air_validate(const uint32_t* prog, size_t words, size_t width, size_t n_public) {
    if (!prog || words < 6 || prog[0] != AIR_MAGIC || prog[1] != 1 || prog[2] != width || prog[4] != n_public || prog[5] != words) return 0;
    if (prog[3] == 0 || prog[3] > (1u << 20)) return 0;
    size_t p = 6;
    for (uint32_t k = 0; k < prog[3]; k++) {
        if (p + 2 > words) return 0;
        uint32_t sel = prog[p++], nt = prog[p++];
        if (sel > 3 || nt == 0) return 0;
        for (uint32_t t = 0; t < nt; t++) {
            if (p + 2 > words) return 0;
            uint32_t coeff = prog[p++], d = prog[p++];
            if (coeff >= BB_P || d > 5 || d + (sel ? 1 : 0) > 5 || p + d > words) return 0;
            for (uint32_t j = 0; j < d; j++) {
                uint32_t v = prog[p++], kind = v >> 30, idx = v & 0xFFFFu;
                if ((v & 0x3FFF0000u) || kind > 2) return 0;
                if (kind == 2 ? idx >= n_public : idx >= width) return 0;
            }
        }
    }
    return p == words;
}

/* log2 of the number of quotient chunks: degree <= 3 -> 1 (two chunks), degree 4 or 5 -> 2 (four chunks; needs log_blowup >= 2) */
int orc_air_log_quotient_degree(const uint32_t* prog) {
    uint32_t maxd = 0;
    size_t p = 6;
    for (uint32_t k = 0; k < prog[3]; k++) {
        uint32_t sel = prog[p++], nt = prog[p++];
        for (uint32_t t = 0; t < nt; t++) {
            p++;
            uint32_t d = prog[p++];
            if (d + (sel ? 1 : 0) > maxd) maxd = d + (sel ? 1 : 0);
            p += d;
        }
    }
    return maxd <= 3 ? 1 : 2;
}

/* digest of the program: the width-16 sponge over the 16-bit halves of every word (halves are field elements whatever the word) */
void orc_air_digest(const uint32_t* prog, size_t words, uint32_t out[8]) {
    uint32_t* limbs = (uint32_t*)malloc(2 * words * 4);
    for (size_t i = 0; i < words; i++) { limbs[2 * i] = prog[i] & 0xFFFFu; limbs[2 * i + 1] = prog[i] >> 16; }
    orc_sponge_hash(limbs, 2 * words, out);
    free(limbs);
}

/* base-field evaluation on a row of the quotient domain: acc = acc * alpha + sel * C_k, in program order */
bb4_t orc__air_fold_base(const uint32_t* prog, const uint32_t* local, const uint32_t* next, const uint32_t* pub,
                         bb_t sel_first, bb_t sel_last, bb_t sel_trans, bb4_t alpha) {
    bb4_t acc = bb4_zero();
    size_t p = 6;
    for (uint32_t k = 0; k < prog[3]; k++) {
        uint32_t sel = prog[p++], nt = prog[p++];
        bb_t c = 0;
        for (uint32_t t = 0; t < nt; t++) {
            bb_t prod = prog[p++];
            uint32_t d = prog[p++];
            for (uint32_t j = 0; j < d; j++) {
                uint32_t v = prog[p++], kind = v >> 30, idx = v & 0xFFFFu;
                prod = bb_mul(prod, kind == 0 ? local[idx] : (kind == 1 ? next[idx] : pub[idx]));
            }
            c = bb_add(c, prod);
        }
        if (sel == 1) c = bb_mul(c, sel_first); else if (sel == 2) c = bb_mul(c, sel_last); else if (sel == 3) c = bb_mul(c, sel_trans);
        acc = bb4_add_base(bb4_mul(acc, alpha), c);
    }
    return acc;
}
/* the same on opened (extension) values: the verifier's side */
bb4_t orc__air_fold_ext(const uint32_t* prog, const bb4_t* local, const bb4_t* next, const uint32_t* pub,
                        bb4_t sel_first, bb4_t sel_last, bb4_t sel_trans, bb4_t alpha) {
    bb4_t acc = bb4_zero();
    size_t p = 6;
    for (uint32_t k = 0; k < prog[3]; k++) {
        uint32_t sel = prog[p++], nt = prog[p++];
        bb4_t c = bb4_zero();
        for (uint32_t t = 0; t < nt; t++) {
            bb4_t prod = bb4_from_base(prog[p++]);
            uint32_t d = prog[p++];
            for (uint32_t j = 0; j < d; j++) {
                uint32_t v = prog[p++], kind = v >> 30, idx = v & 0xFFFFu;
                prod = bb4_mul(prod, kind == 0 ? local[idx] : (kind == 1 ? next[idx] : bb4_from_base(pub[idx])));
            }
            c = bb4_add(c, prod);
        }
        if (sel == 1) c = bb4_mul(c, sel_first); else if (sel == 2) c = bb4_mul(c, sel_last); else if (sel == 3) c = bb4_mul(c, sel_trans);
        acc = bb4_add(bb4_mul(acc, alpha), c);
    }
    return acc;
}

/* quotient values of a program on the coset g * <w_{2^lqd N}>: the first 2^lqd N rows of the bit-reversed LDE; out[p] (extension), bit-reversed */
void orc_quotient_values_air(const uint32_t* prog, const uint32_t* lde, int log_n, size_t width, const uint32_t* pub,
                             const uint32_t alpha_[4], int lqd, uint32_t* out) {
    const int log_m = log_n + lqd;
    const size_t step = (size_t)1 << lqd;              /* the next trace row is `step` points further on the quotient domain */
    const size_t m = (size_t)1 << log_m, n = (size_t)1 << log_n;
    const bb4_t alpha = orc__ld4(alpha_);
    const bb_t w = bb_two_adic_generator(log_m), wn_inv = bb_inv(bb_two_adic_generator(log_n));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < m; i++) {
        bb_t x = bb_mul(BB_GEN, bb_pow(w, i));
        bb_t zh = bb_sub(bb_pow(x, n), 1);
        bb_t sel_first = bb_mul(zh, bb_inv(bb_sub(x, 1)));
        bb_t sel_last = bb_mul(zh, bb_inv(bb_sub(x, wn_inv)));
        bb_t sel_trans = bb_sub(x, wn_inv);
        size_t p = bb_reverse_bits((uint32_t)i, log_m), pn = bb_reverse_bits((uint32_t)((i + step) & (m - 1)), log_m);
        bb4_t acc = orc__air_fold_base(prog, lde + p * width, lde + pn * width, pub, sel_first, sel_last, sel_trans, alpha);
        orc__st4(out + 4 * p, bb4_mul_base(acc, bb_inv(zh)));
    }
}

/* the synthetic AIR of DESIGN.md section 3 (no lookups) as a program; returns the word count (0: buffer too small) */
size_t orc_air_synthetic(size_t width, size_t n_public, uint32_t* out, size_t cap) {
    const size_t G = width / 4, words = 6 + G * ((2 + 3 + 5 + 2) + (2 + 3 + 4 + 3 + 2) + (2 + 3 + 2));
    if (width % 4 || cap < words) return 0;
    size_t p = 0;
    out[p++] = AIR_MAGIC; out[p++] = 1; out[p++] = (uint32_t)width; out[p++] = (uint32_t)(3 * G); out[p++] = (uint32_t)n_public; out[p++] = (uint32_t)words;
    for (size_t g = 0; g < G; g++) {
        const uint32_t a = (uint32_t)(4 * g), b = a + 1, c = a + 2, d = a + 3, NEXT = 1u << 30;
        /* c - a a b - (g + 1) on every row */
        out[p++] = 0; out[p++] = 3;
        out[p++] = 1; out[p++] = 1; out[p++] = c;
        out[p++] = BB_P - 1; out[p++] = 3; out[p++] = a; out[p++] = a; out[p++] = b;
        out[p++] = BB_P - (uint32_t)((g + 1) % BB_P); out[p++] = 0;
        /* d' - a b - c - (2g + 3) on transitions */
        out[p++] = 3; out[p++] = 4;
        out[p++] = 1; out[p++] = 1; out[p++] = NEXT | d;
        out[p++] = BB_P - 1; out[p++] = 2; out[p++] = a; out[p++] = b;
        out[p++] = BB_P - 1; out[p++] = 1; out[p++] = c;
        out[p++] = BB_P - (uint32_t)((2 * g + 3) % BB_P); out[p++] = 0;
        /* d - (5g + 7) on the first row */
        out[p++] = 1; out[p++] = 2;
        out[p++] = 1; out[p++] = 1; out[p++] = d;
        out[p++] = BB_P - (uint32_t)((5 * g + 7) % BB_P); out[p++] = 0;
    }
    return p == words ? words : 0;
}
