/*
 * oracle/stark.c -- synthetic shard AIR, quotient, PCS opening, FRI prover and the
 * verifier, CPU restatement.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED
 * (see oracle/oracle.h).
 *
 * What it restates (published algorithms; the implementing crates are absent from
 * /root/reference, reference call site crates/guest-prover-sp1/src/sp1.rs:116):
 *   p3-uni-stark 0.2.1-succinct `prove`/`verify` (Cargo.lock:4055): commit trace,
 *     sample alpha, quotient on the disjoint coset, split in 2^log_qd chunks, commit,
 *     sample zeta, open trace at zeta and zeta*g, chunks at zeta;
 *   p3-fri 0.2.1-succinct TwoAdicFriPcs::{commit,open} + prover::{commit_phase,
 *     answer_query} + verifier (Cargo.lock:3930): alpha-batched reduced openings
 *     (sum alpha^k (p(x) - p(z)) / (x - z)), fold-by-2 with beta per layer, one
 *     Merkle commitment per layer, PoW grinding, num_queries index openings;
 *   sp1-stark 4.1.4 BabyBearPoseidon2 parameters for core shards (Cargo.lock:6172):
 *     log_blowup 1, 100 queries, 16 PoW bits.
 * What is NOT restated: SP1's real chip AIRs (sp1-core-machine, unobtainable here).
 * They are replaced by the documented synthetic AIR below.
 *
 * Deviation from upstream, on purpose: the opened values are observed by the
 * challenger before the FRI batching challenge is drawn (upstream 0.2.1 draws it
 * first); PoW takes the smallest witness.
 */
#include "oracle.h"
#include "stark_internal.h"
#include <stdlib.h>
#include <string.h>
#define fold_constraints_ext orc__fold_constraints_ext
#define fri_fold_row orc__fri_fold_row
#define row_dot orc__row_dot
#define copy_path orc__copy_path
#define recombine orc__recombine
#define fold_logup orc__fold_logup

/* ------------------------------------------------------------------ */
/* synthetic data                                                      */
/* ------------------------------------------------------------------ */
static uint64_t mix64(uint64_t z) {         /* splitmix64 output function */
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
/* = the (index+1)-th output of splitmix64 seeded with `seed`, reduced mod p */
uint32_t orc_synth_value(uint64_t seed, uint64_t index) {
    return (uint32_t)(mix64(seed + index * 0x9E3779B97F4A7C15ull) % BB_P);
}
void orc_fill_uniform(uint64_t seed, int log_n, size_t width, uint32_t* out) {
    size_t total = ((size_t)1 << log_n) * width;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < total; i++) out[i] = orc_synth_value(seed, i);
}

/* Synthetic AIR, width = 4 G, group g owns columns (a, b, c, d) = 4g .. 4g+3:
 *   C1_g  all rows    :  c - a*a*b - K1_g                = 0     (degree 3)
 *   C2_g  transition  :  d' - a*b - c - K2_g             = 0     (degree 2)
 *   C3_g  first row   :  d - D0_g                        = 0     (degree 1)
 * K1_g = g+1, K2_g = 2g+3, D0_g = 5g+7.  a, b are free (uniform); c, d are derived. */
static inline bb_t air_k1(size_t g) { return (bb_t)((g + 1) % BB_P); }
static inline bb_t air_k2(size_t g) { return (bb_t)((2 * g + 3) % BB_P); }
static inline bb_t air_d0(size_t g) { return (bb_t)((5 * g + 7) % BB_P); }

void orc_gen_trace(uint64_t seed, uint64_t shard, int log_n, size_t width, uint32_t* out) {
    size_t n = (size_t)1 << log_n, G = width / 4;
    uint64_t s = seed + shard;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        for (size_t g = 0; g < G; g++) {
            uint32_t* row = out + i * width + 4 * g;
            bb_t a = orc_synth_value(s, i * width + 4 * g);
            bb_t b = orc_synth_value(s, i * width + 4 * g + 1);
            bb_t c = bb_add(bb_mul(bb_mul(a, a), b), air_k1(g));
            bb_t d;
            if (i == 0) d = air_d0(g);
            else {
                bb_t pa = orc_synth_value(s, (i - 1) * width + 4 * g);
                bb_t pb = orc_synth_value(s, (i - 1) * width + 4 * g + 1);
                bb_t pc = bb_add(bb_mul(bb_mul(pa, pa), pb), air_k1(g));
                d = bb_add(bb_add(bb_mul(pa, pb), pc), air_k2(g));
            }
            row[0] = a; row[1] = b; row[2] = c; row[3] = d;
        }
    }
}

size_t orc_check_trace(const uint32_t* t, int log_n, size_t width) {
    size_t n = (size_t)1 << log_n, G = width / 4, bad = 0;
    for (size_t i = 0; i < n; i++)
        for (size_t g = 0; g < G; g++) {
            const uint32_t* r = t + i * width + 4 * g;
            if (r[2] != bb_add(bb_mul(bb_mul(r[0], r[0]), r[1]), air_k1(g))) bad++;
            if (i + 1 < n) {
                const uint32_t* nx = t + (i + 1) * width + 4 * g;
                if (nx[3] != bb_add(bb_add(bb_mul(r[0], r[1]), r[2]), air_k2(g))) bad++;
            }
            if (i == 0 && r[3] != air_d0(g)) bad++;
        }
    return bad;
}

/* ------------------------------------------------------------------ */
/* LogUp lookups (sp1-stark permutation trace)      */
/* ------------------------------------------------------------------ */
/* With `pairs` = Q > 0, group 2q+1 RECEIVES what group 2q SENDS: its (a, b) columns are the
 * sender's (a, b) under the row permutation pi(i) = 5 i + 3 mod N, which no local constraint
 * can express.  The lookup argument (LogUp, as sp1-stark's generate_permutation_trace builds
 * it per chip): after the main commitment the verifier draws gamma, beta; the permutation
 * trace has Q + 1 extension columns
 *     phi_q[i] = 1/(gamma + a_s[i] + beta b_s[i]) - 1/(gamma + a_r[i] + beta b_r[i])
 *     S[i]     = sum_{r <= i} sum_q phi_q[r]                     (cumulative sum)
 * with constraints (appended after the 3 G main ones, same alpha folding):
 *     L_q  all rows    : phi_q den_s den_r - (den_r - den_s)      = 0   (degree 3)
 *     T1   first row   : S - sum_q phi_q                           = 0
 *     T2   transition  : S' - S - sum_q phi'_q                     = 0
 *     T3   last row    : S - C                                     = 0   (C = 0: the multisets of this table are equal;
 *                        tables that look each other up expose C, and the C's of a shard sum to 0) */
static inline size_t logup_perm_row(size_t i, int log_n) { return (5 * i + 3) & (((size_t)1 << log_n) - 1); }

void orc_gen_trace_logup(uint64_t seed, uint64_t shard, int log_n, size_t width, int pairs, uint32_t* out) {
    size_t n = (size_t)1 << log_n, G = width / 4;
    uint64_t s = seed + shard;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        for (size_t g = 0; g < G; g++) {
            uint32_t* row = out + i * width + 4 * g;
            /* receivers read the sender group's stream at the permuted row */
            int recv = (g & 1) && (int)(g / 2) < pairs;
            size_t sg = recv ? g - 1 : g;
            size_t ri = recv ? logup_perm_row(i, log_n) : i;
            bb_t a = orc_synth_value(s, ri * width + 4 * sg);
            bb_t b = orc_synth_value(s, ri * width + 4 * sg + 1);
            bb_t c = bb_add(bb_mul(bb_mul(a, a), b), air_k1(g));
            bb_t d;
            if (i == 0) d = air_d0(g);
            else {
                size_t pi = recv ? logup_perm_row(i - 1, log_n) : i - 1;
                bb_t pa = orc_synth_value(s, pi * width + 4 * sg);
                bb_t pb = orc_synth_value(s, pi * width + 4 * sg + 1);
                bb_t pc = bb_add(bb_mul(bb_mul(pa, pa), pb), air_k1(g));
                d = bb_add(bb_add(bb_mul(pa, pb), pc), air_k2(g));
            }
            row[0] = a; row[1] = b; row[2] = c; row[3] = d;
        }
    }
}

/* Lookups BETWEEN two tables of equal height: the receiver groups (odd g < 2 pairs) of this table hold the sender groups
 * (g - 1) of the PARTNER table (stream seed + partner_shard, row pitch partner_width) under the same row permutation.
 * Each table's running sum then ends at C = sum(1/den_send own) - sum(1/den_send partner) != 0; the two C's cancel. */
void orc_gen_trace_logup_cross(uint64_t seed, uint64_t shard, uint64_t partner_shard, int log_n, size_t width, size_t partner_width,
                               int pairs, uint32_t* out) {
    size_t n = (size_t)1 << log_n, G = width / 4;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        for (size_t g = 0; g < G; g++) {
            uint32_t* row = out + i * width + 4 * g;
            int recv = (g & 1) && (int)(g / 2) < pairs;
            uint64_t s = seed + (recv ? partner_shard : shard);
            size_t w = recv ? partner_width : width;
            size_t sg = recv ? g - 1 : g;
            size_t ri = recv ? logup_perm_row(i, log_n) : i;
            bb_t a = orc_synth_value(s, ri * w + 4 * sg);
            bb_t b = orc_synth_value(s, ri * w + 4 * sg + 1);
            bb_t c = bb_add(bb_mul(bb_mul(a, a), b), air_k1(g));
            bb_t d;
            if (i == 0) d = air_d0(g);
            else {
                size_t pi = recv ? logup_perm_row(i - 1, log_n) : i - 1;
                bb_t pa = orc_synth_value(s, pi * w + 4 * sg);
                bb_t pb = orc_synth_value(s, pi * w + 4 * sg + 1);
                bb_t pc = bb_add(bb_mul(bb_mul(pa, pa), pb), air_k1(g));
                d = bb_add(bb_add(bb_mul(pa, pb), pc), air_k2(g));
            }
            row[0] = a; row[1] = b; row[2] = c; row[3] = d;
        }
    }
}

bb4_t orc__ld4(const uint32_t* p);
void orc__st4(uint32_t* p, bb4_t v);
#define ld4 orc__ld4
#define st4 orc__st4

/* N x 4 (Q + 1) words: [phi_0 | ... | phi_{Q-1} | S], extension elements flattened */
void orc_perm_trace(const uint32_t* trace, int log_n, size_t width, int pairs,
                    const uint32_t gamma_[4], const uint32_t beta_[4], uint32_t* out) {
    size_t n = (size_t)1 << log_n, wp = 4 * ((size_t)pairs + 1);
    bb4_t gamma = ld4(gamma_), beta = ld4(beta_);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        const uint32_t* row = trace + i * width;
        bb4_t sum = bb4_zero();
        for (int q = 0; q < pairs; q++) {
            bb4_t ds = bb4_add(bb4_add_base(gamma, row[8 * q]), bb4_mul_base(beta, row[8 * q + 1]));
            bb4_t dr = bb4_add(bb4_add_base(gamma, row[8 * q + 4]), bb4_mul_base(beta, row[8 * q + 5]));
            bb4_t phi = bb4_sub(bb4_inv(ds), bb4_inv(dr));
            st4(out + i * wp + 4 * q, phi);
            sum = bb4_add(sum, phi);
        }
        st4(out + i * wp + 4 * pairs, sum);          /* row sum, turned into the running sum below */
    }
    bb4_t run = bb4_zero();
    for (size_t i = 0; i < n; i++) {
        run = bb4_add(run, ld4(out + i * wp + 4 * pairs));
        st4(out + i * wp + 4 * pairs, run);
    }
}

/* ------------------------------------------------------------------ */
/* constraint folding:  acc = acc * alpha + C_k, k in AIR order         */
/* ------------------------------------------------------------------ */
static bb4_t fold_constraints_base(const uint32_t* local, const uint32_t* next, size_t width,
                                   bb_t sel_first, bb_t sel_trans, bb4_t alpha) {
    bb4_t acc = bb4_zero();
    size_t G = width / 4;
    for (size_t g = 0; g < G; g++) {
        bb_t a = local[4 * g], b = local[4 * g + 1], c = local[4 * g + 2], d = local[4 * g + 3];
        bb_t dn = next[4 * g + 3];
        bb_t c1 = bb_sub(bb_sub(c, bb_mul(bb_mul(a, a), b)), air_k1(g));
        bb_t c2 = bb_mul(sel_trans, bb_sub(bb_sub(bb_sub(dn, bb_mul(a, b)), c), air_k2(g)));
        bb_t c3 = bb_mul(sel_first, bb_sub(d, air_d0(g)));
        acc = bb4_add_base(bb4_mul(acc, alpha), c1);
        acc = bb4_add_base(bb4_mul(acc, alpha), c2);
        acc = bb4_add_base(bb4_mul(acc, alpha), c3);
    }
    return acc;
}
bb4_t orc__fold_constraints_ext(const bb4_t* local, const bb4_t* next, size_t width,
                                  bb4_t sel_first, bb4_t sel_trans, bb4_t alpha) {
    bb4_t acc = bb4_zero();
    size_t G = width / 4;
    for (size_t g = 0; g < G; g++) {
        bb4_t a = local[4 * g], b = local[4 * g + 1], c = local[4 * g + 2], d = local[4 * g + 3];
        bb4_t dn = next[4 * g + 3];
        bb4_t c1 = bb4_sub_base(bb4_sub(c, bb4_mul(bb4_mul(a, a), b)), air_k1(g));
        bb4_t c2 = bb4_mul(sel_trans, bb4_sub_base(bb4_sub(bb4_sub(dn, bb4_mul(a, b)), c), air_k2(g)));
        bb4_t c3 = bb4_mul(sel_first, bb4_sub_base(d, air_d0(g)));
        acc = bb4_add(bb4_mul(acc, alpha), c1);
        acc = bb4_add(bb4_mul(acc, alpha), c2);
        acc = bb4_add(bb4_mul(acc, alpha), c3);
    }
    return acc;
}

bb4_t orc__ld4(const uint32_t* p) { bb4_t r; memcpy(r.c, p, 16); return r; }
void orc__st4(uint32_t* p, bb4_t v) { memcpy(p, v.c, 16); }

/* LogUp constraints in extension arithmetic, continuing the Horner fold of `acc`.
 * as/bs/ar/br: sender / receiver tuple of pair q; perm_* hold phi_0..phi_{Q-1}, S. */
bb4_t orc__fold_logup(bb4_t acc, int pairs, const bb4_t* as, const bb4_t* bs, const bb4_t* ar, const bb4_t* br,
                        const bb4_t* perm_local, const bb4_t* perm_next, bb4_t gamma, bb4_t beta,
                        bb4_t sel_first, bb4_t sel_trans, bb4_t sel_last, bb4_t alpha, bb4_t cumsum) {
    bb4_t sum_l = bb4_zero(), sum_n = bb4_zero();
    for (int q = 0; q < pairs; q++) {
        bb4_t ds = bb4_add(bb4_add(gamma, as[q]), bb4_mul(beta, bs[q]));
        bb4_t dr = bb4_add(bb4_add(gamma, ar[q]), bb4_mul(beta, br[q]));
        bb4_t c = bb4_sub(bb4_mul(bb4_mul(perm_local[q], ds), dr), bb4_sub(dr, ds));
        acc = bb4_add(bb4_mul(acc, alpha), c);
        sum_l = bb4_add(sum_l, perm_local[q]);
        sum_n = bb4_add(sum_n, perm_next[q]);
    }
    bb4_t S = perm_local[pairs], Sn = perm_next[pairs];
    acc = bb4_add(bb4_mul(acc, alpha), bb4_mul(sel_first, bb4_sub(S, sum_l)));
    acc = bb4_add(bb4_mul(acc, alpha), bb4_mul(sel_trans, bb4_sub(bb4_sub(Sn, S), sum_n)));
    acc = bb4_add(bb4_mul(acc, alpha), bb4_mul(sel_last, bb4_sub(S, cumsum)));     /* cumsum = 0: lookups closed inside the table */
    return acc;
}

/* ------------------------------------------------------------------ */
/* quotient values on the coset g * <w_2N>  (log_blowup = log_qd = 1)   */
/* ------------------------------------------------------------------ */
void orc_quotient_values_logup(const uint32_t* lde, int log_n, size_t width,
                                const uint32_t* perm_lde, int pairs, const uint32_t gamma_[4], const uint32_t beta_[4],
                                const uint32_t alpha_[4], uint32_t* out) {
    const uint32_t zero[4] = {0, 0, 0, 0};
    orc_quotient_values_logup_c(lde, log_n, width, perm_lde, pairs, gamma_, beta_, alpha_, zero, out);
}
void orc_quotient_values_logup_c(const uint32_t* lde, int log_n, size_t width,
                                  const uint32_t* perm_lde, int pairs, const uint32_t gamma_[4], const uint32_t beta_[4],
                                  const uint32_t alpha_[4], const uint32_t cumsum_[4], uint32_t* out) {
    const bb4_t cumsum = ld4(cumsum_);
    int log_m = log_n + 1;
    size_t m = (size_t)1 << log_m, n = (size_t)1 << log_n, wp = 4 * ((size_t)pairs + 1);
    bb4_t alpha = ld4(alpha_);
    bb4_t gamma = pairs ? ld4(gamma_) : bb4_zero(), beta = pairs ? ld4(beta_) : bb4_zero();
    bb_t w = bb_two_adic_generator(log_m);
    bb_t wn_inv = bb_inv(bb_two_adic_generator(log_n));   /* g_N^-1: last row of H */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < m; i++) {
        bb_t x = bb_mul(BB_GEN, bb_pow(w, i));
        bb_t zh = bb_sub(bb_pow(x, n), 1);                 /* Z_H(x) = x^N - 1 */
        bb_t sel_first = bb_mul(zh, bb_inv(bb_sub(x, 1)));
        bb_t sel_trans = bb_sub(x, wn_inv);
        bb_t inv_zh = bb_inv(zh);
        size_t p = bb_reverse_bits((uint32_t)i, log_m);
        size_t pn = bb_reverse_bits((uint32_t)((i + 2) & (m - 1)), log_m);
        bb4_t acc = fold_constraints_base(lde + p * width, lde + pn * width, width,
                                          sel_first, sel_trans, alpha);
        if (pairs) {
            bb_t sel_last = bb_mul(zh, bb_inv(bb_sub(x, wn_inv)));
            bb4_t as[64], bs[64], ar[64], br[64], pl[65], pn_[65];
            const uint32_t* row = lde + p * width;
            for (int q = 0; q < pairs; q++) {
                as[q] = bb4_from_base(row[8 * q]); bs[q] = bb4_from_base(row[8 * q + 1]);
                ar[q] = bb4_from_base(row[8 * q + 4]); br[q] = bb4_from_base(row[8 * q + 5]);
            }
            for (int q = 0; q <= pairs; q++) { pl[q] = ld4(perm_lde + p * wp + 4 * q); pn_[q] = ld4(perm_lde + pn * wp + 4 * q); }
            acc = fold_logup(acc, pairs, as, bs, ar, br, pl, pn_, gamma, beta, bb4_from_base(sel_first),
                             bb4_from_base(sel_trans), bb4_from_base(sel_last), alpha, cumsum);
        }
        st4(out + 4 * p, bb4_mul_base(acc, inv_zh));
    }
}
void orc_quotient_values(const uint32_t* lde, int log_n, size_t width,
                         const uint32_t alpha_[4], uint32_t* out) {
    orc_quotient_values_logup(lde, log_n, width, NULL, 0, NULL, NULL, alpha_, out);
}

/* ------------------------------------------------------------------ */
/* barycentric opening on the low coset g * <w_N> of a bit-reversed LDE  */
/*   f(z) = ((z/g)^N - 1)/N * sum_i f(x_i) x_i / (z - x_i)              */
/* ------------------------------------------------------------------ */
void orc_open_at(const uint32_t* lde, int log_n, size_t width, const uint32_t z_[4],
                 uint32_t* out) {
    size_t n = (size_t)1 << log_n;
    bb4_t z = ld4(z_);
    bb_t w = bb_two_adic_generator(log_n);
    bb4_t* wts = (bb4_t*)malloc(n * sizeof(bb4_t));   /* indexed by row q */
#pragma omp parallel for schedule(static)
    for (size_t q = 0; q < n; q++) {
        size_t i = bb_reverse_bits((uint32_t)q, log_n);
        bb_t x = bb_mul(BB_GEN, bb_pow(w, i));
        bb4_t den = bb4_sub_base(z, x);                /* z - x */
        wts[q] = bb4_mul_base(bb4_inv(den), x);
    }
    bb4_t zg = bb4_mul_base(z, bb_inv(BB_GEN));
    bb4_t scale = bb4_mul_base(bb4_sub_base(bb4_pow(zg, n), 1), bb_inv((bb_t)(n % BB_P)));
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < width; c++) {
        bb4_t acc = bb4_zero();
        for (size_t q = 0; q < n; q++) acc = bb4_add(acc, bb4_mul_base(wts[q], lde[q * width + c]));
        st4(out + 4 * c, bb4_mul(acc, scale));
    }
    free(wts);
}

/* ------------------------------------------------------------------ */
/* FRI fold:  out[i] = (e0+e1)/2 + beta (e0-e1)/(2 x_i),  x_i = w_{2h}^{bitrev_h(i)} */
/* written as upstream's fold_row:  e0 + (beta - x)(e1 - e0)/(-2x)                   */
/* ------------------------------------------------------------------ */
bb4_t orc__fri_fold_row(size_t index, int log_folded_h, bb4_t beta, bb4_t e0, bb4_t e1) {
    bb_t x = bb_pow(bb_two_adic_generator(log_folded_h + 1),
                    bb_reverse_bits((uint32_t)index, log_folded_h));
    bb_t inv = bb_inv(bb_sub(0, bb_add(x, x)));            /* 1 / (x1 - x0) = 1/(-2x) */
    bb4_t t = bb4_mul(bb4_sub_base(beta, x), bb4_sub(e1, e0));
    return bb4_add(e0, bb4_mul_base(t, inv));
}
void orc_fri_fold(const uint32_t* in, int log_h, const uint32_t beta_[4], uint32_t* out) {
    size_t half = (size_t)1 << (log_h - 1);
    bb4_t beta = ld4(beta_);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < half; i++)
        st4(out + 4 * i, fri_fold_row(i, log_h - 1, beta, ld4(in + 8 * i), ld4(in + 8 * i + 4)));
}

/* FRI fold of arity 2^k (RISC Zero folds by 16: risc0-zkp `fri_fold`, reference Cargo.lock:5057,
 * call site crates/guest-prover-r0/src/prover.rs:90).  Restated from the definition: the 2^k
 * consecutive (bit-reversed) entries of `in` are the evaluations of f on one coset
 * { x w_{2^k}^j }; out[i] = value at beta of the degree < 2^k interpolant through them
 * (= sum_j beta^j f_j(x^(2^k)) for f = sum_j X^j f_j(X^(2^k))).  Naive Lagrange form. */
void orc_fri_fold_k(const uint32_t* in, int log_h, int log_arity, const uint32_t beta_[4], uint32_t* out) {
    size_t arity = (size_t)1 << log_arity, nout = (size_t)1 << (log_h - log_arity);
    bb4_t beta = ld4(beta_);
    bb_t w = bb_two_adic_generator(log_h);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < nout; i++) {
        bb_t xs[64];
        for (size_t j = 0; j < arity; j++)
            xs[j] = bb_pow(w, bb_reverse_bits((uint32_t)(i * arity + j), log_h));
        bb4_t acc = bb4_zero();
        for (size_t j = 0; j < arity; j++) {
            bb4_t num = bb4_one();
            bb_t den = 1;
            for (size_t l = 0; l < arity; l++) {
                if (l == j) continue;
                num = bb4_mul(num, bb4_sub_base(beta, xs[l]));
                den = bb_mul(den, bb_sub(xs[j], xs[l]));
            }
            acc = bb4_add(acc, bb4_mul(ld4(in + 4 * (i * arity + j)), bb4_mul_base(num, bb_inv(den))));
        }
        st4(out + 4 * i, acc);
    }
}

/* ------------------------------------------------------------------ */
/* proof layout (all words little-endian u32, canonical residues)       */
/* ------------------------------------------------------------------ */
/* constraint program in effect for the current prove / verify call (orc_prove_shard_air, orc_verify_shard_air); NULL: the
 * built-in synthetic AIR.  With a program the proof is version 7: extended header, then the 8-word program digest. */
static __thread const uint32_t* g_air = NULL;
static __thread size_t g_air_words = 0;
static __thread int g_air_lqd = 1;      /* log2 of the number of quotient chunks: 1 (degree <= 3) or 2 (degree <= 5, needs log_blowup >= 2) */
#define NQ_CUR ((size_t)1 << (g_air ? g_air_lqd : 1))
#define PROOF_MAGIC 0x41544B5Au   /* "ZKTA" */
#define PROOF_VERSION 1u

/* shape parameters with their defaults resolved (0 = SP1 shape) */
typedef struct { int b, K, F, hw, R, ext; } shape_t;
static int shape_of(int log_n, size_t width, const orc_params_t* prm, shape_t* sh) {
    sh->b = prm->log_blowup;
    sh->K = prm->log_fold ? prm->log_fold : 1;
    sh->F = prm->log_final;
    sh->hw = prm->hash_width ? prm->hash_width : 16;
    sh->ext = !(sh->b == 1 && sh->K == 1 && sh->F == 0 && sh->hw == 16) || prm->code_width != 0;   /* extended header / transcript */
    if (prm->code_width < 0 || prm->code_width % 4 != 0 || (prm->code_width && (size_t)prm->code_width >= width)) return 0;
    if (sh->b < 1 || sh->b > 3 || width % 4 != 0 || width == 0) return 0;
    if (sh->K < 1 || sh->K > 5 || sh->F < 0 || sh->F > 10 || sh->F > log_n || (log_n - sh->F) % sh->K != 0) return 0;
    if (sh->hw != 16 && sh->hw != 24) return 0;
    if (prm->logup_pairs < 0 || prm->logup_pairs > 64 || (size_t)prm->logup_pairs * 8 > width) return 0;
    if (log_n + sh->b > 27) return 0;
    sh->R = (log_n - sh->F) / sh->K;
    return 1;
}

size_t orc_proof_size(int log_n, size_t width, const orc_params_t* prm, size_t n_public) {
    (void)n_public;
    shape_t sh;
    if (!shape_of(log_n, width, prm, &sh)) return 0;
    size_t H = (size_t)(log_n + sh.b);
    size_t Q = (size_t)prm->logup_pairs, wp = Q ? 4 * (Q + 1) : 0;
    const size_t CW = g_air ? 0 : (size_t)prm->code_width;                 /* code / data split: one more header word, root and path */
    size_t words = (g_air ? 20 : (sh.ext ? 12 : (Q ? 9 : 8))) + (CW ? 9 : 0) + 16 + 8 * width + 16 * NQ_CUR + 8 * (size_t)sh.R + 4 * ((size_t)1 << sh.F) + 1;
    size_t perq = width + 4 * NQ_CUR + 16 * H + (CW ? 8 * H : 0);
    if (Q) { words += 8 + 8 * wp; perq += wp + 8 * H; }
    for (int l = 0; l < sh.R; l++) perq += 4 * (((size_t)1 << sh.K) - 1) + 8 * (H - (size_t)sh.K * (l + 1));
    words += (size_t)prm->num_queries * perq;
    return words * 4;
}

static __thread orc_prove_debug_t g_dbg;
void orc_last_prove_debug(orc_prove_debug_t* out) { *out = g_dbg; }

static bb4_t sample_ext(orc_challenger_t* ch) { bb4_t r; orc_chal_sample_ext(ch, r.c); return r; }

static void transcript_init(orc_challenger_t* ch, int log_n, size_t width,
                            const orc_params_t* prm, size_t n_public, const shape_t* sh) {
    orc_chal_init(ch);
    orc_chal_observe(ch, (uint32_t)log_n);
    orc_chal_observe(ch, (uint32_t)width);
    orc_chal_observe(ch, (uint32_t)prm->log_blowup);
    orc_chal_observe(ch, (uint32_t)prm->num_queries);
    orc_chal_observe(ch, (uint32_t)prm->pow_bits);
    orc_chal_observe(ch, (uint32_t)n_public);
    if (sh->ext || g_air) {
        orc_chal_observe(ch, (uint32_t)prm->logup_pairs);
        orc_chal_observe(ch, (uint32_t)sh->K);
        orc_chal_observe(ch, (uint32_t)sh->F);
        orc_chal_observe(ch, (uint32_t)sh->hw);
    } else if (prm->logup_pairs) orc_chal_observe(ch, (uint32_t)prm->logup_pairs);
    if (!g_air && prm->code_width) orc_chal_observe(ch, (uint32_t)prm->code_width);
    if (g_air) {
        uint32_t dg[8];
        orc_air_digest(g_air, g_air_words, dg);
        orc_chal_observe_slice(ch, dg, 8);
    }
}

/* one committed FRI layer folds a row of 2^K adjacent (bit-reversed) entries, K times by 2 with
 * beta, beta^2, beta^4, ...: `row_index` is the row's index in the layer matrix of 2^log_rows rows */
static bb4_t fold_row_k(size_t row_index, int log_rows, int K, bb4_t beta, const bb4_t* ev) {
    bb4_t tmp[32];
    size_t cnt = (size_t)1 << K;
    for (size_t j = 0; j < cnt; j++) tmp[j] = ev[j];
    bb4_t b = beta;
    for (int j = 0; j < K; j++) {
        cnt >>= 1;                                   /* pairs at this level: row_index * cnt + t */
        int log_folded = log_rows + (K - 1 - j);
        for (size_t t = 0; t < cnt; t++)
            tmp[t] = fri_fold_row(row_index * cnt + t, log_folded, b, tmp[2 * t], tmp[2 * t + 1]);
        b = bb4_mul(b, b);
    }
    return tmp[0];
}

/* sum_j alpha^j * row[j] over `w` base-field words */
bb4_t orc__row_dot(const bb4_t* pw, const uint32_t* row, size_t w) {
    bb4_t a = bb4_zero();
    for (size_t j = 0; j < w; j++) a = bb4_add(a, bb4_mul_base(pw[j], row[j]));
    return a;
}
void orc__copy_path(uint32_t* pf, size_t* pos, const uint32_t* tree, size_t leaves, size_t index, int levels) {
    const uint32_t* lvl = tree; size_t cnt = leaves, idx = index;
    for (int k = 0; k < levels; k++) { memcpy(pf + *pos, lvl + 8 * (idx ^ 1), 32); *pos += 8; lvl += 8 * cnt; cnt >>= 1; idx >>= 1; }
}

size_t orc_prove_shard(const uint32_t* trace, int log_n, size_t width,
                       const uint32_t* public_values, size_t n_public,
                       const orc_params_t* prm, uint8_t* proof_bytes, size_t cap) {
    shape_t sh;
    if (!shape_of(log_n, width, prm, &sh)) return 0;
    size_t need = orc_proof_size(log_n, width, prm, n_public);
    if (cap < need) return 0;
    uint32_t* pf = (uint32_t*)proof_bytes;
    size_t pos = 0;
    const int lqd = g_air ? g_air_lqd : 1;
    if (lqd > sh.b) return 0;                                              /* the quotient domain must lie inside the committed LDE domain */
    const int H = log_n + sh.b, Hq = log_n + lqd, Q = prm->logup_pairs;    /* LDE domain 2^H, quotient domain 2^Hq */
    const size_t NQ = (size_t)1 << lqd, QW = 4 * NQ;                       /* quotient chunks, width of the quotient matrix */
    const size_t n = (size_t)1 << log_n, m = (size_t)1 << H, mq = (size_t)1 << Hq, wp = Q ? 4 * ((size_t)Q + 1) : 0;

    const size_t CW = g_air ? 0 : (size_t)prm->code_width;
    if (g_air && prm->code_width) return 0;
    pf[pos++] = PROOF_MAGIC; pf[pos++] = g_air ? 7u : (CW ? 8u : (sh.ext ? 3u : (Q ? 2u : PROOF_VERSION))); pf[pos++] = (uint32_t)log_n;
    pf[pos++] = (uint32_t)width; pf[pos++] = (uint32_t)prm->log_blowup;
    pf[pos++] = (uint32_t)prm->num_queries; pf[pos++] = (uint32_t)prm->pow_bits;
    pf[pos++] = (uint32_t)n_public;
    if (sh.ext || g_air) { pf[pos++] = (uint32_t)Q; pf[pos++] = (uint32_t)sh.K; pf[pos++] = (uint32_t)sh.F; pf[pos++] = (uint32_t)sh.hw; }
    else if (Q) pf[pos++] = (uint32_t)Q;
    if (g_air) { orc_air_digest(g_air, g_air_words, pf + pos); pos += 8; }
    if (CW) pf[pos++] = (uint32_t)CW;

    orc_challenger_t ch;
    transcript_init(&ch, log_n, width, prm, n_public, &sh);

    /* 1. commit the trace: LDE on g*<w_2N>, bit-reversed rows, Merkle tree */
    uint32_t* tlde = (uint32_t*)malloc(m * width * 4);
    orc_coset_lde(trace, tlde, log_n, width, sh.b, BB_GEN);
    uint32_t* ttree = (uint32_t*)malloc((2 * m - 1) * 32);
    uint32_t* ctree = NULL;                            /* code group (columns [0, CW)): its own tree, committed and observed first */
    if (CW) {
        uint32_t* part = (uint32_t*)malloc(m * (width - CW > CW ? width - CW : CW) * 4);
        ctree = (uint32_t*)malloc((2 * m - 1) * 32);
        for (size_t r = 0; r < m; r++) memcpy(part + r * CW, tlde + r * width, CW * 4);
        orc_merkle_tree_hw(part, CW, H, ctree, sh.hw);
        memcpy(pf + pos, ctree + (2 * m - 2) * 8, 32); pos += 8;
        orc_chal_observe_slice(&ch, ctree + (2 * m - 2) * 8, 8);
        for (size_t r = 0; r < m; r++) memcpy(part + r * (width - CW), tlde + r * width + CW, (width - CW) * 4);
        orc_merkle_tree_hw(part, width - CW, H, ttree, sh.hw);          /* data group: the remaining columns */
        free(part);
    } else orc_merkle_tree_hw(tlde, width, H, ttree, sh.hw);
    const uint32_t* troot = ttree + (2 * m - 2) * 8;
    memcpy(pf + pos, troot, 32); pos += 8;
    memcpy(g_dbg.trace_root, troot, 32);
    orc_chal_observe_slice(&ch, troot, 8);
    orc_chal_observe_slice(&ch, public_values, n_public);

    /* 1b. LogUp: lookup challenges, permutation trace, its commitment */
    bb4_t gamma = bb4_zero(), beta_l = bb4_zero();
    uint32_t *plde = NULL, *ptree = NULL;
    if (Q) {
        gamma = sample_ext(&ch);
        beta_l = sample_ext(&ch);
        uint32_t* perm = (uint32_t*)malloc(n * wp * 4);
        orc_perm_trace(trace, log_n, width, Q, gamma.c, beta_l.c, perm);
        plde = (uint32_t*)malloc(m * wp * 4);
        orc_coset_lde(perm, plde, log_n, wp, sh.b, BB_GEN);
        free(perm);
        ptree = (uint32_t*)malloc((2 * m - 1) * 32);
        orc_merkle_tree_hw(plde, wp, H, ptree, sh.hw);
        const uint32_t* proot = ptree + (2 * m - 2) * 8;
        memcpy(pf + pos, proot, 32); pos += 8;
        orc_chal_observe_slice(&ch, proot, 8);
    }

    /* 2. constraint challenge, quotient, chunks, commit */
    bb4_t alpha = sample_ext(&ch);
    memcpy(g_dbg.alpha, alpha.c, 16);
    /* The quotient domain g*<w_2N> is the first 2N rows of the bit-reversed LDE on g*<w_{2^H}>, in the
     * bit-reversed order of its own 2N points -- so the blowup-2 routine applies to those rows as is. */
    uint32_t* qv = (uint32_t*)malloc(mq * 16);         /* bit-reversed like the LDE */
    if (g_air) orc_quotient_values_air(g_air, tlde, log_n, width, public_values, alpha.c, lqd, qv);
    else orc_quotient_values_logup(tlde, log_n, width, plde, Q, gamma.c, beta_l.c, alpha.c, qv);
    /* chunk k = natural rows i = 2j + k  <->  bit-reversed rows [k*N, (k+1)*N);
     * as a matrix on the coset (g w^k) * <w_N> in natural order j: */
    uint32_t* qlde = (uint32_t*)malloc(m * QW * 4);    /* [chunk0 | chunk1 | ...], 4 columns per chunk */
    {
        uint32_t* chunk = (uint32_t*)malloc(n * 4 * 4);
        uint32_t* clde = (uint32_t*)malloc(m * 4 * 4);
        bb_t w2n = bb_two_adic_generator(Hq);
        for (size_t k = 0; k < NQ; k++) {
            for (size_t j = 0; j < n; j++) {
                size_t p = bb_reverse_bits((uint32_t)(NQ * j + k), Hq);
                memcpy(chunk + 4 * j, qv + 4 * p, 16);
            }
            /* values on (g w^k)*<w_N> -> LDE on g*<w_{2^H}>: shift = g / (g w^k) */
            bb_t shift = bb_inv(bb_pow(w2n, (uint64_t)k));
            orc_coset_lde(chunk, clde, log_n, 4, sh.b, shift);
            for (size_t r = 0; r < m; r++) memcpy(qlde + r * QW + 4 * k, clde + r * 4, 16);
        }
        free(chunk); free(clde);
    }
    free(qv);
    uint32_t* qtree = (uint32_t*)malloc((2 * m - 1) * 32);
    orc_merkle_tree_hw(qlde, QW, H, qtree, sh.hw);
    const uint32_t* qroot = qtree + (2 * m - 2) * 8;
    memcpy(pf + pos, qroot, 32); pos += 8;
    memcpy(g_dbg.quotient_root, qroot, 32);
    orc_chal_observe_slice(&ch, qroot, 8);

    /* 3. out-of-domain point, openings */
    bb4_t zeta = sample_ext(&ch);
    memcpy(g_dbg.zeta, zeta.c, 16);
    bb4_t zeta_next = bb4_mul_base(zeta, bb_two_adic_generator(log_n));
    uint32_t* op_local = pf + pos; pos += 4 * width;
    uint32_t* op_next = pf + pos; pos += 4 * width;
    uint32_t *op_pl = NULL, *op_pn = NULL;
    if (Q) { op_pl = pf + pos; pos += 4 * wp; op_pn = pf + pos; pos += 4 * wp; }
    uint32_t* op_q = pf + pos; pos += 4 * QW;
    orc_open_at(tlde, log_n, width, zeta.c, op_local);
    orc_open_at(tlde, log_n, width, zeta_next.c, op_next);
    if (Q) { orc_open_at(plde, log_n, wp, zeta.c, op_pl); orc_open_at(plde, log_n, wp, zeta_next.c, op_pn); }
    orc_open_at(qlde, log_n, QW, zeta.c, op_q);
    orc_chal_observe_slice(&ch, op_local, 4 * width);
    orc_chal_observe_slice(&ch, op_next, 4 * width);
    if (Q) { orc_chal_observe_slice(&ch, op_pl, 4 * wp); orc_chal_observe_slice(&ch, op_pn, 4 * wp); }
    orc_chal_observe_slice(&ch, op_q, 4 * QW);

    /* 4. FRI input: alpha-batched reduced openings at every LDE point.
     * batching order (offsets in powers of the FRI alpha): trace@zeta 0, trace@zeta_next W,
     * [perm@zeta 2W, perm@zeta_next 2W+Wp], quotient@zeta 2W+2Wp */
    bb4_t fa = sample_ext(&ch);
    memcpy(g_dbg.fri_alpha, fa.c, 16);
    size_t np = width > QW ? width : QW;
    if (wp > np) np = wp;
    bb4_t* fapow = (bb4_t*)malloc(np * sizeof(bb4_t));
    fapow[0] = bb4_one();
    for (size_t j = 1; j < np; j++) fapow[j] = bb4_mul(fapow[j - 1], fa);
    bb4_t y_loc = bb4_zero(), y_nxt = bb4_zero(), y_pl = bb4_zero(), y_pn = bb4_zero(), y_q = bb4_zero();
    for (size_t j = 0; j < width; j++) {
        y_loc = bb4_add(y_loc, bb4_mul(fapow[j], ld4(op_local + 4 * j)));
        y_nxt = bb4_add(y_nxt, bb4_mul(fapow[j], ld4(op_next + 4 * j)));
    }
    for (size_t j = 0; j < wp; j++) {
        y_pl = bb4_add(y_pl, bb4_mul(fapow[j], ld4(op_pl + 4 * j)));
        y_pn = bb4_add(y_pn, bb4_mul(fapow[j], ld4(op_pn + 4 * j)));
    }
    for (size_t j = 0; j < QW; j++) y_q = bb4_add(y_q, bb4_mul(fapow[j], ld4(op_q + 4 * j)));
    bb4_t off_next = bb4_pow(fa, width), off_pl = bb4_pow(fa, 2 * width), off_pn = bb4_pow(fa, 2 * width + wp),
          off_q = bb4_pow(fa, 2 * width + 2 * wp);
    bb4_t* cur = (bb4_t*)malloc(m * sizeof(bb4_t));
    {
        bb_t wm = bb_two_adic_generator(H);
#pragma omp parallel for schedule(static)
        for (size_t p = 0; p < m; p++) {
            bb_t x = bb_mul(BB_GEN, bb_pow(wm, bb_reverse_bits((uint32_t)p, H)));
            bb4_t d1 = bb4_inv(bb4_neg(bb4_sub_base(zeta, x)));        /* 1/(x - zeta) */
            bb4_t d2 = bb4_inv(bb4_neg(bb4_sub_base(zeta_next, x)));
            bb4_t at = row_dot(fapow, tlde + p * width, width);
            bb4_t aq = row_dot(fapow, qlde + p * QW, QW);
            bb4_t r = bb4_mul(bb4_sub(at, y_loc), d1);
            r = bb4_add(r, bb4_mul(off_next, bb4_mul(bb4_sub(at, y_nxt), d2)));
            if (Q) {
                bb4_t ap = row_dot(fapow, plde + p * wp, wp);
                r = bb4_add(r, bb4_mul(off_pl, bb4_mul(bb4_sub(ap, y_pl), d1)));
                r = bb4_add(r, bb4_mul(off_pn, bb4_mul(bb4_sub(ap, y_pn), d2)));
            }
            r = bb4_add(r, bb4_mul(off_q, bb4_mul(bb4_sub(aq, y_q), d1)));
            cur[p] = r;
        }
    }
    free(fapow);

    /* 5. FRI commit phase: R committed layers, each a matrix of rows of 2^K adjacent entries */
    const int R = sh.R, K = sh.K;
    const size_t arity = (size_t)1 << K;
    bb4_t** layers = (bb4_t**)malloc((R ? R : 1) * sizeof(bb4_t*));
    uint32_t** ltrees = (uint32_t**)malloc((R ? R : 1) * sizeof(uint32_t*));
    uint32_t* commits = pf + pos; pos += 8 * (size_t)R;
    for (int l = 0; l < R; l++) {
        int lh = H - K * (l + 1);                /* log rows of this layer's matrix */
        size_t rows = (size_t)1 << lh;
        layers[l] = cur;
        ltrees[l] = (uint32_t*)malloc((2 * rows - 1) * 32);
        orc_merkle_tree_hw((const uint32_t*)cur, 4 * arity, lh, ltrees[l], sh.hw);
        const uint32_t* root = ltrees[l] + (2 * rows - 2) * 8;
        memcpy(commits + 8 * l, root, 32);
        orc_chal_observe_slice(&ch, root, 8);
        bb4_t beta = sample_ext(&ch);
        bb4_t* nxt = (bb4_t*)malloc(rows * sizeof(bb4_t));
        if (K == 1) orc_fri_fold((const uint32_t*)cur, lh + 1, beta.c, (uint32_t*)nxt);
        else {
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < rows; i++) nxt[i] = fold_row_k(i, lh, K, beta, cur + i * arity);
        }
        cur = nxt;
    }
    /* 2^(F+b) evaluations (bit-reversed, on <w_{2^(F+b)}>) of a polynomial of < 2^F coefficients remain:
     * interpolate and send the coefficients */
    int const_ok = 1;
    {
        const int lf = sh.F + sh.b;
        const size_t nf = (size_t)1 << lf, keep = (size_t)1 << sh.F;
        uint32_t* ev = (uint32_t*)malloc(nf * 16);
        for (size_t i = 0; i < nf; i++) st4(ev + 4 * i, cur[bb_reverse_bits((uint32_t)i, lf)]);
        orc_ntt(ev, lf, 4, 1);                   /* the DFT is F_p-linear: the 4 coordinates transform separately */
        for (size_t i = keep; i < nf; i++)
            for (int e = 0; e < 4; e++) if (ev[4 * i + e] != 0) const_ok = 0;
        memcpy(pf + pos, ev, keep * 16);
        orc_chal_observe_slice(&ch, ev, 4 * keep);
        pos += 4 * keep;
        free(ev);
    }
    free(cur);

    /* 6. proof of work, queries */
    uint32_t witness = orc_chal_grind(&ch, prm->pow_bits);
    g_dbg.pow_witness = witness;
    pf[pos++] = witness;
    for (int q = 0; q < prm->num_queries; q++) {
        size_t index = orc_chal_sample_bits(&ch, H);
        memcpy(pf + pos, tlde + index * width, width * 4); pos += width;
        if (CW) copy_path(pf, &pos, ctree, m, index, H);
        copy_path(pf, &pos, ttree, m, index, H);
        if (Q) { memcpy(pf + pos, plde + index * wp, wp * 4); pos += wp; copy_path(pf, &pos, ptree, m, index, H); }
        memcpy(pf + pos, qlde + index * QW, QW * 4); pos += QW;
        copy_path(pf, &pos, qtree, m, index, H);
        size_t idx = index;
        for (int l = 0; l < R; l++) {
            int lh = H - K * (l + 1);
            size_t row = idx >> K, own = idx & (arity - 1);
            for (size_t j = 0; j < arity; j++)
                if (j != own) { st4(pf + pos, layers[l][row * arity + j]); pos += 4; }
            copy_path(pf, &pos, ltrees[l], (size_t)1 << lh, row, lh);
            idx = row;
        }
    }
    for (int l = 0; l < R; l++) { free(layers[l]); free(ltrees[l]); }
    free(ctree);
    free(layers); free(ltrees); free(tlde); free(ttree); free(qlde); free(qtree); free(plde); free(ptree);
    if (!const_ok) return 0;
    return pos * 4 == need ? need : 0;
}

/* ------------------------------------------------------------------ */
/* verifier                                                             */
/* ------------------------------------------------------------------ */
/* value at zeta of an extension column committed as 4 base columns: sum_e x^e * v_e(zeta) */
bb4_t orc__recombine(const uint32_t* opened4) {
    bb4_t r = bb4_zero();
    for (int e = 0; e < 4; e++) {
        bb4_t basis = bb4_zero(); basis.c[e] = 1;
        r = bb4_add(r, bb4_mul(basis, ld4(opened4 + 4 * e)));
    }
    return r;
}

int orc_verify_shard(const uint8_t* proof_bytes, size_t len, int log_n, size_t width,
                     const uint32_t* public_values, size_t n_public,
                     const orc_params_t* prm) {
    shape_t sh;
    if (!shape_of(log_n, width, prm, &sh)) return 1;
    if (len != orc_proof_size(log_n, width, prm, n_public)) return 2;
    const uint32_t* pf = (const uint32_t*)proof_bytes;
    size_t pos = 0;
    const int lqd = g_air ? g_air_lqd : 1;
    if (lqd > sh.b) return 1;
    const size_t NQ = (size_t)1 << lqd, QW = 4 * NQ;
    const int H = log_n + sh.b, Hq = log_n + lqd, R = sh.R, K = sh.K, Q = prm->logup_pairs;
    const size_t n = (size_t)1 << log_n, wp = Q ? 4 * ((size_t)Q + 1) : 0, arity = (size_t)1 << K;
    const size_t CW = g_air ? 0 : (size_t)prm->code_width;
    if (g_air && prm->code_width) return 1;
    if (pf[0] != PROOF_MAGIC || pf[1] != (g_air ? 7u : (CW ? 8u : (sh.ext ? 3u : (Q ? 2u : PROOF_VERSION)))) || pf[2] != (uint32_t)log_n ||
        pf[3] != (uint32_t)width || pf[4] != (uint32_t)prm->log_blowup ||
        pf[5] != (uint32_t)prm->num_queries || pf[6] != (uint32_t)prm->pow_bits ||
        pf[7] != (uint32_t)n_public) return 3;
    pos = 8;
    if (sh.ext || g_air) {
        if (pf[8] != (uint32_t)Q || pf[9] != (uint32_t)sh.K || pf[10] != (uint32_t)sh.F || pf[11] != (uint32_t)sh.hw) return 3;
        pos = 12;
    } else if (Q) { if (pf[8] != (uint32_t)Q) return 3; pos = 9; }
    if (g_air) {
        uint32_t dg[8];
        orc_air_digest(g_air, g_air_words, dg);
        if (memcmp(dg, pf + pos, 32) != 0) return 3;
        pos += 8;
    }
    if (CW) { if (pf[pos] != (uint32_t)CW) return 3; pos++; }
    for (size_t i = pos; i < len / 4; i++) if (pf[i] >= BB_P) return 4;   /* canonical words only */

    orc_challenger_t ch;
    transcript_init(&ch, log_n, width, prm, n_public, &sh);
    const uint32_t* croot = NULL;
    if (CW) { croot = pf + pos; pos += 8; orc_chal_observe_slice(&ch, croot, 8); }
    const uint32_t* troot = pf + pos; pos += 8;
    orc_chal_observe_slice(&ch, troot, 8);
    orc_chal_observe_slice(&ch, public_values, n_public);
    bb4_t gamma = bb4_zero(), beta_l = bb4_zero();
    const uint32_t* proot = NULL;
    if (Q) {
        gamma = sample_ext(&ch);
        beta_l = sample_ext(&ch);
        proot = pf + pos; pos += 8;
        orc_chal_observe_slice(&ch, proot, 8);
    }
    const uint32_t* qroot = pf + pos; pos += 8;
    bb4_t alpha = sample_ext(&ch);
    orc_chal_observe_slice(&ch, qroot, 8);
    bb4_t zeta = sample_ext(&ch);
    bb_t gn = bb_two_adic_generator(log_n);
    bb4_t zeta_next = bb4_mul_base(zeta, gn);
    const uint32_t* op_local = pf + pos; pos += 4 * width;
    const uint32_t* op_next = pf + pos; pos += 4 * width;
    const uint32_t *op_pl = NULL, *op_pn = NULL;
    if (Q) { op_pl = pf + pos; pos += 4 * wp; op_pn = pf + pos; pos += 4 * wp; }
    const uint32_t* op_q = pf + pos; pos += 4 * QW;
    orc_chal_observe_slice(&ch, op_local, 4 * width);
    orc_chal_observe_slice(&ch, op_next, 4 * width);
    if (Q) { orc_chal_observe_slice(&ch, op_pl, 4 * wp); orc_chal_observe_slice(&ch, op_pn, 4 * wp); }
    orc_chal_observe_slice(&ch, op_q, 4 * QW);

    /* (a) constraint check at zeta */
    {
        bb4_t* loc = (bb4_t*)malloc(width * sizeof(bb4_t));
        bb4_t* nxt = (bb4_t*)malloc(width * sizeof(bb4_t));
        for (size_t j = 0; j < width; j++) { loc[j] = ld4(op_local + 4 * j); nxt[j] = ld4(op_next + 4 * j); }
        bb4_t zn = bb4_pow(zeta, n);
        bb4_t zh = bb4_sub_base(zn, 1);
        bb4_t sel_first = bb4_mul(zh, bb4_inv(bb4_sub_base(zeta, 1)));
        bb4_t sel_trans = bb4_sub_base(zeta, bb_inv(gn));
        bb4_t folded = g_air ? orc__air_fold_ext(g_air, loc, nxt, public_values, sel_first, bb4_mul(zh, bb4_inv(bb4_sub_base(zeta, bb_inv(gn)))), sel_trans, alpha)
                             : fold_constraints_ext(loc, nxt, width, sel_first, sel_trans, alpha);
        if (Q) {
            bb4_t sel_last = bb4_mul(zh, bb4_inv(bb4_sub_base(zeta, bb_inv(gn))));
            bb4_t as[64], bs[64], ar[64], br[64], pl[65], pn[65];
            for (int q = 0; q < Q; q++) { as[q] = loc[8 * q]; bs[q] = loc[8 * q + 1]; ar[q] = loc[8 * q + 4]; br[q] = loc[8 * q + 5]; }
            for (int q = 0; q <= Q; q++) { pl[q] = recombine(op_pl + 16 * q); pn[q] = recombine(op_pn + 16 * q); }
            folded = fold_logup(folded, Q, as, bs, ar, br, pl, pn, gamma, beta_l, sel_first, sel_trans, sel_last, alpha, bb4_zero());
        }
        free(loc); free(nxt);
        /* quotient(zeta) = sum_k zps_k(zeta) * q_k(zeta); chunk k lives on the coset s_k <w_N>, s_k = g w_{2^Hq}^k, and
         * zps_k = prod_{j != k} Z_Dj(zeta) / Z_Dj(s_k) with Z_Dj(x) = (x / s_j)^N - 1 vanishes on every other chunk's coset */
        bb_t wq = bb_two_adic_generator(Hq);
        bb_t sN[4];
        for (size_t k = 0; k < NQ; k++) sN[k] = bb_pow(bb_mul(BB_GEN, bb_pow(wq, k)), n);      /* s_k^N */
        bb4_t quot = bb4_zero();
        for (size_t k = 0; k < NQ; k++) {
            bb4_t zps = bb4_one();
            for (size_t j = 0; j < NQ; j++) {
                if (j == k) continue;
                bb_t sjn_inv = bb_inv(sN[j]);
                bb4_t num = bb4_sub_base(bb4_mul_base(zn, sjn_inv), 1);                 /* Z_Dj(zeta) */
                bb_t den = bb_sub(bb_mul(sN[k], sjn_inv), 1);                           /* Z_Dj(s_k)  */
                zps = bb4_mul(zps, bb4_mul_base(num, bb_inv(den)));
            }
            quot = bb4_add(quot, bb4_mul(zps, recombine(op_q + 16 * k)));
        }
        if (!bb4_eq(bb4_mul(folded, bb4_inv(zh)), quot)) return 10;
    }

    /* (b) FRI */
    bb4_t fa = sample_ext(&ch);
    size_t np = width > QW ? width : QW;
    if (wp > np) np = wp;
    bb4_t* fapow = (bb4_t*)malloc(np * sizeof(bb4_t));
    fapow[0] = bb4_one();
    for (size_t j = 1; j < np; j++) fapow[j] = bb4_mul(fapow[j - 1], fa);
    bb4_t y_loc = bb4_zero(), y_nxt = bb4_zero(), y_pl = bb4_zero(), y_pn = bb4_zero(), y_q = bb4_zero();
    for (size_t j = 0; j < width; j++) {
        y_loc = bb4_add(y_loc, bb4_mul(fapow[j], ld4(op_local + 4 * j)));
        y_nxt = bb4_add(y_nxt, bb4_mul(fapow[j], ld4(op_next + 4 * j)));
    }
    for (size_t j = 0; j < wp; j++) {
        y_pl = bb4_add(y_pl, bb4_mul(fapow[j], ld4(op_pl + 4 * j)));
        y_pn = bb4_add(y_pn, bb4_mul(fapow[j], ld4(op_pn + 4 * j)));
    }
    for (size_t j = 0; j < QW; j++) y_q = bb4_add(y_q, bb4_mul(fapow[j], ld4(op_q + 4 * j)));
    bb4_t off_next = bb4_pow(fa, width), off_pl = bb4_pow(fa, 2 * width), off_pn = bb4_pow(fa, 2 * width + wp),
          off_q = bb4_pow(fa, 2 * width + 2 * wp);

    const uint32_t* commits = pf + pos; pos += 8 * (size_t)R;
    bb4_t* betas = (bb4_t*)malloc((R ? R : 1) * sizeof(bb4_t));
    for (int l = 0; l < R; l++) {
        orc_chal_observe_slice(&ch, commits + 8 * l, 8);
        betas[l] = sample_ext(&ch);
    }
    const size_t keep = (size_t)1 << sh.F;
    const uint32_t* final_poly = pf + pos; pos += 4 * keep;     /* coefficients, lowest first */
    orc_chal_observe_slice(&ch, final_poly, 4 * keep);
    uint32_t witness = pf[pos++];
    int rc = 0;
    if (!orc_chal_check_witness(&ch, prm->pow_bits, witness)) rc = 20;

    bb_t wm = bb_two_adic_generator(H);
    for (int q = 0; q < prm->num_queries && rc == 0; q++) {
        size_t index = orc_chal_sample_bits(&ch, H);
        const uint32_t* trow = pf + pos; pos += width;
        const uint32_t* cpath = NULL;
        if (CW) { cpath = pf + pos; pos += 8 * (size_t)H; }
        const uint32_t* tpath = pf + pos; pos += 8 * (size_t)H;
        const uint32_t *prow = NULL, *ppath = NULL;
        if (Q) { prow = pf + pos; pos += wp; ppath = pf + pos; pos += 8 * (size_t)H; }
        const uint32_t* qrow = pf + pos; pos += QW;
        const uint32_t* qpath = pf + pos; pos += 8 * (size_t)H;
        if (CW && orc_merkle_verify_hw(croot, H, index, trow, CW, cpath, sh.hw)) { rc = 33; break; }
        if (orc_merkle_verify_hw(troot, H, index, trow + CW, width - CW, tpath, sh.hw)) { rc = 30; break; }
        if (Q && orc_merkle_verify_hw(proot, H, index, prow, wp, ppath, sh.hw)) { rc = 32; break; }
        if (orc_merkle_verify_hw(qroot, H, index, qrow, QW, qpath, sh.hw)) { rc = 31; break; }
        bb_t x = bb_mul(BB_GEN, bb_pow(wm, bb_reverse_bits((uint32_t)index, H)));
        bb4_t d1 = bb4_inv(bb4_neg(bb4_sub_base(zeta, x)));
        bb4_t d2 = bb4_inv(bb4_neg(bb4_sub_base(zeta_next, x)));
        bb4_t at = row_dot(fapow, trow, width), aq = row_dot(fapow, qrow, QW);
        bb4_t ro = bb4_mul(bb4_sub(at, y_loc), d1);
        ro = bb4_add(ro, bb4_mul(off_next, bb4_mul(bb4_sub(at, y_nxt), d2)));
        if (Q) {
            bb4_t ap = row_dot(fapow, prow, wp);
            ro = bb4_add(ro, bb4_mul(off_pl, bb4_mul(bb4_sub(ap, y_pl), d1)));
            ro = bb4_add(ro, bb4_mul(off_pn, bb4_mul(bb4_sub(ap, y_pn), d2)));
        }
        ro = bb4_add(ro, bb4_mul(off_q, bb4_mul(bb4_sub(aq, y_q), d1)));

        bb4_t folded = ro;          /* single height: the reduced opening enters at layer 0 */
        size_t idx = index;
        for (int l = 0; l < R; l++) {
            int lh = H - K * (l + 1);
            size_t row = idx >> K, own = idx & (arity - 1);
            bb4_t ev[32];
            for (size_t j = 0; j < arity; j++) {
                if (j == own) ev[j] = folded;
                else { ev[j] = ld4(pf + pos); pos += 4; }
            }
            const uint32_t* path = pf + pos; pos += 8 * (size_t)lh;
            uint32_t rowbuf[4 * 32];
            for (size_t j = 0; j < arity; j++) memcpy(rowbuf + 4 * j, ev[j].c, 16);
            if (orc_merkle_verify_hw(commits + 8 * l, lh, row, rowbuf, 4 * arity, path, sh.hw)) { rc = 40 + (l < 50 ? l : 50); break; }
            folded = fold_row_k(row, lh, K, betas[l], ev);
            idx = row;
        }
        if (rc) break;
        /* the final polynomial at this query's point of the last domain <w_{2^(F+b)}> (Horner) */
        {
            const int lf = sh.F + sh.b;
            bb_t xf = bb_pow(bb_two_adic_generator(lf), bb_reverse_bits((uint32_t)idx, lf));
            bb4_t v = bb4_zero();
            for (size_t i = keep; i-- > 0;) v = bb4_add(bb4_mul_base(v, xf), ld4(final_poly + 4 * i));
            if (!bb4_eq(folded, v)) { rc = 100; break; }
        }
    }
    free(fapow); free(betas);
    if (rc == 0 && pos * 4 != len) rc = 5;
    return rc;
}

/* ------------------------------------------------------------------ */
/* the same prover / verifier with the AIR supplied as a constraint program (oracle/air.c); proof version 7 */
/* ------------------------------------------------------------------ */
size_t orc_proof_size_air(int log_n, size_t width, const orc_params_t* prm, size_t n_public, int log_quotient_degree) {
    static const uint32_t marker[1] = {0};
    const uint32_t* saved = g_air;
    const int saved_lqd = g_air_lqd;
    g_air = marker; g_air_lqd = log_quotient_degree;
    size_t r = (prm->logup_pairs || log_quotient_degree < 1 || log_quotient_degree > 2 || log_quotient_degree > prm->log_blowup) ? 0 : orc_proof_size(log_n, width, prm, n_public);
    g_air = saved; g_air_lqd = saved_lqd;
    return r;
}
size_t orc_prove_shard_air(const uint32_t* prog, size_t prog_words, const uint32_t* trace, int log_n, size_t width,
                           const uint32_t* public_values, size_t n_public, const orc_params_t* prm, uint8_t* proof_bytes, size_t cap) {
    if (prm->logup_pairs || !orc_air_validate(prog, prog_words, width, n_public)) return 0;
    g_air = prog; g_air_words = prog_words; g_air_lqd = orc_air_log_quotient_degree(prog);
    size_t r = orc_prove_shard(trace, log_n, width, public_values, n_public, prm, proof_bytes, cap);
    g_air = NULL; g_air_words = 0;
    return r;
}
int orc_verify_shard_air(const uint32_t* prog, size_t prog_words, const uint8_t* proof_bytes, size_t len, int log_n, size_t width,
                         const uint32_t* public_values, size_t n_public, const orc_params_t* prm) {
    if (prm->logup_pairs || !orc_air_validate(prog, prog_words, width, n_public)) return 1;
    g_air = prog; g_air_words = prog_words; g_air_lqd = orc_air_log_quotient_degree(prog);
    int r = orc_verify_shard(proof_bytes, len, log_n, width, public_values, n_public, prm);
    g_air = NULL; g_air_words = 0;
    return r;
}
