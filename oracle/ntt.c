/*
 * oracle/ntt.c -- radix-2 NTT and coset low-degree extension, CPU restatement.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle/oracle.h).
 *
 * Restates the published algorithm behind p3-dft 0.2.1-succinct (reference
 * Cargo.lock:3903; reached from crates/guest-prover-sp1/src/sp1.rs:116 via
 * sp1-stark `commit` -> p3-fri TwoAdicFriPcs::commit -> coset_lde_batch):
 *   dft_batch:        X[k] = sum_j x[j] w_N^(jk) over every column, natural order;
 *   coset_lde_batch:  c = idft(x); c[j] *= shift^j; zero-pad to N*2^b; dft;
 *   bit_reverse_rows: row bitrev(i) of the result holds evaluation index i.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

uint32_t orc_bb_mul(uint32_t a, uint32_t b) { return bb_mul(a, b); }
uint32_t orc_bb_inv(uint32_t a) { return bb_inv(a); }
uint32_t orc_bb_pow(uint32_t a, uint64_t e) { return bb_pow(a, e); }
uint32_t orc_two_adic_generator(int bits) { return bb_two_adic_generator(bits); }
void orc_bb4_mul(const uint32_t a[4], const uint32_t b[4], uint32_t out[4]) {
    bb4_t x, y; memcpy(x.c, a, 16); memcpy(y.c, b, 16);
    bb4_t r = bb4_mul(x, y); memcpy(out, r.c, 16);
}
void orc_bb4_inv(const uint32_t a[4], uint32_t out[4]) {
    bb4_t x; memcpy(x.c, a, 16);
    bb4_t r = bb4_inv(x); memcpy(out, r.c, 16);
}
void orc_to_monty(const uint32_t* in, uint32_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = bb_to_monty(in[i]);
}
void orc_from_monty(const uint32_t* in, uint32_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = bb_from_monty(in[i]);
}

void orc_dft_naive(const uint32_t* in, uint32_t* out, int log_n, size_t width, int inverse) {
    size_t n = (size_t)1 << log_n;
    bb_t w = bb_two_adic_generator(log_n);
    if (inverse) w = bb_inv(w);
    bb_t ninv = inverse ? bb_inv((bb_t)(n % BB_P)) : 1;
    for (size_t k = 0; k < n; k++) {
        bb_t wk = bb_pow(w, k);
        for (size_t c = 0; c < width; c++) {
            bb_t acc = 0, t = 1;
            for (size_t j = 0; j < n; j++) {
                acc = bb_add(acc, bb_mul(in[j * width + c], t));
                t = bb_mul(t, wk);
            }
            out[k * width + c] = bb_mul(acc, ninv);
        }
    }
}

static void bit_reverse_rows_inplace(uint32_t* a, int log_n, size_t width) {
    size_t n = (size_t)1 << log_n;
#pragma omp parallel
    {
        uint32_t* tmp = (uint32_t*)malloc(width * sizeof(uint32_t));
#pragma omp for schedule(static)
        for (size_t i = 0; i < n; i++) {                /* (i, j = rev(i)) with i < j: every pair is swapped by exactly one iteration */
            size_t j = bb_reverse_bits((uint32_t)i, log_n);
            if (i < j) {
                memcpy(tmp, a + i * width, width * 4);
                memcpy(a + i * width, a + j * width, width * 4);
                memcpy(a + j * width, tmp, width * 4);
            }
        }
        free(tmp);
    }
}

/* x * w mod p for x, w < p without a division, so that the loops over a row vectorise: x w < 2^62,
 * q = ((x w >> 30) mu) >> 32 with mu = floor(2^62 / p) is floor(x w / p) or up to 2 less, so x w - q p < 3 p.
 * Same value as bb_mul (bb.h: `%`); tests/test_oracle.py holds both against the naive DFT. */
static inline uint32_t mul_row(uint32_t x, uint32_t w) {
    const uint64_t xw = (uint64_t)x * w;
    const uint64_t q = ((xw >> 30) * 2290649223ull) >> 32;
    uint32_t r = (uint32_t)(xw - q * BB_P);             /* < 3 p < 2^32 */
    r = r >= BB_P ? r - BB_P : r;
    return r >= BB_P ? r - BB_P : r;
}

/* decimation-in-time: bit-reverse the rows, then log_n layers of butterflies
 * (a, b) -> (a + w b, a - w b); natural order in, natural order out. */
void orc_ntt(uint32_t* a, int log_n, size_t width, int inverse) {
    size_t n = (size_t)1 << log_n;
    if (log_n == 0) return;
    bit_reverse_rows_inplace(a, log_n, width);
    bb_t root = bb_two_adic_generator(log_n);
    if (inverse) root = bb_inv(root);
    /* twiddles w^0 .. w^(n/2-1) */
    bb_t* tw = (bb_t*)malloc((n / 2) * sizeof(bb_t));
    tw[0] = 1;
    for (size_t i = 1; i < n / 2; i++) tw[i] = bb_mul(tw[i - 1], root);
    int layer = 0;
    /* two layers per sweep over the matrix (the same butterflies in the same order per element: rows i, i + h of layer l, then
     * rows i, i + 2h of layer l + 1, held in registers in between) -- a full-size LDE is bound by memory, not by arithmetic */
    for (; layer + 1 < log_n; layer += 2) {
        size_t half = (size_t)1 << layer;
        size_t step = n / (2 * half);
#pragma omp parallel for schedule(static)
        for (size_t q = 0; q < n / 4; q++) {
            size_t grp = q / half, j = q % half;
            size_t i0 = grp * 4 * half + j;
            const uint32_t w1 = tw[j * step], w2 = tw[j * (step / 2)], w3 = tw[(j + half) * (step / 2)];
            uint32_t* restrict r0 = a + i0 * width;
            uint32_t* restrict r1 = a + (i0 + half) * width;
            uint32_t* restrict r2 = a + (i0 + 2 * half) * width;
            uint32_t* restrict r3 = a + (i0 + 3 * half) * width;
            for (size_t c = 0; c < width; c++) {
                uint32_t t, s, d;
                /* layer l: (r0, r1) and (r2, r3), twiddle w1 */
                t = mul_row(r1[c], w1); s = r0[c] + t; d = r0[c] + (BB_P - t);
                const uint32_t a0 = s >= BB_P ? s - BB_P : s, a1 = d >= BB_P ? d - BB_P : d;
                t = mul_row(r3[c], w1); s = r2[c] + t; d = r2[c] + (BB_P - t);
                const uint32_t a2 = s >= BB_P ? s - BB_P : s, a3 = d >= BB_P ? d - BB_P : d;
                /* layer l + 1: (a0, a2) with w2, (a1, a3) with w3 */
                t = mul_row(a2, w2); s = a0 + t; d = a0 + (BB_P - t);
                r0[c] = s >= BB_P ? s - BB_P : s; r2[c] = d >= BB_P ? d - BB_P : d;
                t = mul_row(a3, w3); s = a1 + t; d = a1 + (BB_P - t);
                r1[c] = s >= BB_P ? s - BB_P : s; r3[c] = d >= BB_P ? d - BB_P : d;
            }
        }
    }
    for (; layer < log_n; layer++) {
        size_t half = (size_t)1 << layer;        /* butterfly span */
        size_t step = n / (2 * half);            /* twiddle stride */
#pragma omp parallel for schedule(static)
        for (size_t bf = 0; bf < n / 2; bf++) {
            size_t grp = bf / half, j = bf % half;
            size_t i0 = grp * 2 * half + j, i1 = i0 + half;
            const uint32_t w = tw[j * step];
            uint32_t* restrict r0 = a + i0 * width;
            uint32_t* restrict r1 = a + i1 * width;
            for (size_t c = 0; c < width; c++) {
                const uint32_t t = mul_row(r1[c], w);
                const uint32_t u = r0[c];
                const uint32_t s = u + t, d = u + (BB_P - t);
                r0[c] = s >= BB_P ? s - BB_P : s;
                r1[c] = d >= BB_P ? d - BB_P : d;
            }
        }
    }
    free(tw);
    if (inverse) {
        const uint32_t ninv = bb_inv((bb_t)(n % BB_P));
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n * width; i++) a[i] = mul_row(a[i], ninv);
    }
}

void orc_coset_lde(const uint32_t* in, uint32_t* out, int log_n, size_t width,
                   int log_blowup, uint32_t shift) {
    size_t n = (size_t)1 << log_n;
    int log_m = log_n + log_blowup;
    size_t m = (size_t)1 << log_m;
    /* natural-order buffer of the extended size */
    uint32_t* buf = (uint32_t*)calloc(m * width, sizeof(uint32_t));
    memcpy(buf, in, n * width * sizeof(uint32_t));
    orc_ntt(buf, log_n, width, 1);                 /* coefficients c_j, j < n */
    bb_t* sp = (bb_t*)malloc(n * sizeof(bb_t));    /* shift^j */
    sp[0] = 1;
    for (size_t j = 1; j < n; j++) sp[j] = bb_mul(sp[j - 1], shift);
#pragma omp parallel for schedule(static)
    for (size_t j = 0; j < n; j++) {               /* c_j *= shift^j */
        const uint32_t s = sp[j];
        uint32_t* restrict row = buf + j * width;
        for (size_t c = 0; c < width; c++) row[c] = mul_row(row[c], s);
    }
    free(sp);
    /* rows n..m-1 are already zero */
    orc_ntt(buf, log_m, width, 0);                 /* evaluations on shift * <w_m> */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < m; i++) {
        size_t r = bb_reverse_bits((uint32_t)i, log_m);
        memcpy(out + r * width, buf + i * width, width * sizeof(uint32_t));
    }
    free(buf);
}
