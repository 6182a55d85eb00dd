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
    uint32_t* tmp = (uint32_t*)malloc(width * sizeof(uint32_t));
    for (size_t i = 0; i < n; i++) {
        size_t j = bb_reverse_bits((uint32_t)i, log_n);
        if (i < j) {
            memcpy(tmp, a + i * width, width * 4);
            memcpy(a + i * width, a + j * width, width * 4);
            memcpy(a + j * width, tmp, width * 4);
        }
    }
    free(tmp);
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
    for (int layer = 0; layer < log_n; layer++) {
        size_t half = (size_t)1 << layer;        /* butterfly span */
        size_t step = n / (2 * half);            /* twiddle stride */
#pragma omp parallel for schedule(static)
        for (size_t bf = 0; bf < n / 2; bf++) {
            size_t grp = bf / half, j = bf % half;
            size_t i0 = grp * 2 * half + j, i1 = i0 + half;
            bb_t w = tw[j * step];
            uint32_t* r0 = a + i0 * width;
            uint32_t* r1 = a + i1 * width;
            for (size_t c = 0; c < width; c++) {
                bb_t t = bb_mul(r1[c], w);
                bb_t u = r0[c];
                r0[c] = bb_add(u, t);
                r1[c] = bb_sub(u, t);
            }
        }
    }
    free(tw);
    if (inverse) {
        bb_t ninv = bb_inv((bb_t)(n % BB_P));
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n * width; i++) a[i] = bb_mul(a[i], ninv);
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
    bb_t s = 1;
    for (size_t j = 0; j < n; j++) {               /* c_j *= shift^j */
        for (size_t c = 0; c < width; c++) buf[j * width + c] = bb_mul(buf[j * width + c], s);
        s = bb_mul(s, shift);
    }
    /* rows n..m-1 are already zero */
    orc_ntt(buf, log_m, width, 0);                 /* evaluations on shift * <w_m> */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < m; i++) {
        size_t r = bb_reverse_bits((uint32_t)i, log_m);
        memcpy(out + r * width, buf + i * width, width * sizeof(uint32_t));
    }
    free(buf);
}
