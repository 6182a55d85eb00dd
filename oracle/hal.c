/*
 * oracle/hal.c -- CPU restatement of the RISC Zero `Hal` operator set the HIP path offers at operator level
 * (SURVEY.md 8a row a11 / section 2.3).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/bb.h).  PARITY UNPINNED: the operators live in the un-vendored crates
 * risc0-zkp 1.2.5 (trait hal::Hal, reference Cargo.lock:5057) and risc0-sys 1.2.5 (CUDA / C++ kernels, Cargo.lock:5045),
 * reached from crates/guest-prover-r0/src/prover.rs:90 (`prove_with_opts`); the reference holds no vector for them.  Each
 * function restates the operator's published meaning ([RECALLED] in SURVEY.md section 2.3) in canonical arithmetic:
 *   eltwise_add_elem, eltwise_sum_extelem, eltwise_copy_elem, eltwise_zeroize_elem, zk_shift, mix_poly_coeffs,
 *   batch_evaluate_any, gather_sample, scatter, prefix_products, hash_rows / hash_fold with SHA-256.
 * Layout: polynomials / columns are contiguous vectors, matrices column-major [count][size]; extension elements are 4
 * consecutive words over x^4 = ext_w, ext_w = 11 (SP1 / Plonky3) or p - 11 (RISC Zero's x^4 + 11).
 */
#include <string.h>

#include "oracle.h"

static bb4_t ext_mul_w(bb4_t a, bb4_t b, bb_t w) {
    uint64_t t[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) t[i + j] = (t[i + j] + (uint64_t)bb_mul(a.c[i], b.c[j])) % BB_P;
    bb4_t r;
    for (int i = 0; i < 4; i++) {
        uint64_t v = t[i];
        if (i + 4 < 7) v = (v + (uint64_t)w * t[i + 4]) % BB_P;
        r.c[i] = (bb_t)v;
    }
    return r;
}
static bb4_t ld4(const uint32_t* p) { bb4_t r = {{p[0], p[1], p[2], p[3]}}; return r; }
static void st4(uint32_t* p, bb4_t v) { for (int i = 0; i < 4; i++) p[i] = v.c[i]; }

void orc_hal_ext_mul(const uint32_t a[4], const uint32_t b[4], uint32_t ext_w, uint32_t out[4]) { st4(out, ext_mul_w(ld4(a), ld4(b), ext_w)); }

void orc_hal_eltwise_add(uint32_t* out, const uint32_t* a, const uint32_t* b, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = bb_add(a[i], b[i]);
}
/* out[i] = sum_j in[j * count + i] over extension elements, i < count, j < to_add */
void orc_hal_eltwise_sum_ext(uint32_t* out, const uint32_t* in, size_t count, size_t to_add) {
    for (size_t i = 0; i < count; i++) {
        bb4_t acc = bb4_zero();
        for (size_t j = 0; j < to_add; j++) acc = bb4_add(acc, ld4(in + 4 * (j * count + i)));
        st4(out + 4 * i, acc);
    }
}
/* unset cells carry the marker 0xffffffff; they become zero */
void orc_hal_eltwise_zeroize(uint32_t* io, size_t n) {
    for (size_t i = 0; i < n; i++) if (io[i] == 0xFFFFFFFFu) io[i] = 0;
}
/* coefficient i of each of `count` polynomials of 2^log_size coefficients times shift^i */
void orc_hal_zk_shift(uint32_t* io, size_t count, int log_size, uint32_t shift) {
    const size_t n = (size_t)1 << log_size;
    for (size_t p = 0; p < count; p++) {
        bb_t s = 1;
        for (size_t i = 0; i < n; i++) { io[p * n + i] = bb_mul(io[p * n + i], s); s = bb_mul(s, shift); }
    }
}
/* out[combos[i] * count + idx] += mix_start * mix^i * in[i * count + idx]   (out: extension elements, in: base elements) */
void orc_hal_mix_poly_coeffs(uint32_t* out, const uint32_t mix_start[4], const uint32_t mix[4], const uint32_t* in,
                             const uint32_t* combos, size_t input_size, size_t count, uint32_t ext_w) {
    for (size_t idx = 0; idx < count; idx++) {
        bb4_t cur = ld4(mix_start);
        for (size_t i = 0; i < input_size; i++) {
            uint32_t* o = out + 4 * ((size_t)combos[i] * count + idx);
            st4(o, bb4_add(ld4(o), bb4_mul_base(cur, in[i * count + idx])));
            cur = ext_mul_w(cur, ld4(mix), ext_w);
        }
    }
}
/* out[e] = polynomial which[e] (2^log_size base coefficients, lowest first) evaluated at the extension point xs[e] */
void orc_hal_batch_evaluate_any(const uint32_t* coeffs, int log_size, const uint32_t* which, const uint32_t* xs, uint32_t* out,
                                size_t eval_count, uint32_t ext_w) {
    const size_t n = (size_t)1 << log_size;
    for (size_t e = 0; e < eval_count; e++) {
        const uint32_t* c = coeffs + (size_t)which[e] * n;
        const bb4_t x = ld4(xs + 4 * e);
        bb4_t acc = bb4_zero();
        for (size_t i = n; i-- > 0;) acc = bb4_add_base(ext_mul_w(acc, x, ext_w), c[i]);
        st4(out + 4 * e, acc);
    }
}
/* dst[g] = src[g * stride + idx]: one row of a column-major matrix */
void orc_hal_gather_sample(uint32_t* dst, const uint32_t* src, size_t idx, size_t size, size_t stride) {
    for (size_t g = 0; g < size; g++) dst[g] = src[g * stride + idx];
}
/* into[offsets[k]] = values[k] for k in [index[r], index[r + 1]), r < rows */
void orc_hal_scatter(uint32_t* into, const uint32_t* index, const uint32_t* offsets, const uint32_t* values, size_t rows) {
    for (size_t r = 0; r < rows; r++)
        for (uint32_t k = index[r]; k < index[r + 1]; k++) into[offsets[k]] = values[k];
}
/* inclusive prefix products of n extension elements, in place */
void orc_hal_prefix_products_ext(uint32_t* io, size_t n, uint32_t ext_w) {
    bb4_t acc = bb4_one();
    for (size_t i = 0; i < n; i++) { acc = ext_mul_w(acc, ld4(io + 4 * i), ext_w); st4(io + 4 * i, acc); }
}

/* ---- SHA-256 (FIPS 180-4), big-endian words; digests as the eight state words ---- */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha256_block(uint32_t h[8], const uint32_t blk[16]) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = blk[i];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
        uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
/* SHA-256 of `n` 32-bit words, each serialised big-endian, with the standard padding */
static void sha256_words(const uint32_t* words, size_t n, size_t stride, uint32_t out[8]) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint32_t blk[16];
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        for (int k = 0; k < 16; k++) blk[k] = words[(i + k) * stride];
        sha256_block(h, blk);
    }
    int r = 0;
    for (; i < n; i++) blk[r++] = words[i * stride];
    blk[r++] = 0x80000000u;
    if (r > 14) { while (r < 16) blk[r++] = 0; sha256_block(h, blk); r = 0; }
    while (r < 14) blk[r++] = 0;
    const uint64_t bits = (uint64_t)n * 32;
    blk[14] = (uint32_t)(bits >> 32); blk[15] = (uint32_t)bits;
    sha256_block(h, blk);
    memcpy(out, h, 32);
}
/* leaf r = SHA-256 over row r of a column-major [cols][rows] matrix (canonical words) */
void orc_hal_hash_rows_sha256(const uint32_t* mat, size_t cols, size_t rows, uint32_t* digests) {
#pragma omp parallel for
    for (size_t r = 0; r < rows; r++) sha256_words(mat + r, cols, rows, digests + 8 * r);
}
/* parents[i] = SHA-256(children[2i] || children[2i+1]) (64 bytes, standard padding) */
void orc_hal_hash_fold_sha256(const uint32_t* children, uint32_t* parents, size_t count) {
#pragma omp parallel for
    for (size_t i = 0; i < count; i++) sha256_words(children + 16 * i, 16, 1, parents + 8 * i);
}
