/*
 * oracle/poseidon2_x8.c -- the SAME width-16 Poseidon2 sponge and compression as poseidon2.c, eight rows at a time in the
 * 64-bit lanes of an AVX-512 register.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle/oracle.h).
 *
 * Why: a full-size oracle proof (2^20 x 256: 67 M leaf permutations) spends half its time in the scalar permutation, and the GPU
 * suite runs a dozen of them (VERDICT r5 "What's weak" 8).  This file restates poseidon2.c's permute / sponge / compress lane-wise
 * -- canonical residues in [0, p) as everywhere in the oracle, products reduced by Barrett (mu = floor(2^62 / p)) instead of the
 * scalar code's `%`, the small-constant matrices by additions -- and shares no code with the product (csrc/p2_x16.cpp is the
 * product's own host SIMD; nothing under oracle/ includes it).  It is used only when the CPU has AVX-512 F + DQ AND a start-up
 * self-check finds it equal to the scalar functions on 4 096 pseudo-random states, sponges of every length 0 .. 40 and
 * compressions; a mismatch aborts the process (an oracle that disagrees with itself must not go on).  ORC_NO_SIMD=1 keeps the
 * scalar path (tests/test_oracle.py compares the two on whole trees).
 * Reference: the algorithm is poseidon2.c's (p3-poseidon2 / p3-symmetric 0.2.1-succinct, reference Cargo.lock:4030,4044).
 */
#include "oracle.h"
#include "p2_params.h"
#include <immintrin.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define TGT __attribute__((target("avx512f,avx512dq")))
typedef __m512i v8;

static inline TGT v8 vset1(uint64_t x) { return _mm512_set1_epi64((long long)x); }
/* [0, 2p) -> [0, p) */
static inline TGT v8 vfix(v8 s) { return _mm512_min_epu64(s, _mm512_sub_epi64(s, vset1(BB_P))); }
static inline TGT v8 vadd(v8 a, v8 b) { return vfix(_mm512_add_epi64(a, b)); }
/* a b mod p for a, b < p: x = a b < 2^62, q = ((x >> 30) mu) >> 32 >= floor(x / p) - 2, so x - q p < 3 p */
static inline TGT v8 vmul(v8 a, v8 b) {
    const v8 x = _mm512_mul_epu32(a, b);
    const v8 q = _mm512_srli_epi64(_mm512_mul_epu32(_mm512_srli_epi64(x, 30), vset1(2290649223ull)), 32);      /* floor(2^62 / p) */
    return vfix(vfix(_mm512_sub_epi64(x, _mm512_mul_epu32(q, vset1(BB_P)))));
}
static inline TGT v8 vsbox7(v8 x) {
    const v8 x2 = vmul(x, x), x4 = vmul(x2, x2), x6 = vmul(x4, x2);
    return vmul(x6, x);
}
/* M4 = circ(2, 3, 1, 1) on four lanes' worth of state words, by additions */
static inline TGT void vm4(v8* s) {
    const v8 a = s[0], b = s[1], c = s[2], d = s[3];
    const v8 ab = vadd(a, b), cd = vadd(c, d), all = vadd(ab, cd);
    /* row i = all + s_i + 2 s_{i+1}: (2,3,1,1) . (a,b,c,d) = a + b + c + d + a + 2 b */
    s[0] = vadd(vadd(all, a), vadd(b, b));
    s[1] = vadd(vadd(all, b), vadd(c, c));
    s[2] = vadd(vadd(all, c), vadd(d, d));
    s[3] = vadd(vadd(all, d), vadd(a, a));
}
static inline TGT void vexternal(v8 s[16]) {
    for (int b = 0; b < 4; b++) vm4(s + 4 * b);
    v8 sums[4];
    for (int k = 0; k < 4; k++) sums[k] = vadd(vadd(s[k], s[4 + k]), vadd(s[8 + k], s[12 + k]));
    for (int i = 0; i < 16; i++) s[i] = vadd(s[i], sums[i % 4]);
}
static inline TGT void vinternal(v8 s[16]) {
    v8 sum = s[0];
    for (int i = 1; i < 16; i++) sum = vadd(sum, s[i]);
    for (int i = 0; i < 16; i++) s[i] = vadd(vmul(s[i], vset1(P2_INTERNAL_DIAG[i])), sum);
}
static TGT void vpermute(v8 s[16]) {
    vexternal(s);
    for (int r = 0; r < P2_ROUNDS_F / 2; r++) {
        for (int i = 0; i < 16; i++) s[i] = vsbox7(vadd(s[i], vset1(P2_EXTERNAL_RC[r][i])));
        vexternal(s);
    }
    for (int r = 0; r < P2_ROUNDS_P; r++) {
        s[0] = vsbox7(vadd(s[0], vset1(P2_INTERNAL_RC[r])));
        vinternal(s);
    }
    for (int r = P2_ROUNDS_F / 2; r < P2_ROUNDS_F; r++) {
        for (int i = 0; i < 16; i++) s[i] = vsbox7(vadd(s[i], vset1(P2_EXTERNAL_RC[r][i])));
        vexternal(s);
    }
}
static inline TGT v8 vgather(const uint32_t* const p[8], size_t i) {
    return _mm512_set_epi64(p[7][i], p[6][i], p[5][i], p[4][i], p[3][i], p[2][i], p[1][i], p[0][i]);
}
static inline TGT void vscatter8(const v8 st[16], uint32_t* const out[8]) {
    uint64_t tmp[8];
    for (int j = 0; j < 8; j++) {
        _mm512_storeu_si512((void*)tmp, st[j]);
        for (int l = 0; l < 8; l++) out[l][j] = (uint32_t)tmp[l];
    }
}
/* eight sponges over inputs of ONE length (rows of one matrix set) */
static TGT void sponge_x8(const uint32_t* const in[8], size_t n, uint32_t* const out[8]) {
    v8 st[16];
    for (int j = 0; j < 16; j++) st[j] = _mm512_setzero_si512();
    size_t pos = 0;
    for (size_t i = 0; i < n; i++) {
        st[pos++] = vgather(in, i);
        if (pos == 8) { vpermute(st); pos = 0; }
    }
    if (pos != 0) vpermute(st);
    vscatter8(st, out);
}
static TGT void compress_x8(const uint32_t* const left[8], const uint32_t* const right[8], uint32_t* const out[8]) {
    v8 st[16];
    for (int j = 0; j < 8; j++) { st[j] = vgather(left, (size_t)j); st[8 + j] = vgather(right, (size_t)j); }
    vpermute(st);
    vscatter8(st, out);
}

/* ---- dispatch: AVX-512 present, not switched off, and equal to the scalar functions on the self-check */
static uint64_t mix64(uint64_t* s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static int selfcheck(void) {
    uint64_t seed = 0x5A4B544C53ull;
    uint32_t in[8][48], want[8][8], got[8][8];
    const uint32_t* ip[8]; uint32_t* op[8];
    for (int l = 0; l < 8; l++) { ip[l] = in[l]; op[l] = got[l]; }
    for (int round = 0; round < 512; round++) {
        const size_t n = (size_t)(round % 41);
        for (int l = 0; l < 8; l++) for (int j = 0; j < 48; j++) in[l][j] = (uint32_t)(mix64(&seed) % BB_P);
        if (round == 0) for (int j = 0; j < 48; j++) { in[0][j] = BB_P - 1; in[1][j] = 0; }      /* the edges of the range */
        for (int l = 0; l < 8; l++) orc_sponge_hash(in[l], n, want[l]);
        sponge_x8(ip, n, op);
        if (memcmp(want, got, sizeof want) != 0) return 0;
        const uint32_t* lp[8]; const uint32_t* rp[8];
        for (int l = 0; l < 8; l++) { lp[l] = in[l]; rp[l] = in[l] + 8; orc_compress(in[l], in[l] + 8, want[l]); }
        compress_x8(lp, rp, op);
        if (memcmp(want, got, sizeof want) != 0) return 0;
    }
    return 1;
}
int orc_simd_enabled(void) {
    static int state = -1;                      /* (first call may race between threads: both compute the same answer) */
    if (state >= 0) return state;
    int on = 0;
    const char* off = getenv("ORC_NO_SIMD");
    if (!(off && off[0] && off[0] != '0') && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq")) {
        if (!selfcheck()) { fprintf(stderr, "oracle: the AVX-512 Poseidon2 disagrees with the scalar one -- refusing to go on\n"); abort(); }
        on = 1;
    }
    state = on;
    return on;
}
void orc_sponge_hash_x8(const uint32_t* const in[8], size_t n, uint32_t* const out[8]) {
    if (orc_simd_enabled()) { sponge_x8(in, n, out); return; }
    for (int l = 0; l < 8; l++) orc_sponge_hash(in[l], n, out[l]);
}
void orc_compress_x8(const uint32_t* const left[8], const uint32_t* const right[8], uint32_t* const out[8]) {
    if (orc_simd_enabled()) { compress_x8(left, right, out); return; }
    for (int l = 0; l < 8; l++) orc_compress(left[l], right[l], out[l]);
}
