// p2_x16.cpp -- sixteen Poseidon2 (width 16) permutations at once on the HOST, one state per AVX-512 lane.
//
// The reference checks every proof on the CPU right after proving it (client.verify, crates/guest-prover-sp1/src/sp1.rs:120); a
// verifier spends its time in Poseidon2: ~200-300 permutations per query (the leaf of the opened trace row, the Merkle paths, the FRI
// layers), 100 queries.  One permutation is a single dependency chain, so the vector unit is used ACROSS the queries instead: element
// i of sixteen independent states lives in one 512-bit register, every step of the permutation is element-wise, and there are no
// shuffles.  Same arithmetic as the scalar p2_permute (poseidon2.cuh): canonical Montgomery residues in, canonical out, bit for bit.
// Compiled for the host only, with -mavx512f -mavx512dq; p2x16_available() reports at run time whether the CPU has them (the callers
// fall back to the scalar permutation otherwise).
#include <immintrin.h>

#include <atomic>

#include "p2_x16.h"
#include "poseidon2.cuh"

namespace zk {

static std::atomic<bool> g_enabled{true};
bool p2x16_available() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
    return ok && g_enabled.load(std::memory_order_relaxed);
}
bool p2x16_enable(bool on) { return g_enabled.exchange(on); }

namespace {
typedef __m512i V;
inline V splat(uint32_t x) { return _mm512_set1_epi32((int)x); }
inline V vadd(V a, V b) {
    const V t = _mm512_add_epi32(a, b);                          // < 2P < 2^32
    return _mm512_min_epu32(t, _mm512_sub_epi32(t, splat(P)));   // t - P wraps high when t < P
}
inline V vdbl(V a) { return vadd(a, a); }
// Montgomery product of canonical residues: the 32 x 32 -> 64 multiplier works on the even 32-bit lanes, so the odd lanes go
// through it a second time shifted down; (x - m P) >> 32 with m = lo(x) P^-1 mod 2^32, then + P when negative
inline V vmul(V a, V b) {
    const V mu = splat(MONTY_MU), p = splat(P);
    const V pe = _mm512_mul_epu32(a, b);
    const V po = _mm512_mul_epu32(_mm512_srli_epi64(a, 32), _mm512_srli_epi64(b, 32));
    const V qe = _mm512_mul_epu32(_mm512_mul_epu32(pe, mu), p);
    const V qo = _mm512_mul_epu32(_mm512_mul_epu32(po, mu), p);
    const V de = _mm512_sub_epi64(pe, qe), dod = _mm512_sub_epi64(po, qo);      // low halves cancel; the result sits in the high halves
    const V r = _mm512_mask_blend_epi32(0xAAAA, _mm512_srli_epi64(de, 32), dod);
    return _mm512_min_epu32(r, _mm512_add_epi32(r, p));
}
inline V sbox(V x) {
    const V x2 = vmul(x, x), x3 = vmul(x2, x), x4 = vmul(x2, x2);
    return vmul(x3, x4);
}
inline void m4(V& x0, V& x1, V& x2, V& x3) {                     // circ(2, 3, 1, 1), the operation order of p2_m4
    const V t01 = vadd(x0, x1), t23 = vadd(x2, x3), t0123 = vadd(t01, t23);
    const V t01123 = vadd(t0123, x1), t01233 = vadd(t0123, x3);
    const V n3 = vadd(t01233, vdbl(x0)), n1 = vadd(t01123, vdbl(x2));
    x0 = vadd(t01123, t01);
    x2 = vadd(t01233, t23);
    x1 = n1;
    x3 = n3;
}
inline void external_linear(V s[16]) {
    for (int b = 0; b < 16; b += 4) m4(s[b], s[b + 1], s[b + 2], s[b + 3]);
    V t[4];
    for (int j = 0; j < 4; j++) t[j] = vadd(vadd(s[j], s[4 + j]), vadd(s[8 + j], s[12 + j]));
    for (int b = 0; b < 16; b += 4)
        for (int j = 0; j < 4; j++) s[b + j] = vadd(s[b + j], t[j]);
}
inline void internal_linear(V s[16]) {
    const V a = vadd(vadd(s[0], s[1]), vadd(s[2], s[3])), b = vadd(vadd(s[4], s[5]), vadd(s[6], s[7]));
    const V c = vadd(vadd(s[8], s[9]), vadd(s[10], s[11])), d = vadd(vadd(s[12], s[13]), vadd(s[14], s[15]));
    const V sum = vadd(vadd(a, b), vadd(c, d));
    for (int i = 0; i < 16; i++) s[i] = vadd(vmul(s[i], splat(P2K.diag[i])), sum);
}
}  // namespace

void p2x16_permute(uint32_t st[16][16]) {
    V s[16];
    for (int i = 0; i < 16; i++) s[i] = _mm512_loadu_si512((const void*)st[i]);
    external_linear(s);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 16; i++) s[i] = sbox(vadd(s[i], splat(P2K.ext_rc[r][i])));
        external_linear(s);
    }
    for (int r = 0; r < 13; r++) {
        s[0] = sbox(vadd(s[0], splat(P2K.int_rc[r])));
        internal_linear(s);
    }
    for (int r = 4; r < 8; r++) {
        for (int i = 0; i < 16; i++) s[i] = sbox(vadd(s[i], splat(P2K.ext_rc[r][i])));
        external_linear(s);
    }
    for (int i = 0; i < 16; i++) _mm512_storeu_si512((void*)st[i], s[i]);
}

void p2x16_to_monty(uint32_t v[16]) {
    const V x = _mm512_loadu_si512((const void*)v);
    _mm512_storeu_si512((void*)v, vmul(x, splat(MONTY_R2)));      // canonical -> Montgomery: multiply by R^2
}

}  // namespace zk
