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

// ---- ONE permutation, the state in one register (element i in 32-bit lane i): what a transcript -- a single serial sponge chain -- can use of the vector
// unit.  External rounds: sixteen S-boxes per vmul; M4 = circ(2, 3, 1, 1) on every 128-bit block as y = (x + r1 + r2 + r3) + x + 2 r1 with r_k the block rotated
// by k lanes, then the sum over the four blocks.  Internal rounds: lane 0 lives in a scalar register (its S-box is the dependency chain: four scalar products),
// lanes 1 .. 15 take their diagonal product and the sum of the others beside it.  Canonical Montgomery residues in and out; modular additions are exact, so the
// order they are taken in does not change a bit of the result.
namespace {
#define ZK_ROT1(v) _mm512_shuffle_epi32(v, (_MM_PERM_ENUM)0x39)
#define ZK_ROT2(v) _mm512_shuffle_epi32(v, (_MM_PERM_ENUM)0x4E)
#define ZK_ROT3(v) _mm512_shuffle_epi32(v, (_MM_PERM_ENUM)0x93)
#define ZK_BLK1(v) _mm512_shuffle_i32x4(v, v, 0x39)
#define ZK_BLK2(v) _mm512_shuffle_i32x4(v, v, 0x4E)
#define ZK_BLK3(v) _mm512_shuffle_i32x4(v, v, 0x93)
inline V external_linear_h(V s) {
    const V r1 = ZK_ROT1(s), r2 = ZK_ROT2(s), r3 = ZK_ROT3(s);
    const V sum = vadd(vadd(s, r1), vadd(r2, r3));
    const V y = vadd(vadd(sum, s), vdbl(r1));
    const V t = vadd(vadd(y, ZK_BLK1(y)), vadd(ZK_BLK2(y), ZK_BLK3(y)));
    return vadd(y, t);
}
inline uint32_t lane0(V v) { return (uint32_t)_mm_cvtsi128_si32(_mm512_castsi512_si128(v)); }
void permute_h(uint32_t st[16]) {
    V s = _mm512_loadu_si512((const void*)st);
    const V diag = _mm512_loadu_si512((const void*)P2K.diag);
    s = external_linear_h(s);
    for (int r = 0; r < 4; r++) s = external_linear_h(sbox(vadd(s, _mm512_loadu_si512((const void*)P2K.ext_rc[r]))));
    uint32_t s0 = lane0(s);
    s = _mm512_maskz_mov_epi32(0xFFFE, s);                       // lanes 1 .. 15; lane 0 stays zero in here
    for (int r = 0; r < 13; r++) {
        V t = vadd(s, ZK_ROT2(s));                               // the others' sum, in every lane
        t = vadd(t, ZK_ROT1(t));
        t = vadd(t, ZK_BLK2(t));
        t = vadd(t, ZK_BLK1(t));
        const V prod = vmul(s, diag);
        const uint32_t x0 = p2_sbox(fadd(s0, P2K.int_rc[r]));
        const uint32_t sum = fadd(x0, lane0(t));
        s0 = fadd(fmul(x0, P2K.diag[0]), sum);
        s = _mm512_maskz_mov_epi32(0xFFFE, vadd(prod, splat(sum)));
    }
    s = _mm512_mask_set1_epi32(s, 1, (int)s0);
    for (int r = 4; r < 8; r++) s = external_linear_h(sbox(vadd(s, _mm512_loadu_si512((const void*)P2K.ext_rc[r]))));
    _mm512_storeu_si512((void*)st, s);
}
// checked once per table set against the scalar form on states that include the extremes; a mismatch keeps the scalar form for good
bool permute_h_checked() {
    uint64_t z = 0xD1B54A32D192ED03ull;
    for (int round = 0; round < 6; round++) {
        uint32_t a[16], b[16];
        for (int e = 0; e < 16; e++) {
            z = z * 6364136223846793005ull + 1442695040888963407ull;
            a[e] = b[e] = round == 0 ? ((e & 1) ? P - 1 : 0u) : (uint32_t)((z >> 33) % P);
        }
        permute_h(a);
        p2_permute_scalar(b);
        for (int e = 0; e < 16; e++) if (a[e] != b[e]) return false;
    }
    return true;
}
}  // namespace

bool p2h_permute(uint32_t st[16]) {
    if (!p2x16_available()) return false;
    static const bool ok = permute_h_checked();                 // (with the tables in effect at the first call; another set is another instance of the same arithmetic)
    if (!ok) return false;
    permute_h(st);
    return true;
}

void p2x16_to_monty(uint32_t v[16]) {
    const V x = _mm512_loadu_si512((const void*)v);
    _mm512_storeu_si512((void*)v, vmul(x, splat(MONTY_R2)));      // canonical -> Montgomery: multiply by R^2
}

}  // namespace zk
