// babybear.cuh -- BabyBear (p = 2^31 - 2^27 + 1) in Montgomery form, R = 2^32, and the
// quartic extension F_p[x]/(x^4 - 11).  Shared by the gfx950 kernels and the host-side
// transcript code of libzkhip (host + device + constexpr in one definition).
//
// Replaces, for the shard-prove hot path, the arithmetic that the reference reaches
// through p3-baby-bear 0.2.1-succinct (reference Cargo.lock:3845) below
// crates/guest-prover-sp1/src/sp1.rs:116.  Values are canonical Montgomery residues
// in [0, p): the same in-memory form Plonky3's MontyField31 uses, which is what the
// row-major trace matrices crossing the C ABI (include/zkhip.h) hold.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#define ZK_D __device__ __forceinline__
#else
#define ZK_HD inline
#define ZK_D inline
#endif

namespace zk {

constexpr uint32_t P = 0x78000001u;        // 2013265921
constexpr uint32_t MONTY_MU = 0x88000001u; // P^-1 mod 2^32  (= 2^31 + 2^27 + 1)
constexpr uint32_t MONTY_R1 = 0x0ffffffeu; // R mod P   = Montgomery form of 1
constexpr uint32_t MONTY_R2 = 1172168163u; // R^2 mod P
constexpr uint32_t GEN = 31u;              // multiplicative generator (canonical)
constexpr int TWO_ADICITY = 27;
constexpr uint32_t TWO_ADIC_GEN = 440564289u;  // 31^15, order 2^27 (canonical)
constexpr uint32_t EXT_W = 11u;            // x^4 = 11 (canonical)

// ---- raw Montgomery arithmetic on uint32_t (all operands/results in [0, P)) ----
// montgomery reduction of x < P * 2^32:  x * 2^-32 mod P
ZK_HD constexpr uint32_t monty_reduce(uint64_t x) {
    uint32_t lo = (uint32_t)x;
    uint32_t hi = (uint32_t)(x >> 32);
    uint32_t m = lo * MONTY_MU;
    uint32_t t = (uint32_t)(((uint64_t)m * P) >> 32);
    uint32_t r = hi - t;          // in (-P, P): wraps high when hi < t
    uint32_t r2 = r + P;
    return r < r2 ? r : r2;       // unsigned min picks the in-range one
}
ZK_HD constexpr uint32_t fmul(uint32_t a, uint32_t b) { return monty_reduce((uint64_t)a * b); }
ZK_HD constexpr uint32_t fadd(uint32_t a, uint32_t b) {
    uint32_t s = a + b;          // < 2P < 2^32
    uint32_t t = s - P;          // wraps high when s < P
    return s < t ? s : t;        // unsigned min
}
ZK_HD constexpr uint32_t fsub(uint32_t a, uint32_t b) {
    uint32_t d = a - b;          // wraps high when a < b
    uint32_t t = d + P;
    return d < t ? d : t;
}
ZK_HD constexpr uint32_t fneg(uint32_t a) { return a ? P - a : 0u; }
ZK_HD constexpr uint32_t fdbl(uint32_t a) { return fadd(a, a); }
ZK_HD constexpr uint32_t to_monty(uint32_t canonical) { return (uint32_t)((((uint64_t)canonical) << 32) % P); }
ZK_HD constexpr uint32_t from_monty(uint32_t m) { return monty_reduce((uint64_t)m); }
ZK_HD constexpr uint32_t fpow(uint32_t a, uint64_t e) {
    uint32_t r = MONTY_R1;
    while (e) { if (e & 1) r = fmul(r, a); a = fmul(a, a); e >>= 1; }
    return r;
}
ZK_HD constexpr uint32_t finv(uint32_t a) { return fpow(a, (uint64_t)P - 2); }
// element of order 2^bits, Montgomery form
ZK_HD constexpr uint32_t two_adic_generator(int bits) {
    uint32_t g = to_monty(TWO_ADIC_GEN);
    for (int i = bits; i < TWO_ADICITY; i++) g = fmul(g, g);
    return g;
}
// representative of x (mod P) in (-P/2, P/2]
ZK_HD constexpr int32_t centered(uint32_t x) { return x > P / 2 ? (int32_t)x - (int32_t)P : (int32_t)x; }
constexpr uint32_t MONTY_GEN = to_monty(GEN);
constexpr uint32_t MONTY_EXT_W = to_monty(EXT_W);
constexpr uint32_t MONTY_INV2 = to_monty((P + 1) / 2);

// ---- gfx950-tuned device primitives ------------------------------------------------------
// Measured on MI355X (tools/microbench): add/sub/logic/shift 2.4 clk per wave64 instruction,
// min/max/add3/lshl_add/*_co and every integer multiply 4.2 clk, and a v_cndmask right
// behind the v_sub_co that feeds it is nearly free (pair = 4.8 clk).  So:
//   dmont_lazy : 2 x v_mad_u64_u32 + v_mul_lo_u32 (12.8 clk), result in [0, 2P), no fix-up;
//                the multiplicand may be ANY 32-bit value as long as the other factor is < P
//   dred       : [0, 2P) -> [0, P) as v_subrev_co + v_cndmask (4.8 clk)
//   dsub_lazy  : a - b + P in (0, 2P), two plain adds (4.8 clk), feeds dmont_lazy directly
#if defined(__HIPCC__)
constexpr uint32_t MONTY_MU_NEG = 0x77ffffffu;   // -P^-1 mod 2^32
ZK_D uint32_t dmont_lazy(uint32_t a, uint32_t b) {
    uint64_t x = (uint64_t)a * b;
    uint32_t m = (uint32_t)x * MONTY_MU_NEG;
    uint64_t y = x + (uint64_t)m * P;            // low word cancels; < 2^33 P
    return (uint32_t)(y >> 32);
}
ZK_D uint32_t dred(uint32_t x) {
    uint32_t t;
    asm("v_subrev_co_u32_e32 %0, vcc, 0x78000001, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "=&v"(t) : "v"(x) : "vcc");
    return t;
}
ZK_D uint32_t dmul(uint32_t a, uint32_t b) { return dred(dmont_lazy(a, b)); }
ZK_D uint32_t dadd(uint32_t a, uint32_t b) { return dred(a + b); }
ZK_D uint32_t dsub(uint32_t a, uint32_t b) {
    uint32_t t, u;
    asm("v_sub_co_u32_e32 %0, vcc, %2, %3\n\tv_add_u32_e32 %1, 0x78000001, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc"
        : "=&v"(t), "=&v"(u) : "v"(a), "v"(b) : "vcc");
    return t;
}
ZK_D uint32_t dsub_lazy(uint32_t a, uint32_t b) { return a + (P - b); }
// a0*b0 + a1*b1 reduced together (operands canonical): 3 x v_mad_u64_u32 + v_mul_lo_u32 + dred
ZK_D uint32_t dmr2(uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
    uint64_t x = (uint64_t)a0 * b0 + (uint64_t)a1 * b1;       // < 2 P^2
    uint32_t m = (uint32_t)x * MONTY_MU_NEG;
    uint64_t y = x + (uint64_t)m * P;                         // < 2 P^2 + 2^32 P < 2^64
    return dred((uint32_t)(y >> 32));                         // y / 2^32 < 2 P
}
ZK_D uint32_t ddbl(uint32_t a) { return dred(a + a); }
// Montgomery reduction of an accumulated 64-bit value x < 2^32 P, result in [0, 2P)
ZK_D uint32_t dmred_lazy(uint64_t x) {
    uint32_t m = (uint32_t)x * MONTY_MU_NEG;
    uint64_t y = x + (uint64_t)m * P;
    return (uint32_t)(y >> 32);
}
// a * b + c as one v_mad_u64_u32 (the 64-bit addend rides along for free)
ZK_D uint64_t dmac(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }
// Running dot product in a 64-bit accumulator.  Invariant between calls: acc < 2^32 P.  Two products of
// canonical factors add < 2 P^2, so acc < 2^32 P + 2 P^2 < 2^64 and its high word is < 2P: one conditional
// subtraction of P from the HIGH word (= subtracting 2^32 P, a multiple of P) restores the invariant.
// 2 x v_mad_u64_u32 + v_subrev_co + v_cndmask per pair of products; no Montgomery reduction until the end.
ZK_D void dacc2(uint64_t& acc, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
    const uint64_t t = dmac(a1, b1, dmac(a0, b0, acc));
    acc = ((uint64_t)dred((uint32_t)(t >> 32)) << 32) | (uint32_t)t;
}
// Signed Montgomery product: for ANY int32 a, b the result r = (a b - m P) / 2^32 (m = a b P^-1 mod 2^32 taken as int32) is
// congruent to a b / 2^32 and |r| <= |a b| / 2^32 + P/2 -- so (-P, P) is closed under it (P^2 / 2^32 < 0.47 P) and a chain
// of products needs no conditional subtraction at all: v_mad_i64_i32 + v_mul_lo_u32 + v_mad_i64_i32 (12.6 clk).
constexpr uint32_t MONTY_MU_POS = 0x88000001u;   // P^-1 mod 2^32
static_assert((uint32_t)(P * MONTY_MU_POS) == 1u, "P^-1 mod 2^32");
ZK_D int32_t dsmont(int32_t a, int32_t b) {
    const int64_t x = (int64_t)a * b;
    const int32_t m = (int32_t)((uint32_t)x * MONTY_MU_POS);
    const int64_t y = x + (int64_t)m * (int64_t)(-(int32_t)P);
    return (int32_t)(y >> 32);
}
// Montgomery reduction of a signed 64-bit value: x / 2^32 mod P with |result| <= |x| / 2^32 + P/2
ZK_D int32_t dsmred(int64_t x) {
    const int32_t m = (int32_t)((uint32_t)x * MONTY_MU_POS);
    const int64_t y = x + (int64_t)m * (int64_t)(-(int32_t)P);
    return (int32_t)(y >> 32);
}
ZK_D int64_t dsmac(int32_t a, int32_t b, int64_t c) { return (int64_t)a * b + c; }     // one v_mad_i64_i32
// the same with the instruction spelled out and a wave-uniform multiplier: for sums of the form sum_j a_j * k the compiler
// otherwise factors k out and builds the 64-bit sum from 2 instructions per term (sign extension + 64-bit add)
ZK_D int64_t dsmac_uniform(int32_t a, int32_t k_sgpr, int64_t c) {
    uint64_t carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(c), "=s"(carry) : "v"(a), "s"(k_sgpr));
    return c;
}
// a * k + c with the instruction spelled out, k wave-uniform (an SGPR): the multipliers of the Poseidon2 internal layer come
// from the parameter tables in constant memory, and once their sign extension is hoisted out of the round loop the compiler no
// longer sees a 32 x 32 -> 64 product (it builds a 64 x 32 one from v_mad_u64_u32 + fix-ups: 2.6x the instructions)
ZK_D int64_t dsmac_s(int32_t a, int32_t k_sgpr, int64_t c) {
    int64_t d;
    uint64_t carry;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=&v"(d), "=s"(carry) : "v"(a), "s"(k_sgpr), "v"(c));
    return d;
}
// (-P, P) -> [0, P): v_add + v_min_u32 (a negative value is a huge unsigned one, adding P wraps it into range)
ZK_D uint32_t dcanon(int32_t x) { const uint32_t u = (uint32_t)x, v = u + P; return v < u ? v : u; }
// the same product plus P: an unsigned value in (0, 2P) when |a b| < 2^31 P
ZK_D uint32_t dsmont_plus_p(int32_t a, int32_t b) { return (uint32_t)dsmont(a, b) + P; }
ZK_D uint32_t dacc_finish(uint64_t acc) { return dred(dmred_lazy(acc)); }    // canonical acc / 2^32 mod P
#endif

// ---- quartic extension, coefficients in Montgomery form ----
struct Ext { uint32_t c[4]; };

ZK_HD constexpr Ext ext_zero() { return Ext{{0, 0, 0, 0}}; }
ZK_HD constexpr Ext ext_one() { return Ext{{MONTY_R1, 0, 0, 0}}; }
ZK_HD constexpr Ext ext_from_base(uint32_t a) { return Ext{{a, 0, 0, 0}}; }
ZK_HD constexpr bool ext_eq(const Ext& a, const Ext& b) {
    return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3];
}
ZK_HD constexpr Ext ext_add(const Ext& a, const Ext& b) {
    return Ext{{fadd(a.c[0], b.c[0]), fadd(a.c[1], b.c[1]), fadd(a.c[2], b.c[2]), fadd(a.c[3], b.c[3])}};
}
ZK_HD constexpr Ext ext_sub(const Ext& a, const Ext& b) {
    return Ext{{fsub(a.c[0], b.c[0]), fsub(a.c[1], b.c[1]), fsub(a.c[2], b.c[2]), fsub(a.c[3], b.c[3])}};
}
ZK_HD constexpr Ext ext_neg(const Ext& a) { return Ext{{fneg(a.c[0]), fneg(a.c[1]), fneg(a.c[2]), fneg(a.c[3])}}; }
ZK_HD constexpr Ext ext_mul_base(const Ext& a, uint32_t b) {
    return Ext{{fmul(a.c[0], b), fmul(a.c[1], b), fmul(a.c[2], b), fmul(a.c[3], b)}};
}
ZK_HD constexpr Ext ext_add_base(Ext a, uint32_t b) { a.c[0] = fadd(a.c[0], b); return a; }
ZK_HD constexpr Ext ext_sub_base(Ext a, uint32_t b) { a.c[0] = fsub(a.c[0], b); return a; }
// (a*b) with x^4 = W: pairs of products are summed in 64 bits and reduced together
// (2 P^2 < P 2^32), which halves the number of Montgomery reductions.
ZK_HD constexpr uint32_t mr2(uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
    return monty_reduce((uint64_t)a0 * b0 + (uint64_t)a1 * b1);
}
ZK_HD constexpr Ext ext_mul(const Ext& a, const Ext& b) {
    const uint32_t* x = a.c; const uint32_t* y = b.c;
    // high half  h_k = sum_{i+j = k+4}
    uint32_t h0 = fadd(mr2(x[1], y[3], x[2], y[2]), fmul(x[3], y[1]));
    uint32_t h1 = mr2(x[2], y[3], x[3], y[2]);
    uint32_t h2 = fmul(x[3], y[3]);
    uint32_t l0 = fmul(x[0], y[0]);
    uint32_t l1 = mr2(x[0], y[1], x[1], y[0]);
    uint32_t l2 = fadd(mr2(x[0], y[2], x[1], y[1]), fmul(x[2], y[0]));
    uint32_t l3 = fadd(mr2(x[0], y[3], x[1], y[2]), mr2(x[2], y[1], x[3], y[0]));
    return Ext{{fadd(l0, fmul(h0, MONTY_EXT_W)), fadd(l1, fmul(h1, MONTY_EXT_W)),
                fadd(l2, fmul(h2, MONTY_EXT_W)), l3}};
}
ZK_HD constexpr Ext ext_sqr(const Ext& a) { return ext_mul(a, a); }
ZK_HD constexpr Ext ext_pow(Ext a, uint64_t e) {
    Ext r = ext_one();
    while (e) { if (e & 1) r = ext_mul(r, a); a = ext_mul(a, a); e >>= 1; }
    return r;
}
// Frobenius x -> x^p acts on the basis as x^i -> z^i x^i with z = W^((p-1)/4).
constexpr uint32_t FROB_Z1 = to_monty(1728404513u);           // 11^((p-1)/4)
constexpr uint32_t FROB_Z2 = fmul(FROB_Z1, FROB_Z1);          // = -1
constexpr uint32_t FROB_Z3 = fmul(FROB_Z2, FROB_Z1);
ZK_HD constexpr Ext ext_frobenius(const Ext& a) {
    return Ext{{a.c[0], fmul(a.c[1], FROB_Z1), fmul(a.c[2], FROB_Z2), fmul(a.c[3], FROB_Z3)}};
}
// inverse through the norm to F_p:  a^-1 = a^(r-1) / a^r,  r = 1 + p + p^2 + p^3
ZK_HD constexpr Ext ext_inv(const Ext& a) {
    Ext f1 = ext_frobenius(a);
    Ext f2 = ext_frobenius(f1);
    Ext f3 = ext_frobenius(f2);
    Ext conj = ext_mul(ext_mul(f1, f2), f3);
    // norm = (a * conj) lies in F_p: only coefficient 0 is needed
    const uint32_t* x = a.c; const uint32_t* y = conj.c;
    uint32_t h0 = fadd(mr2(x[1], y[3], x[2], y[2]), fmul(x[3], y[1]));
    uint32_t norm = fadd(fmul(x[0], y[0]), fmul(h0, MONTY_EXT_W));
    return ext_mul_base(conj, finv(norm));
}

#if defined(__HIPCC__)
// Device form of ext_mul (same value): every output coefficient is ONE running 64-bit sum of four products (dacc2) and one
// Montgomery reduction; the x^4 = W wrap-around is folded into b beforehand (3 products).  6 multiplies + 4 conditional
// subtractions per coefficient against roughly twice that through the portable form.
ZK_D Ext ext_mul_dev(const Ext& a, const Ext& b) {
    const uint32_t w1 = dmul(b.c[1], MONTY_EXT_W), w2 = dmul(b.c[2], MONTY_EXT_W), w3 = dmul(b.c[3], MONTY_EXT_W);
    uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    dacc2(s0, a.c[0], b.c[0], a.c[1], w3); dacc2(s0, a.c[2], w2, a.c[3], w1);
    dacc2(s1, a.c[0], b.c[1], a.c[1], b.c[0]); dacc2(s1, a.c[2], w3, a.c[3], w2);
    dacc2(s2, a.c[0], b.c[2], a.c[1], b.c[1]); dacc2(s2, a.c[2], b.c[0], a.c[3], w3);
    dacc2(s3, a.c[0], b.c[3], a.c[1], b.c[2]); dacc2(s3, a.c[2], b.c[1], a.c[3], b.c[0]);
    return Ext{{dacc_finish(s0), dacc_finish(s1), dacc_finish(s2), dacc_finish(s3)}};
}
ZK_D Ext ext_mul_base_dev(const Ext& a, uint32_t b) { return Ext{{dmul(a.c[0], b), dmul(a.c[1], b), dmul(a.c[2], b), dmul(a.c[3], b)}}; }
// a^(P-2) with the tuned product: P - 2 = 0x77ffffff
ZK_D uint32_t finv_dev(uint32_t a) {
    uint32_t r = MONTY_R1;
    uint32_t e = P - 2;
#pragma unroll 1
    while (e) { if (e & 1u) r = dmul(r, a); a = dmul(a, a); e >>= 1; }
    return r;
}
// device form of ext_inv (same value, zero maps to zero)
ZK_D Ext ext_inv_dev(const Ext& a) {
    const Ext f1 = Ext{{a.c[0], dmul(a.c[1], FROB_Z1), dmul(a.c[2], FROB_Z2), dmul(a.c[3], FROB_Z3)}};
    const Ext f2 = Ext{{f1.c[0], dmul(f1.c[1], FROB_Z1), dmul(f1.c[2], FROB_Z2), dmul(f1.c[3], FROB_Z3)}};
    const Ext f3 = Ext{{f2.c[0], dmul(f2.c[1], FROB_Z1), dmul(f2.c[2], FROB_Z2), dmul(f2.c[3], FROB_Z3)}};
    const Ext conj = ext_mul_dev(ext_mul_dev(f1, f2), f3);
    // norm = coefficient 0 of a * conj (the others vanish)
    uint64_t s0 = 0;
    dacc2(s0, a.c[0], conj.c[0], a.c[1], dmul(conj.c[3], MONTY_EXT_W));
    dacc2(s0, a.c[2], dmul(conj.c[2], MONTY_EXT_W), a.c[3], dmul(conj.c[1], MONTY_EXT_W));
    return ext_mul_base_dev(conj, finv_dev(dacc_finish(s0)));
}
#endif

ZK_HD constexpr uint32_t reverse_bits(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

}  // namespace zk
