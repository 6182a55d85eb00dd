// ntt_bfly.cuh -- the 32-point decimation-in-frequency network in registers (signed-Montgomery butterflies, literal twiddles):
// shared by the pass kernels (ntt.hip) and the fused inverse-second / forward-first LDE kernel (ntt_fused.hip).
// Replaces the butterfly layers of p3-dft 0.2.1-succinct Radix2DitParallel (reference Cargo.lock:3903).
#pragma once
#include "babybear.cuh"

namespace zk {

struct Tw32 { uint32_t w[16]; };
constexpr Tw32 make_tw32(bool inv) {
    Tw32 t{};
    uint32_t g = two_adic_generator(5);
    if (inv) g = finv(g);
    uint32_t x = MONTY_R1;
    for (int j = 0; j < 16; j++) { t.w[j] = x; x = fmul(x, g); }
    return t;
}
constexpr Tw32 TW32_FWD = make_tw32(false);
constexpr Tw32 TW32_INV = make_tw32(true);

constexpr int rev5(int r) {
    return ((r & 1) << 4) | ((r & 2) << 2) | (r & 4) | ((r & 8) >> 2) | ((r & 16) >> 4);
}

// stage S of the 32-point decimation-in-frequency network (S = 0 pairs i, i+16).  Inputs and outputs are canonical: a butterfly is
// add + dred for the sum and, for the twiddled difference, one v_sub + a signed Montgomery product with the bias P 2^32 as its
// addend + dred (dbfly_mul below; 3 + 6 instructions).  LAZY_OUT (only legal for the twiddle-free stage 4): the sum stays a plain
// a + b in [0, 2P) and the difference a - b + P -- or, with SD, the int32 a - b -- because the next thing that touches them is a
// Montgomery multiplication (dmul / dmul_sd).
// (a - b) * w for canonical a, b and a compile-time twiddle, through the signed Montgomery product: the difference is taken as an
// int32 in (-P, P) (one v_sub instead of the two additions of a - b + P), the twiddle is centred (|w| <= P/2), and the bias P 2^32
// rides as the addend of the first v_mad_i64_i32, so the reduced value lands in (0.26 P, 1.74 P) and one conditional subtraction
// finishes: 6 instructions instead of 7 per twiddled butterfly.
ZK_D uint32_t dbfly_mul(uint32_t a, uint32_t b, int32_t wc, int64_t bias) {
    const int32_t d = (int32_t)(a - b);
    const int64_t x = (int64_t)d * wc + bias;
    const int32_t m = (int32_t)((uint32_t)x * MONTY_MU_POS);
    const int64_t y = x + (int64_t)m * (int64_t)(-(int32_t)P);
    return dred((uint32_t)(y >> 32));
}
// d * w for a signed difference d = a - b in (-P, P) left by the lazy last stage and a canonical table twiddle: result canonical
ZK_D uint32_t dmul_sd(uint32_t d_bits, uint32_t w, int64_t bias) {
    const int64_t x = (int64_t)(int32_t)d_bits * (int32_t)w + bias;           // |d w| < P^2 < P 2^32: positive with the bias
    const int32_t m = (int32_t)((uint32_t)x * MONTY_MU_POS);
    const int64_t y = x + (int64_t)m * (int64_t)(-(int32_t)P);
    return dred((uint32_t)(y >> 32));                                          // (0.03 P, 1.97 P) -> [0, P)
}
// SD (with LAZY_OUT): the differences are left as int32 a - b (one instruction) for dmul_sd instead of a - b + P (two)
template <bool INV, int S, bool LAZY_OUT = false, bool SD = false>
ZK_D void dif_stage(uint32_t (&x)[32], int64_t bias) {
    constexpr int half = 16 >> S;
    constexpr int stride = 16 / half;
    static_assert(!LAZY_OUT || S == 4, "lazy outputs only after the twiddle-free stage");
#pragma unroll
    for (int base = 0; base < 32; base += 2 * half) {
#pragma unroll
        for (int j = 0; j < half; j++) {
            const uint32_t a = x[base + j], b = x[base + j + half];
            if (LAZY_OUT) {
                x[base + j] = a + b;
                x[base + j + half] = SD ? a - b : dsub_lazy(a, b);
            } else {
                x[base + j] = dadd(a, b);
#ifdef NTT_UNSIGNED_BFLY
                x[base + j + half] = (j == 0) ? dsub(a, b) : dmul(dsub_lazy(a, b), (INV ? TW32_INV : TW32_FWD).w[j * stride]);
#else
                x[base + j + half] = (j == 0) ? dsub(a, b) : dbfly_mul(a, b, centered((INV ? TW32_INV : TW32_FWD).w[j * stride]), bias);
#endif
            }
        }
    }
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

}  // namespace zk
