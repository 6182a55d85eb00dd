// ntt_small.hip -- a WHOLE coset LDE of a 2^11 .. 2^15-row matrix as ONE launch: inverse transform, coset shift and the forward
// transform of every coset with the column in registers, 4 + 4 * 2^b bytes per trace cell instead of the six 8-byte passes of the
// two-pass path (ntt.hip) at these heights.
//
// Replaces, for these heights, p3-dft 0.2.1-succinct Radix2DitParallel::coset_lde_batch (reference Cargo.lock:3903) on the path below
// crates/guest-prover-sp1/src/sp1.rs:116 -- the heights of BASELINE.json configs[2]'s transcript proofs (a 13 KB transcript: 2^14 x 640)
// and of the recursion machines' smaller chips.
//
// Shape.  N = 1024 P rows, P = 2 .. 32.  A workgroup of 1024 threads owns CT = 32 / P adjacent columns
// (2^15 elements = its 64 waves' worth of 32 registers each): thread (t, c), t < 32 P, holds rows t + 32 P n1 of column c.  A transform
// is three register phases -- 32-point DIF over n1, 32-point DIF, P-point DIF, the networks of ntt_bfly.cuh -- with the twiddles between
// them generated as a running power of ONE table word per thread (w_N^t, then w_{32 P}^u: no N-word table fits beside the exchange
// buffer), and two exchanges through ONE 132 KiB LDS buffer:
//   E1  (k1, e) -> thread (k1, u), e = u + P n1'      block pitch 1056 words: the two 32-lane halves of a wave hit different bank halves
//   E2  inside a k1 group, (k2, u) -> thread u'' = k2 mod P      slot (u + k2) mod P: conflict-free both ways WITHOUT a padded pitch
//                                                                (a (P + 1)-pitch would need 192 KiB at P = 2)
//   E3  coefficient j = k1 + 32 k2 + 1024 k3 -> natural order t + 32 P n1 for the forward transforms      word (j + j / 32) CT + c
// The coefficients then stay in 32 VGPRs; per coset they are multiplied by shift_t^j / N (a table row, coalesced, L2-resident) into
// a second set of 32 VGPRs, transformed, and stored bit-reversed.  1024 threads x <= 128 VGPRs = four waves per SIMD, one workgroup per CU.
// Global accesses are CT x 4 bytes per row (8 bytes at 2^14 rows): the workgroups of one 128-byte row chunk are consecutive on ONE XCD
// (block -> column group map below), so a line is fetched from HBM once and hit in that XCD's L2 by the others.
// No MFMA: a 31-bit modular butterfly is not a dense contraction.
#include <atomic>

#include "babybear.cuh"
#include "kernels.h"
#include "batch.h"
#include "ntt_bfly.cuh"

namespace zk {

namespace {
constexpr int SMALL_MAX_DEVICES = 64;
constexpr int E_PITCH = 1056;                       // words per k1 block of E1 / E2: 1024 + 32
constexpr int E_WORDS = 32 * E_PITCH;               // = (N + N / 32) CT: the one exchange buffer, 132 KiB
constexpr size_t SMALL_LDS_BYTES = (size_t)E_WORDS * 4;

ZK_D void wg_sync() { __syncthreads(); }

// x[r] *= w^(rev5(r)), r = 1 .. 31, as a running power (x[0] only reduced): inputs lazy or canonical, outputs canonical
ZK_D void twiddle_chain(uint32_t (&x)[32], uint32_t w) {
    uint32_t p = w;
    x[0] = dred(x[0]);
#pragma unroll
    for (int k = 1; k < 32; k++) {
        x[rev5(k)] = dmul(x[rev5(k)], p);
        if (k < 31) p = dmul(p, w);
    }
}
template <bool INV, bool LAZY_LAST>
ZK_D void dif32(uint32_t (&x)[32], int64_t bias) {
    dif_stage<INV, 0>(x, bias);
    dif_stage<INV, 1>(x, bias);
    dif_stage<INV, 2>(x, bias);
    dif_stage<INV, 3>(x, bias);
    if (LAZY_LAST) dif_stage<INV, 4, true, false>(x, bias);
    else dif_stage<INV, 4>(x, bias);
}
// the last LOGP stages of the 32-point network = P-point DIFs over the low LOGP bits of the register index
template <bool INV, int LOGP, bool LAZY_LAST>
ZK_D void difP(uint32_t (&x)[32], int64_t bias) {
    if (LOGP >= 5) dif_stage<INV, 0>(x, bias);
    if (LOGP >= 4) dif_stage<INV, 1>(x, bias);
    if (LOGP >= 3) dif_stage<INV, 2>(x, bias);
    if (LOGP >= 2) dif_stage<INV, 3>(x, bias);
    if (LAZY_LAST) dif_stage<INV, 4, true, false>(x, bias);
    else dif_stage<INV, 4>(x, bias);
}

// One N-point DIF of the column in x (thread t holds elements t + 32 P n1, canonical) -- phases A, B, C with E1, E2 between them.
// On return thread (k1 = t >> LOGP, u = t & (P - 1)) holds in x[rho] the output k = k1 + 32 k2 + 1024 k3 with
// k2 = (rho & ~(P - 1)) + u and k3 = bitrev_LOGP(rho & (P - 1)); canonical, or lazy ([0, 2P)) with LAZY_LAST.
template <bool INV, int LOGP, bool LAZY_LAST>
ZK_D void transform(uint32_t (&x)[32], uint32_t* lds, const uint32_t* tw, int t, int c, int64_t bias) {
    constexpr int Pn = 1 << LOGP, CT = 32 >> LOGP, M = 32 * Pn;
    const int k1 = t >> LOGP, u = t & (Pn - 1);
    const uint32_t wa = tw[t], wb = tw[32 * u];                // w_N^(+-t);  w_N^(+-32 u) = w_M^(+-u)
    // ---- phase A over n1, twiddle w_N^(t k1)
    dif32<INV, true>(x, bias);
    twiddle_chain(x, wa);
    // ---- E1: x[r] is element e = t of problem rev5(r); thread (k1, u) takes elements u + P n1' of problem k1
    wg_sync();                                                   // the buffer's previous readers are done
    {
        uint32_t* wp = lds + t * CT + c;
#pragma unroll
        for (int r = 0; r < 32; r++) wp[rev5(r) * E_PITCH] = x[r];
    }
    wg_sync();
    {
        const uint32_t* rp = lds + k1 * E_PITCH + u * CT + c;
#pragma unroll
        for (int n = 0; n < 32; n++) x[n] = rp[n * (Pn * CT)];
    }
    // ---- phase B over n1', twiddle w_M^(u k2)
    dif32<INV, true>(x, bias);
    twiddle_chain(x, wb);
    // ---- E2 inside the k1 group: value (k2 = rev5(r), u) -> thread k2 mod P, register (k2 & ~(P - 1)) | u
    wg_sync();
    {
        uint32_t* gp = lds + k1 * E_PITCH + c;
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const int k2 = rev5(r);
            gp[(k2 * Pn + ((u + k2) & (Pn - 1))) * CT] = x[r];
        }
    }
    wg_sync();
    {
        const uint32_t* gp = lds + k1 * E_PITCH + c;
#pragma unroll
        for (int rho = 0; rho < 32; rho++) {
            const int tt = rho & (Pn - 1), g = rho & ~(Pn - 1);
            const int k2 = g + u;                                // (g is a multiple of P: (tt + k2) mod P = (tt + u) mod P)
            x[rho] = gp[(k2 * Pn + ((tt + u) & (Pn - 1))) * CT];
        }
    }
    // ---- phase C: P-point DIFs over u
    difP<INV, LOGP, LAZY_LAST>(x, bias);
}
constexpr uint32_t brev_small(uint32_t v, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((v >> i) & 1u) << (bits - 1 - i);
    return r;
}
}  // namespace

template <int LOGP>
__device__ __forceinline__ void lde_small_kernel_body(const LdeSmallArgs& a) {
    constexpr int Pn = 1 << LOGP, CT = 32 >> LOGP, LOG_CT = 5 - LOGP, M = 32 * Pn;
    extern __shared__ uint32_t lds[];
    int64_t bias = (int64_t)((uint64_t)P << 32);
    asm volatile("" : "+v"(bias));
    const int tid = threadIdx.x;
    const int c = tid & (CT - 1), t = tid >> LOG_CT;
    // block -> column group: blocks b, b + 8, b + 16 ... share an XCD (round-robin dispatch); the P column groups of one 128-byte row
    // chunk take P consecutive slots of ONE XCD, so that they run at the same time on the same L2
    uint32_t cg;
    {
        const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        cg = ((slot >> LOGP) * 8u + xcd) * (uint32_t)Pn + (slot & (uint32_t)(Pn - 1));
    }
    if (cg >= a.groups) return;                                  // (the grid is rounded up to whole XCD rounds)
    const uint32_t col = cg * CT + c;
    const bool active = col < a.ncols;
    const uint32_t lcol = active ? col : 0u;                     // inactive lanes load a valid address, never store

    uint32_t coef[32];
    {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.in), 0, 0xFFFFFFFFu, 0x00020000);
        const uint32_t voff = 4u * ((uint32_t)((uint64_t)t * a.in_ld) + lcol);
        const uint32_t step = (uint32_t)(4u * (uint64_t)M * a.in_ld);
#pragma unroll
        for (int n1 = 0; n1 < 32; n1++) coef[n1] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, n1 * step, 0);
    }
    // ---- coefficients: inverse transform, then E3 into the order the forward transforms start from
    transform<true, LOGP, true>(coef, lds, a.tw_inv, t, c, bias);
    const int k1 = t >> LOGP, u = t & (Pn - 1);
    wg_sync();
    {
#pragma unroll
        for (int rho = 0; rho < 32; rho++) {
            const uint32_t k2 = (uint32_t)(rho & ~(Pn - 1)) + (uint32_t)u, k3 = brev_small((uint32_t)(rho & (Pn - 1)), LOGP);
            const uint32_t j = (uint32_t)k1 + 32u * k2 + 1024u * k3;
            lds[(j + (j >> 5)) * CT + c] = coef[rho];
        }
    }
    wg_sync();
    {
#pragma unroll
        for (int n1 = 0; n1 < 32; n1++) {
            const uint32_t j = (uint32_t)t + (uint32_t)M * n1;
            coef[n1] = lds[(j + (j >> 5)) * CT + c];
        }
    }
    // ---- every coset: x = coef * shift_t^j / N, forward transform, bit-reversed store
    const uint32_t rowbase = (__brev((uint32_t)k1) >> 27) * (uint32_t)M + (__brev((uint32_t)u) >> 27) * (uint32_t)Pn;   // rev5(k1) 32 P + rev5(u) P
    for (uint32_t cs = 0; cs < a.cosets; cs++) {
        uint32_t x[32];
        {
            const uint32_t* pre = a.pre[cs] + t;
#pragma unroll
            for (int n1 = 0; n1 < 32; n1++) x[n1] = dmul(coef[n1], pre[n1 * M]);
        }
        transform<false, LOGP, false>(x, lds, a.tw_fwd, t, c, bias);
        if (active) {
            const auto ors = __builtin_amdgcn_make_buffer_rsrc(a.out[cs], 0, 0xFFFFFFFFu, 0x00020000);
            const uint32_t ostep = (uint32_t)(4u * a.out_ld);
            const uint32_t voff = rowbase * ostep + 4u * col;
#pragma unroll
            for (int rho = 0; rho < 32; rho++) {
                // row = rev5(k1) 32 P + rev5(k2) P + (rho & (P - 1)),  k2 = (rho & ~(P - 1)) + u: rev5 splits over the disjoint bits
                const uint32_t roff = (uint32_t)rev5(rho & ~(Pn - 1)) * (uint32_t)Pn + (uint32_t)(rho & (Pn - 1));
                __builtin_amdgcn_raw_buffer_store_b32(x[rho], ors, voff, roff * ostep, 0);
            }
        }
    }
}

template <int LOGP>
__global__ void __launch_bounds__(1024) lde_small_kernel(LdeSmallArgs a) { lde_small_kernel_body<LOGP>(a); }
struct lde_small_kernel_bargs { LdeSmallArgs a; static lde_small_kernel_bargs make(LdeSmallArgs a) { return lde_small_kernel_bargs{a}; } };
template <int LOGP>
__global__ void __launch_bounds__(1024) lde_small_kernel_batch(const lde_small_kernel_bargs* __restrict__ zk_arr) {
    const lde_small_kernel_bargs& zk_b = zk_arr[blockIdx.z];
    lde_small_kernel_body<LOGP>(zk_b.a);
}

bool lde_small_supported(const LdeSmallArgs& a, int log_n) {
    if (log_n < 11 || log_n > 15 || a.cosets < 1 || a.cosets > 16 || a.ncols == 0) return false;
    const uint64_t n = 1ull << log_n;
    // buffer addressing: every byte offset of a column group's rows must fit 32 bits
    if (4ull * ((n - 1) * a.in_ld + a.ncols) >= (1ull << 32) || 4ull * ((n - 1) * a.out_ld + a.ncols) >= (1ull << 32)) return false;
    return true;
}

template <int LOGP>
static hipError_t launch_small_k(const LdeSmallArgs& a_, hipStream_t s) {
    LdeSmallArgs a = a_;
    constexpr int Pn = 1 << LOGP, CT = 32 >> LOGP;
    a.groups = (a.ncols + CT - 1) / CT;
    const uint32_t round = 8u * Pn;                               // one 128-byte row chunk per XCD
    const uint32_t blocks = (a.groups + round - 1) / round * round;
    static std::atomic<bool> configured[SMALL_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SMALL_MAX_DEVICES) return hipErrorInvalidDevice;
    if (!configured[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)lde_small_kernel<LOGP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMALL_LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)lde_small_kernel_batch<LOGP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMALL_LDS_BYTES);
        if (e != hipSuccess) return e;
        configured[dev].store(true, std::memory_order_release);
    }
    ZK_LAUNCH((lde_small_kernel<LOGP>), (lde_small_kernel_batch<LOGP>), lde_small_kernel_bargs, dim3(blocks), dim3(1024), SMALL_LDS_BYTES, s, a);
    return hipGetLastError();
}

hipError_t launch_lde_small(const LdeSmallArgs& a, int log_n, hipStream_t s) {
    if (!lde_small_supported(a, log_n)) return hipErrorInvalidValue;
    switch (log_n) {
        case 11: return launch_small_k<1>(a, s);
        case 12: return launch_small_k<2>(a, s);
        case 13: return launch_small_k<3>(a, s);
        case 14: return launch_small_k<4>(a, s);
        default: return launch_small_k<5>(a, s);
    }
}

}  // namespace zk
