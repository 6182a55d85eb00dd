// stark.hip -- gfx950 kernels for the STARK stages of a shard proof over the synthetic AIR:
// domain tables, quotient values, out-of-domain opening (barycentric), alpha-batched
// reduced openings, FRI fold, proof-of-work search and the query gather.
//
// Replaces, on the path below crates/guest-prover-sp1/src/sp1.rs:116: sp1-stark 4.1.4
// `quotient_values` (reference Cargo.lock:6172), p3-uni-stark folders (:4055), p3-fri
// TwoAdicFriPcs::open + prover::{commit_phase, answer_query} (:3930), p3-challenger
// `grind` (:3875).  RISC Zero twins behind risc0-zkp Hal (:5057): eval_check,
// batch_evaluate_any, mix_poly_coeffs, fri_fold, gather_sample.
//
// Layout: every matrix is row-major, rows in bit-reversed evaluation order (row p holds
// the point x_p = g * w_2N^bitrev(p)); extension elements are 4 consecutive words.
// Row-wise kernels give a row to a group of L <= 64 adjacent lanes so that a wave reads
// whole rows with 16-byte loads and reduces with cross-lane shuffles (no LDS traffic).
#include <atomic>

#include "poseidon2.cuh"
#include "kernels.h"
#include "batch.h"

namespace zk {

hipError_t stark_upload_p2_tables(const P2Tables& t, hipStream_t s) { return p2_upload_tables(t, s); }


ZK_D Ext ld_ext(const uint32_t* p) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    return Ext{{v.x, v.y, v.z, v.w}};
}
ZK_D void st_ext(uint32_t* p, const Ext& e) { *reinterpret_cast<uint4*>(p) = make_uint4(e.c[0], e.c[1], e.c[2], e.c[3]); }

// sum an extension element over the `width` lanes of a lane group (width power of two <= 64)
ZK_D Ext group_sum(Ext v, int width) {
    for (int off = width >> 1; off > 0; off >>= 1) {
        Ext o;
        o.c[0] = __shfl_down(v.c[0], off, width);
        o.c[1] = __shfl_down(v.c[1], off, width);
        o.c[2] = __shfl_down(v.c[2], off, width);
        o.c[3] = __shfl_down(v.c[3], off, width);
        v = Ext{{dadd(v.c[0], o.c[0]), dadd(v.c[1], o.c[1]), dadd(v.c[2], o.c[2]), dadd(v.c[3], o.c[3])}};
    }
    return v;
}

// ------------------------------------------------------------------ domain tables
// LDE domain g <w_M>, M = 2^(log_n + log_blowup), bit-reversed: x_p = g w_M^bitrev(p).  The quotient domain
// g <w_2N> is its first 2N positions (and x_p there is g w_2N^bitrev_2N(p)), so the selector tables have 2N entries.
__device__ __forceinline__ void domain_tables_kernel_body(uint32_t* xs, uint32_t* sel_first, uint32_t* sel_last, uint32_t* itw, int log_n, int log_blowup, uint32_t g_pow_n) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const int H = log_n + log_blowup, lq = log_blowup < 2 ? log_blowup : 2, Hq = log_n + lq;   // selectors on the largest quotient domain (2^lq N points)
    if (p >= (1u << H)) return;
    const uint32_t e = __brev(p) >> (32 - H);
    const uint32_t wm = two_adic_generator(H);
    const uint32_t x = fmul(MONTY_GEN, fpow(wm, e));
    xs[p] = x;
    if (p < (1u << Hq)) {
        const uint32_t eq = __brev(p) >> (32 - Hq);
        const uint32_t xn = fmul(g_pow_n, fpow(two_adic_generator(lq), eq & ((1u << lq) - 1u)));   // x^N = g^N w_{2^lq}^eq
        const uint32_t zh = fsub(xn, MONTY_R1);
        sel_first[p] = fmul(zh, finv(fsub(x, MONTY_R1)));
        sel_last[p] = fmul(zh, finv(fsub(x, finv(two_adic_generator(log_n)))));   // Z_H(x) / (x - w_N^-1)
    }
    if (p < (1u << (H - 1))) {
        const uint32_t ei = (H > 1) ? (__brev(p) >> (32 - (H - 1))) : 0u;
        itw[p] = fmul(finv(fpow(wm, ei)), MONTY_INV2);                  // 1 / (2 w_M^bitrev(p))
    }
}
__global__ void domain_tables_kernel(uint32_t* xs, uint32_t* sel_first, uint32_t* sel_last, uint32_t* itw, int log_n, int log_blowup, uint32_t g_pow_n) { domain_tables_kernel_body(xs, sel_first, sel_last, itw, log_n, log_blowup, g_pow_n); }
struct domain_tables_kernel_bargs { uint32_t* xs; uint32_t* sel_first; uint32_t* sel_last; uint32_t* itw; int log_n; int log_blowup; uint32_t g_pow_n; static domain_tables_kernel_bargs make(uint32_t* xs, uint32_t* sel_first, uint32_t* sel_last, uint32_t* itw, int log_n, int log_blowup, uint32_t g_pow_n) { return domain_tables_kernel_bargs{xs, sel_first, sel_last, itw, log_n, log_blowup, g_pow_n}; } };
__global__ void domain_tables_kernel_batch(const domain_tables_kernel_bargs* __restrict__ zk_arr) { const domain_tables_kernel_bargs& zk_b = zk_arr[blockIdx.z]; domain_tables_kernel_body(zk_b.xs, zk_b.sel_first, zk_b.sel_last, zk_b.itw, zk_b.log_n, zk_b.log_blowup, zk_b.g_pow_n); }

hipError_t launch_domain_tables(uint32_t* xs, uint32_t* sel_first, uint32_t* sel_last, uint32_t* itw, int log_n, int log_blowup, hipStream_t s) {
    const uint32_t m = 1u << (log_n + log_blowup);
    const uint32_t gpn = fpow(MONTY_GEN, (uint64_t)1 << log_n);
    ZK_LAUNCH(domain_tables_kernel, domain_tables_kernel_batch, domain_tables_kernel_bargs, dim3((m + 255) / 256), dim3(256), 0, s, xs, sel_first, sel_last, itw, log_n, log_blowup, gpn);
    return hipGetLastError();
}

// 16-byte load with the non-temporal policy: for matrices that are streamed once per kernel (2 GiB LDEs do not fit any cache)
typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
ZK_D uint4 ld_stream(const uint32_t* p) {
    const u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
// ------------------------------------------------------------------ quotient
// acc = sum_k alpha^(K-1-k) C_k(x_p), k = 3 g + type, then * 1/Z_H(x_p).  The host passes, per column
// group, the three extension weights alpha^(K-1-3g-t) and the three constants of the group's constraints
// in Montgomery form (16 words per group).  Output in NATURAL chunk order: chunk (e & 1), row (e >> 1),
// e = bitrev(p).
//
// The transition constraint of row e reads d of row e + 2 (the next row of the trace on the blown-up
// domain), which sits at an unrelated bit-reversed position: giving every row its own lanes would read the
// 2 GiB LDE twice.  So a group of L <= 16 lanes walks a CHAIN e, e+2, ..., e+2(K-1) and keeps the row it
// loaded as "next" for the following step: (K+1)/K row reads per row.  Each lane owns up to four column
// groups, folds its constraints into four 64-bit running sums (dacc2 / dacc1), and the lanes are summed
// with DPP row operations.
ZK_D void dacc1(uint64_t& acc, uint32_t a0, uint32_t b0) {
    const uint64_t t = dmac(a0, b0, acc);
    acc = ((uint64_t)dred((uint32_t)(t >> 32)) << 32) | (uint32_t)t;
}
template <int CTRL>
ZK_D uint32_t dpp_mov(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false); }
// sum over the aligned group of L lanes (L = 1, 2, 4, 8, 16); every lane of the group ends with the total
ZK_D uint32_t row_group_sum(uint32_t v, int L) {
    if (L > 1) v = dadd(v, dpp_mov<0xB1>(v));        // quad_perm [1,0,3,2]
    if (L > 2) v = dadd(v, dpp_mov<0x4E>(v));        // quad_perm [2,3,0,1]
    if (L > 4) v = dadd(v, dpp_mov<0x141>(v));       // row_half_mirror: quads 0<->1, 2<->3 (quads are uniform by now)
    if (L > 8) v = dadd(v, dpp_mov<0x140>(v));       // row_mirror: halves 0<->1
    return v;
}
// LogUp constraints of the in-table lookup pairs, one LANE per point (weights continue after the G main entries: L_q, then T1, T2, T3):
//   sum_q w_q (phi_q ds_q dr_q - (dr_q - ds_q)) + F1 (S - sum_q phi_q) + F2 (S' - S - sum_q phi'_q) + F3 (S - cumsum),
//   F1 = w_Q sel_first, F2 = w_{Q+1} sel_trans, F3 = w_{Q+2} sel_last.
// Until round 5 the chain kernel below evaluated these inline on the first `pairs` lanes of a row's lane group: with 1 .. 4 pairs on 8 or
// 16 lanes a wavefront waited for five to eight serial extension products on a few lanes (a 2^21 x 96 LDE with 3 pairs took 0.82 ms
// against 0.25 without -- profiles/r05_multichip_*).  Here every lane of a wavefront has a point of its own; the chain kernel adds the
// result on lane 0.  The weights are wave-uniform (scalar loads).
__device__ __forceinline__ void logup_addend_kernel_body(const QuotientArgs& a) {
    const int H = a.log_n + 1;
    const uint32_t m = 1u << H;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const uint32_t e = __brev(p) >> (32 - H);
    const uint32_t pn = __brev((e + 2) & (m - 1)) >> (32 - H);
    const uint32_t* row = a.lde + (uint64_t)p * a.ld;
    const uint32_t* prow = a.perm + (uint64_t)p * a.perm_ld;
    const uint32_t* pnrow = a.perm + (uint64_t)pn * a.perm_ld;
    const uint32_t* wl = a.alpha_pow + 16 * (a.width / 4);          // [Q + 3] ext weights
    const uint32_t sel_trans = dsub(a.xs[p], a.wn_inv);
    Ext r = ext_zero(), sphi = ext_zero(), sphin = ext_zero();
    for (uint32_t q = 0; q < a.pairs; q++) {
        const uint4 vs = *reinterpret_cast<const uint4*>(row + 8 * q);
        const uint4 vr = *reinterpret_cast<const uint4*>(row + 8 * q + 4);
        const Ext ds = ext_add(ext_add_base(a.gamma, vs.x), ext_mul_base_dev(a.beta, vs.y));
        const Ext dr = ext_add(ext_add_base(a.gamma, vr.x), ext_mul_base_dev(a.beta, vr.y));
        const Ext phi = ld_ext(prow + 4 * q), phin = ld_ext(pnrow + 4 * q);
        const Ext c = ext_sub(ext_mul_dev(ext_mul_dev(phi, ds), dr), ext_sub(dr, ds));
        r = ext_add(r, ext_mul_dev(c, ld_ext(wl + 4 * q)));
        sphi = ext_add(sphi, phi);
        sphin = ext_add(sphin, phin);
    }
    const Ext S = ld_ext(prow + 4 * a.pairs), Sn = ld_ext(pnrow + 4 * a.pairs);
    const Ext F1 = ext_mul_base_dev(ld_ext(wl + 4 * a.pairs), a.sel_first[p]);
    const Ext F2 = ext_mul_base_dev(ld_ext(wl + 4 * (a.pairs + 1)), sel_trans);
    const Ext F3 = ext_mul_base_dev(ld_ext(wl + 4 * (a.pairs + 2)), a.sel_last[p]);
    r = ext_add(r, ext_mul_dev(F1, ext_sub(S, sphi)));
    r = ext_add(r, ext_mul_dev(F2, ext_sub(ext_sub(Sn, S), sphin)));
    r = ext_add(r, ext_mul_dev(F3, ext_sub(S, a.cumsum)));
    st_ext(a.addend_out + 4 * (uint64_t)p, r);
}
__global__ void __launch_bounds__(256) logup_addend_kernel(QuotientArgs a) { logup_addend_kernel_body(a); }
struct logup_addend_kernel_bargs { QuotientArgs a; static logup_addend_kernel_bargs make(QuotientArgs a) { return logup_addend_kernel_bargs{a}; } };
__global__ void __launch_bounds__(256) logup_addend_kernel_batch(const logup_addend_kernel_bargs* __restrict__ zk_arr) { const logup_addend_kernel_bargs& zk_b = zk_arr[blockIdx.z]; logup_addend_kernel_body(zk_b.a); }

hipError_t launch_logup_addend(const QuotientArgs& a, hipStream_t s) {
    const uint64_t m = 2ull << a.log_n;
    ZK_LAUNCH(logup_addend_kernel, logup_addend_kernel_batch, logup_addend_kernel_bargs, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
constexpr int QCHAIN = 16;
template <int NG>
#ifndef QWPE
#define QWPE 2
#endif
__device__ __forceinline__ void quotient_kernel_body(const QuotientArgs& a) {
    const int L = a.lanes_per_row;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = gid % L;
    const int H = a.log_n + 1;
    const uint32_t m = 1u << H;
    const uint32_t nchains = m / QCHAIN;
    const bool live = gid / L < nchains;              // dead lanes compute chain 0 and store nothing (DPP needs them)
    const uint32_t chain = live ? gid / L : 0u;
    const uint32_t parity = chain & 1u;
    uint32_t e = 2u * ((chain >> 1) * QCHAIN) + parity;
    const uint32_t G = a.width / 4;
    const uint32_t inv_zh = parity ? a.inv_zh_odd : a.inv_zh_even;

    uint4 cur[NG], nxt[NG];
    uint32_t p = __brev(e) >> (32 - H);
#pragma unroll
    for (int t = 0; t < NG; t++) {
        const uint32_t g = lane + L * t;
        cur[t] = g < G ? *reinterpret_cast<const uint4*>(a.lde + (uint64_t)p * a.ld + 4 * g) : make_uint4(0, 0, 0, 0);
    }
    // the lane's column groups are the same for every row of the chain: up to four groups keep their weights and constants in
    // registers (64 VGPRs); wider rows reload them from L1 per step rather than drop to one wave per SIMD
    constexpr bool HOIST = NG <= 4;
    uint4 W[HOIST ? NG : 1][4];
    if (HOIST) {
#pragma unroll
        for (int t = 0; t < NG; t++) {
            const uint32_t g = lane + L * t;
            const uint4* wp = reinterpret_cast<const uint4*>(a.alpha_pow + 16 * (g < G ? g : 0u));
#pragma unroll
            for (int i = 0; i < 4; i++) W[HOIST ? t : 0][i] = wp[i];
        }
    }
    for (int step = 0; step < QCHAIN; step++) {
        const uint32_t en = (e + 2) & (m - 1);
        const uint32_t pn = __brev(en) >> (32 - H);
#pragma unroll
        for (int t = 0; t < NG; t++) {
            const uint32_t g = lane + L * t;
            nxt[t] = g < G ? ld_stream(a.lde + (uint64_t)pn * a.ld + 4 * g) : make_uint4(0, 0, 0, 0);
        }
        const uint32_t x = a.xs[p];
        const uint32_t sel_first = a.sel_first[p];
        const uint32_t sel_trans = dsub(x, a.wn_inv);
        uint64_t acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < NG; t++) {
            const uint32_t g = lane + L * t;
            if (g < G) {
                const uint4 v = cur[t];
                const uint4* wp = reinterpret_cast<const uint4*>(a.alpha_pow + 16 * g);
                const uint4 w0 = HOIST ? W[HOIST ? t : 0][0] : wp[0], w1 = HOIST ? W[HOIST ? t : 0][1] : wp[1];
                const uint4 w2 = HOIST ? W[HOIST ? t : 0][2] : wp[2], kk = HOIST ? W[HOIST ? t : 0][3] : wp[3];   // kk = (g+1, 2g+3, 5g+7, -) in Montgomery form
                // c1 = c - a a b - k1 ; c2 = sel_trans (d' - a b - c - k2) ; c3 = sel_first (d - d0)
                const uint32_t aab = dmul(dmont_lazy(v.x, v.x), v.y);
                const uint32_t c1 = dsub(dsub(v.z, aab), kk.x);
                const uint32_t ab = dmul(v.x, v.y);
                const uint32_t c2 = dmul(dsub_lazy(dsub(dsub(nxt[t].w, ab), v.z), kk.y), sel_trans);
                const uint32_t c3 = dmul(dsub_lazy(v.w, kk.z), sel_first);
                dacc2(acc[0], w0.x, c1, w1.x, c2); dacc1(acc[0], w2.x, c3);
                dacc2(acc[1], w0.y, c1, w1.y, c2); dacc1(acc[1], w2.y, c3);
                dacc2(acc[2], w0.z, c1, w1.z, c2); dacc1(acc[2], w2.z, c3);
                dacc2(acc[3], w0.w, c1, w1.w, c2); dacc1(acc[3], w2.w, c3);
            }
        }
        Ext r = Ext{{dacc_finish(acc[0]), dacc_finish(acc[1]), dacc_finish(acc[2]), dacc_finish(acc[3])}};
        if (a.addend && lane == 0) r = ext_add(r, ld_ext(a.addend + 4 * (uint64_t)p));     // the LogUp constraints of this point (logup_addend_kernel)
#pragma unroll
        for (int i = 0; i < 4; i++) r.c[i] = row_group_sum(r.c[i], L);
        if (lane == 0 && live) {
            r = ext_mul_base_dev(r, inv_zh);
            st_ext(a.out + ((uint64_t)parity * (m >> 1) + (e >> 1)) * 4, r);
            if (a.lde_out) st_ext(a.lde_out + (uint64_t)p * a.lde_ld + 4u * parity, r);
        }
#pragma unroll
        for (int t = 0; t < NG; t++) cur[t] = nxt[t];
        e = en; p = pn;
    }
}
template <int NG>
#ifndef QWPE
#define QWPE 2
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(QWPE, QWPE))) quotient_kernel(QuotientArgs a) { quotient_kernel_body<NG>(a); }
struct quotient_kernel_bargs { QuotientArgs a; static quotient_kernel_bargs make(QuotientArgs a) { return quotient_kernel_bargs{a}; } };
template <int NG>
#ifndef QWPE
#define QWPE 2
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(QWPE, QWPE))) quotient_kernel_batch(const quotient_kernel_bargs* __restrict__ zk_arr) { const quotient_kernel_bargs& zk_b = zk_arr[blockIdx.z]; quotient_kernel_body<NG>(zk_b.a); }

hipError_t launch_quotient(const QuotientArgs& a, hipStream_t s) {
    const uint64_t m = 2ull << a.log_n;
    const uint64_t threads = (m / QCHAIN) * a.lanes_per_row;
    const uint32_t G = a.width / 4;
    const int ng = (int)((G + a.lanes_per_row - 1) / a.lanes_per_row);
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    switch (ng) {
        case 1: ZK_LAUNCH(quotient_kernel<1>, quotient_kernel_batch<1>, quotient_kernel_bargs, grid, block, 0, s, a); break;
        case 2: ZK_LAUNCH(quotient_kernel<2>, quotient_kernel_batch<2>, quotient_kernel_bargs, grid, block, 0, s, a); break;
        case 3: ZK_LAUNCH(quotient_kernel<3>, quotient_kernel_batch<3>, quotient_kernel_bargs, grid, block, 0, s, a); break;
        case 4: ZK_LAUNCH(quotient_kernel<4>, quotient_kernel_batch<4>, quotient_kernel_bargs, grid, block, 0, s, a); break;
        case 5: case 6: case 7: case 8: ZK_LAUNCH(quotient_kernel<8>, quotient_kernel_batch<8>, quotient_kernel_bargs, grid, block, 0, s, a); break;
        default: ZK_LAUNCH(quotient_kernel<16>, quotient_kernel_batch<16>, quotient_kernel_bargs, grid, block, 0, s, a); break;    // width <= 1024
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ quotient of a constraint program (the AIR as data)
// One thread per point of the quotient domain interprets the program (air.h): every lane of a wave executes the same instruction
// stream, so the program words and the constraint weights are wave-uniform (scalar loads) and only the trace values differ per
// lane.  A lane reads its own row and the "next" row (row e + 2 of the blown-up domain, an unrelated bit-reversed position) word by
// word; consecutive lanes hold consecutive bit-reversed positions, i.e. adjacent 4 * ld-byte rows, and a row's lines stay in L1 / L2
// across the variables of a program.  This is the generic path: the synthetic AIR keeps its specialised kernel (quotient_kernel), which
// streams rows once with 16-byte loads; bytes per point here are the same 2 * 4 * width, the instruction count is what differs.
__device__ __forceinline__ void quotient_air_kernel_body(const QuotientAirArgs& a) {
    const int H = a.log_n + a.log_qd;
    const uint32_t m = 1u << H, nq = 1u << a.log_qd;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const uint32_t e = __brev(p) >> (32 - H);
    const uint32_t pn = __brev((e + nq) & (m - 1)) >> (32 - H);       // the next trace row is 2^log_qd points further on the quotient domain
    const uint32_t* local = a.lde + (uint64_t)p * a.ld;
    const uint32_t* next = a.lde + (uint64_t)pn * a.ld;
    const uint32_t x = a.xs[p];
    const uint32_t sel_first = a.sel_first[p], sel_last = a.sel_last[p], sel_trans = dsub(x, a.wn_inv);
    uint64_t acc[4] = {0, 0, 0, 0};
    const uint32_t* w = a.body;
    for (uint32_t k = 0; k < a.n_constraints; k++) {
        const uint32_t sel = *w++, nt = *w++;
        uint32_t c = 0;
        for (uint32_t t = 0; t < nt; t++) {
            uint32_t prod = *w++;
            const uint32_t d = *w++;
            for (uint32_t j = 0; j < d; j++) {
                const uint32_t v = *w++, kind = v >> 30, idx = v & 0xFFFFu;
                const uint32_t val = kind == 0 ? local[idx] : (kind == 1 ? next[idx] : a.pub[idx]);
                prod = dmul(prod, val);
            }
            c = dadd(c, prod);
        }
        if (sel) c = dmul(c, sel == 1 ? sel_first : (sel == 2 ? sel_last : sel_trans));     // sel is wave-uniform
        const uint4 wt = *reinterpret_cast<const uint4*>(a.weights + 4 * (uint64_t)k);
        dacc1(acc[0], wt.x, c); dacc1(acc[1], wt.y, c); dacc1(acc[2], wt.z, c); dacc1(acc[3], wt.w, c);
    }
    const uint32_t chunk = e & (nq - 1u);
    Ext r = Ext{{dacc_finish(acc[0]), dacc_finish(acc[1]), dacc_finish(acc[2]), dacc_finish(acc[3])}};
    if (a.addend) r = ext_add(r, ld_ext(a.addend + 4 * (uint64_t)p));
    const uint32_t iz = chunk == 0 ? a.inv_zh[0] : (chunk == 1 ? a.inv_zh[1] : (chunk == 2 ? a.inv_zh[2] : a.inv_zh[3]));
    r = ext_mul_base_dev(r, iz);
    st_ext(a.out + ((uint64_t)chunk * (m >> a.log_qd) + (e >> a.log_qd)) * 4, r);
    if (a.lde_out) st_ext(a.lde_out + (uint64_t)p * a.lde_ld + 4u * chunk, r);
}
__global__ void __launch_bounds__(256) quotient_air_kernel(QuotientAirArgs a) { quotient_air_kernel_body(a); }
struct quotient_air_kernel_bargs { QuotientAirArgs a; static quotient_air_kernel_bargs make(QuotientAirArgs a) { return quotient_air_kernel_bargs{a}; } };
__global__ void __launch_bounds__(256) quotient_air_kernel_batch(const quotient_air_kernel_bargs* __restrict__ zk_arr) { const quotient_air_kernel_bargs& zk_b = zk_arr[blockIdx.z]; quotient_air_kernel_body(zk_b.a); }

// The term-parallel form: a workgroup owns a GROUP of 8 points of the quotient domain and its lanes split the TERMS of the flattened
// program (air.h, air_term_records: one record per distinct monomial).  The 8 points are 8 consecutive rows of the trace domain on one
// coset, so 9 rows of the LDE serve them (the next row of point q is the local row of point q + 1); the rows are staged into LDS once,
// 16 bytes per lane and row (kernels.h, AIR_GP: column groups of four, the 9 rows of a group 4 words apart), together with the points'
// selector values, the constant 1 and the public values.  A lane loads its 32-byte record once and applies it to all 8 points (a factor
// of the 8 points: 8 LDS words 4 apart; the record table stays in L2), and the extension coefficients ride in four 64-bit running sums
// per point (dacc2: terms go in pairs, one conditional subtraction per two products, one Montgomery reduction at the end).  The lanes'
// partial sums are then added through LDS.  Against the row-per-lane interpreter above: no strided global gathers (a lane there
// touches 64 cache lines per load instruction) and ~8 x less program traffic per point; 40 - 50 x faster on the 608-column SHA-256
// chip.
// first natural index of group g: cosets interleave (groups g, g + 1 are the same 8 trace rows on neighbouring cosets)
__device__ __forceinline__ uint32_t air_group_e0(const QuotientAirArgs& a, uint32_t g) { return ((g >> a.log_qd) * 8u << a.log_qd) + (g & ((1u << a.log_qd) - 1u)); }
__device__ __forceinline__ uint32_t air_row_of(const QuotientAirArgs& a, uint32_t e0, uint32_t r) {      // LDE row of the group's r-th trace row
    const int H = a.log_n + a.log_qd;
    return __brev((e0 + (r << a.log_qd)) & ((1u << H) - 1u)) >> (32 - H);
}
__device__ __forceinline__ void air_load8(const uint32_t* slots, uint32_t base, uint32_t (&v)[8]) {
    const uint32_t* b = slots + base;
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = b[4 * q];
}
// selector / constant / public-value slots of the group's 8 points
template <int NT>
__device__ __forceinline__ void air_stage_extras(const QuotientAirArgs& a, uint32_t* slots, uint32_t e0, uint32_t tid) {
    const uint32_t W4 = a.width >> 2;
    if (tid < 8) {
        const uint32_t p = air_row_of(a, e0, tid);
        const uint4 ex = {a.sel_first[p], a.sel_last[p], dsub(a.xs[p], a.wn_inv), MONTY_R1};
        *reinterpret_cast<uint4*>(slots + AIR_GP * W4 + 4u * tid) = ex;      // slots 2W .. 2W + 3: the first group after the columns
    }
    for (uint32_t i = tid >> 3; i < a.n_public; i += NT / 8) slots[air_lds_base(2 * a.width + AIR_SLOT_EXTRA + i, a.width) + 4u * (tid & 7u)] = a.pub[i];
}
__device__ __forceinline__ void air_store_point(const QuotientAirArgs& a, uint32_t e, Ext r) {
    const int H = a.log_n + a.log_qd;
    const uint32_t m = 1u << H, nq = 1u << a.log_qd;
    const uint32_t p = __brev(e) >> (32 - H);
    if (a.addend) r = ext_add(r, ld_ext(a.addend + 4 * (uint64_t)p));
    const uint32_t chunk = e & (nq - 1u);
    const uint32_t iz = chunk == 0 ? a.inv_zh[0] : (chunk == 1 ? a.inv_zh[1] : (chunk == 2 ? a.inv_zh[2] : a.inv_zh[3]));
    r = ext_mul_base_dev(r, iz);
    st_ext(a.out + ((uint64_t)chunk * (m >> a.log_qd) + (e >> a.log_qd)) * 4, r);
    if (a.lde_out) st_ext(a.lde_out + (uint64_t)p * a.lde_ld + 4u * chunk, r);
}
// One group of 8 points per workgroup of NT lanes (64, 128 or 256: one, two or four wavefronts).  What limits this kernel is not
// arithmetic but how many groups a CU holds while their rows are on the way (PMC on the first form, 256 lanes and 16 staged rows per
// group: SQ_WAIT_ANY 65 - 86 % of the wave cycles): a group's footprint is its 9 rows in LDS (23 KB at 608 columns), so the FEWER lanes
// a group takes the more groups are resident -- seven single wavefronts per CU against three or four 256-lane workgroups.  Lanes beyond
// one wavefront only pay when a program has enough terms per point to keep them busy; the launcher picks NT from the record count.
template <int NT>
__device__ __forceinline__ void quotient_air_terms_kernel_body(const QuotientAirArgs& a) {
    constexpr int PTS = 8;
    static_assert(NT == 64 || NT == 128 || NT == 256, "one, two or four wavefronts per group");
    extern __shared__ uint32_t slots[];
    const uint32_t tid = threadIdx.x, W4 = a.width >> 2;
    const uint32_t mask = (1u << (a.log_n + a.log_qd)) - 1u;
    const uint32_t e0 = air_group_e0(a, blockIdx.x);
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const uint4* row = reinterpret_cast<const uint4*>(a.lde + (uint64_t)air_row_of(a, e0, r) * a.ld);
        for (uint32_t cg = tid; cg < W4; cg += NT) *reinterpret_cast<uint4*>(slots + AIR_GP * cg + 4u * r) = row[cg];
    }
    air_stage_extras<NT>(a, slots, e0, tid);
    __syncthreads();
    uint64_t acc[PTS][4];
#pragma unroll
    for (int q = 0; q < PTS; q++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[q][i] = 0;
    const uint4* recs = reinterpret_cast<const uint4*>(a.recs);
    auto load8 = [&](uint32_t o, uint32_t (&v)[PTS]) { air_load8(slots, o, v); };
    auto product = [&](const uint4& o, uint32_t (&prod)[PTS]) {
        const uint32_t n = o.z >> 16;
        uint32_t v[PTS];
        load8(o.x & 0xFFFFu, prod);
        const uint32_t offs[4] = {o.x >> 16, o.y & 0xFFFFu, o.y >> 16, o.z & 0xFFFFu};
        for (uint32_t k = 1; k < n; k++) {
            load8(offs[k - 1], v);
#pragma unroll
            for (int q = 0; q < PTS; q++) prod[q] = dmul(prod[q], v[q]);
        }
    };
    // terms go in PAIRS (the table is padded to an even count): two products share one conditional subtraction per running sum (dacc2)
    for (uint32_t t = 2 * tid; t < a.n_terms; t += 2 * NT) {
        const uint4 ca = recs[2 * (size_t)t], oa = recs[2 * (size_t)t + 1], cb = recs[2 * (size_t)t + 2], ob = recs[2 * (size_t)t + 3];
        uint32_t pa[PTS], pb[PTS];
        product(oa, pa);
        product(ob, pb);
#pragma unroll
        for (int q = 0; q < PTS; q++) {
            dacc2(acc[q][0], ca.x, pa[q], cb.x, pb[q]); dacc2(acc[q][1], ca.y, pa[q], cb.y, pb[q]);
            dacc2(acc[q][2], ca.z, pa[q], cb.z, pb[q]); dacc2(acc[q][3], ca.w, pa[q], cb.w, pb[q]);
        }
    }
    __syncthreads();                                   // the slots are dead: the same LDS now carries 32 values x NT partials, row pitch NT + 1
    constexpr int PITCH = NT + 1, SL = NT / 32;        // SL lanes per value in the first round
#pragma unroll
    for (int q = 0; q < PTS; q++)
#pragma unroll
        for (int i = 0; i < 4; i++) slots[(size_t)(q * 4 + i) * PITCH + tid] = dacc_finish(acc[q][i]);
    __syncthreads();
    {
        const uint32_t j = tid / SL, sl = tid % SL;
        uint32_t sum = 0;
        for (uint32_t i = sl; i < (uint32_t)NT; i += SL) sum = dadd(sum, slots[(size_t)j * PITCH + i]);
        __syncthreads();
        slots[(size_t)j * PITCH + sl] = sum;
    }
    __syncthreads();
    if (tid < (uint32_t)PTS) {
        Ext r;
        for (int i = 0; i < 4; i++) {
            uint32_t sum = 0;
            for (int k = 0; k < SL; k++) sum = dadd(sum, slots[(size_t)(tid * 4 + i) * PITCH + k]);
            r.c[i] = sum;
        }
        air_store_point(a, (e0 + (tid << a.log_qd)) & mask, r);
    }
}
template <int NT>
__global__ void __launch_bounds__(NT) quotient_air_terms_kernel(QuotientAirArgs a) { quotient_air_terms_kernel_body<NT>(a); }
struct quotient_air_terms_kernel_bargs { QuotientAirArgs a; static quotient_air_terms_kernel_bargs make(QuotientAirArgs a) { return quotient_air_terms_kernel_bargs{a}; } };
template <int NT>
__global__ void __launch_bounds__(NT) quotient_air_terms_kernel_batch(const quotient_air_terms_kernel_bargs* __restrict__ zk_arr) { const quotient_air_terms_kernel_bargs& zk_b = zk_arr[blockIdx.z]; quotient_air_terms_kernel_body<NT>(zk_b.a); }

// The WIDE form, for term-heavy programs (SHA-256 chip: 3 366 records).  The kernel above gives a lane a RECORD and 8 points: its 8
// LDS reads per factor hit random banks (every lane another column), and that, not arithmetic, is its time (the four SIMDs of a CU
// share one LDS: ~1 000 clk per record and wavefront, of which ~400 are VALU).  Here a lane owns a POINT and the whole wavefront walks
// the SAME record:
//   * a workgroup stages 65 consecutive trace rows of one coset as a COLUMN-major tile (pitch 65 words): lane p reads column c of its
//     point at word 65 c + p, its next row at 65 c + p + 1 -- consecutive banks across the wavefront, one ds_read_b32 per factor for
//     64 points (the form above: 32 reads for 64 points, conflicting); the selector values, the constant 1 and the public values are
//     further columns of the tile, so a factor is an LDS word whatever its kind;
//   * the records are wave-uniform: scalar loads, no vector memory traffic for the program at all, coefficients as SGPR operands;
//   * records come grouped by their number of factors (air_term_records_wide): one branch-free loop per class, so that the reads of
//     several records are in flight under the products of the previous ones -- with branches on the factor count the same kernel was
//     latency-bound at 8.7 ms per 2^21 points of the SHA-256 chip;
//   * the NW wavefronts of the workgroup (8 or 16: one tile fills the LDS, so the workgroup IS the CU's occupancy) split the records
//     and add their four 64-bit sums through LDS at the end.
// w4_recip = ceil(2^32 / (width / 4)).
typedef uint32_t air_u32x16 __attribute__((ext_vector_type(16)));      // a pair of records: one 64-byte scalar load
template <int N>
__device__ __forceinline__ uint32_t air_wide_product(const uint32_t* mine, uint32_t o0, uint32_t o1, uint32_t o2) {
    uint32_t v[5];                                      // all reads first, then the products
    v[0] = mine[o0 & 0xFFFFu];
    if (N > 1) v[1] = mine[o0 >> 16];
    if (N > 2) v[2] = mine[o1 & 0xFFFFu];
    if (N > 3) v[3] = mine[o1 >> 16];
    if (N > 4) v[4] = mine[o2 & 0xFFFFu];
    uint32_t prod = v[0];
#pragma unroll
    for (int k = 1; k < N; k++) prod = dmul(prod, v[k]);
    return prod;
}
// PAIRS pairs of records (a pair: coefficients a, offsets a, coefficients b, offsets b = 64 contiguous bytes; records go in pairs:
// dacc2) starting at r: all their LDS reads go out together, then the products, then the sums
template <int N, int PAIRS>
__device__ __forceinline__ void air_wide_trip(const uint32_t* mine, const uint4* r, uint64_t (&acc)[4]) {
    air_u32x16 rec[PAIRS];
#pragma unroll
    for (int i = 0; i < PAIRS; i++) rec[i] = reinterpret_cast<const air_u32x16*>(r)[i];          // wave-uniform address: scalar loads
    uint32_t pa[PAIRS], pb[PAIRS];
#pragma unroll
    for (int i = 0; i < PAIRS; i++) {
        pa[i] = air_wide_product<N>(mine, rec[i][4], rec[i][5], rec[i][6]);
        pb[i] = air_wide_product<N>(mine, rec[i][12], rec[i][13], rec[i][14]);
    }
#pragma unroll
    for (int i = 0; i < PAIRS; i++) {
        dacc2(acc[0], rec[i][0], pa[i], rec[i][8], pb[i]); dacc2(acc[1], rec[i][1], pa[i], rec[i][9], pb[i]);
        dacc2(acc[2], rec[i][2], pa[i], rec[i][10], pb[i]); dacc2(acc[3], rec[i][3], pa[i], rec[i][11], pb[i]);
    }
}
// The wavefront's share of one class: a CONTIGUOUS run of pairs (so that a trip's records are one contiguous block), whole trips of
// four pairs per wavefront (a remainder trip costs a full load latency for a quarter of the work); the classes start their deal at
// different wavefronts, so that the short ends do not all fall on the last one.  ~450 clk of arithmetic per trip against one
// scalar-load latency and one LDS latency: four wavefronts per SIMD cover that without software prefetch (scalar loads share the
// LDS counter and return out of order, so a prefetch would be waited for with the first LDS value anyway).
// Tried and slower: branches on the factor count inside one loop (no load can move above a branch: 8.7 ms per 2^21 points of the
// SHA-256 chip); records through broadcast vector loads (64 lanes x 16 bytes occupy the vector memory pipe like any load: 8.1 ms);
// records as one 16-byte load per lane handed round with v_readlane (6.2 ms).
template <int N, int NW>
__device__ __forceinline__ void air_wide_class(const uint32_t* mine, const uint4* recs, uint32_t begin, uint32_t end, uint32_t wave, uint32_t same, uint64_t (&acc)[4]) {
    const uint32_t pairs = (end - begin) >> 1, per = ((pairs + 4u * NW - 1u) / (4u * NW)) * 4u;
    const uint32_t slot = (wave + 3u * N) % NW;
    uint32_t p = slot * per;
    const uint32_t pe = p + per < pairs ? p + per : pairs;
    const uint4* r = recs + 2 * (size_t)begin + 4 * (size_t)(same ? 0u : p);          // (A/B: every wavefront walks the first run)
    for (; p + 4u <= pe; p += 4u, r += 16) air_wide_trip<N, 4>(mine, r, acc);
    if (p + 2u <= pe) { air_wide_trip<N, 2>(mine, r, acc); p += 2u; r += 8; }
    if (p < pe) air_wide_trip<N, 1>(mine, r, acc);
}
// (A chained variant -- a workgroup walking 16 consecutive tiles, the next tile's rows prefetched into registers under the records of
// the current one, as quotient_air_chain_kernel does -- was built and measured: with 1 024 lanes per workgroup the 40 prefetch
// registers do not fit beside the term loop's, it spills, 6.5 ms.  Staging stays un-overlapped: ~0.65 of the 5.0 ms.)
template <int NW>
__device__ __forceinline__ void quotient_air_wide_kernel_body(const QuotientAirArgs& a, uint32_t w4_recip) {
    constexpr uint32_t NT = 64u * NW, PITCH = AIR_WIDE_PITCH, ROWS = AIR_WIDE_POINTS + 1u;
    extern __shared__ uint32_t tile[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t W = a.width, W4 = W >> 2;
    const uint32_t mask = (1u << (a.log_n + a.log_qd)) - 1u, nq = 1u << a.log_qd;
    const uint32_t coset = blockIdx.x & (nq - 1u), blk = blockIdx.x >> a.log_qd;
    const uint32_t e0 = ((blk * AIR_WIDE_POINTS) << a.log_qd) + coset;
    // ---- the tile: 65 rows x W columns, transposed on the way in (a lane moves 16 bytes = 4 columns of one row)
    const uint32_t total = ROWS * W4;
#ifdef ZKHIP_AB_HOOKS
    if (!(a.no_chain & 2u))                              // A/B: no staging (the tile holds whatever the LDS held)
#endif
#pragma unroll 4
    for (uint32_t idx = tid; idx < total; idx += NT) {
        const uint32_t r = __umulhi(idx, w4_recip);              // idx / W4 (exact: idx < 2^16)
        const uint32_t cg = idx - r * W4;
        const uint4 v = ld_stream(a.lde + (uint64_t)air_row_of(a, e0, r) * a.ld + 4u * cg);
        uint32_t* d = tile + 4u * cg * PITCH + r;
        d[0] = v.x; d[PITCH] = v.y; d[2 * PITCH] = v.z; d[3 * PITCH] = v.w;
    }
    // ---- the extra columns: is_first, is_last, is_transition, 1, then the public values
    if (tid < ROWS) {
        const uint32_t p = air_row_of(a, e0, tid);
        uint32_t* d = tile + W * PITCH + tid;
        d[0] = a.sel_first[p]; d[PITCH] = a.sel_last[p]; d[2 * PITCH] = dsub(a.xs[p], a.wn_inv); d[3 * PITCH] = MONTY_R1;
    }
    for (uint32_t idx = tid; idx < a.n_public * ROWS; idx += NT) {
        const uint32_t i = idx / ROWS, r = idx - i * ROWS;
        tile[(W + AIR_SLOT_EXTRA + i) * PITCH + r] = a.pub[i];
    }
    __syncthreads();
    const uint32_t* mine = tile + lane;
    uint64_t acc[4] = {0, 0, 0, 0};
    const uint4* recs = reinterpret_cast<const uint4*>(a.recs);
    const uint32_t* cls = a.cls;
#ifdef ZKHIP_AB_HOOKS
    const uint32_t same = a.no_chain & 8u;               // A/B: all wavefronts read the same records (scalar-cache hits)
    if (!(a.no_chain & 4u)) {                            // A/B: no terms
#else
    constexpr uint32_t same = 0;
#endif
    air_wide_class<1, NW>(mine, recs, cls[0], cls[1], wave, same, acc);
    air_wide_class<2, NW>(mine, recs, cls[1], cls[2], wave, same, acc);
    air_wide_class<3, NW>(mine, recs, cls[2], cls[3], wave, same, acc);
    air_wide_class<4, NW>(mine, recs, cls[3], cls[4], wave, same, acc);
    air_wide_class<5, NW>(mine, recs, cls[4], cls[5], wave, same, acc);
#ifdef ZKHIP_AB_HOOKS
    }
#endif
    __syncthreads();                                   // the tile is dead: the LDS now carries [NW][4][64] partial sums
#pragma unroll
    for (int c = 0; c < 4; c++) tile[(wave * 4u + c) * 64u + lane] = dacc_finish(acc[c]);
    __syncthreads();
    if (tid < AIR_WIDE_POINTS) {
        Ext r;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            uint32_t sum = 0;
            for (uint32_t w = 0; w < (uint32_t)NW; w++) sum = dadd(sum, tile[(w * 4u + c) * 64u + tid]);
            r.c[c] = sum;
        }
        air_store_point(a, (e0 + (tid << a.log_qd)) & mask, r);
    }
}
template <int NW>
__global__ void __launch_bounds__(64 * NW) quotient_air_wide_kernel(QuotientAirArgs a, uint32_t w4_recip) { quotient_air_wide_kernel_body<NW>(a, w4_recip); }
struct quotient_air_wide_kernel_bargs { QuotientAirArgs a; uint32_t w4_recip; static quotient_air_wide_kernel_bargs make(QuotientAirArgs a, uint32_t w4_recip) { return quotient_air_wide_kernel_bargs{a, w4_recip}; } };
template <int NW>
__global__ void __launch_bounds__(64 * NW) quotient_air_wide_kernel_batch(const quotient_air_wide_kernel_bargs* __restrict__ zk_arr) {
    // (by value, unlike the generated twins: the record table's address must end up in SGPRs for the records to be scalar loads)
    const quotient_air_wide_kernel_bargs zk_b = zk_arr[blockIdx.z];
    quotient_air_wide_kernel_body<NW>(zk_b.a, zk_b.w4_recip);
}

// The chained form of the same kernel.  What the kernel above spends its time on is not terms but rows: every workgroup stages 9
// rows, waits for them with nothing else to do, and leaves again (a 608-column program of 8 terms: 4.3 ms per 2^21 points = 1.2 TB/s).
// Here a workgroup walks CHAIN consecutive groups of one coset:
//   * the 9th row of a group is the first row of the next, so every LDE row is fetched ONCE (8 rows per group instead of 9);
//   * the next group's 8 rows are in flight (registers, 16 bytes per lane and load, NPF loads per lane) while the current group's
//     terms are evaluated, and so are its selector values;
//   * the lanes' partial sums are combined by a reduce-scatter through the wavefront's cross-lane network (32 values per lane ->
//     one value per lane pair, 32 shuffles + 32 additions) instead of three barriers around an LDS transpose, so the staged rows
//     stay valid (row 8 becomes row 0 by a copy inside LDS) and a group costs two barriers;
//   * 32 lanes finish one coefficient of one point each (addend, 1 / Z_H, the two stores), 128 contiguous bytes per store.
// idx -> (row, column group) of the lane's k-th prefetch slot: idx = tid + k NT < 8 W4, row = 1 + idx / W4.
template <int NT, int NPF>
__device__ __forceinline__ void quotient_air_chain_kernel_body(const QuotientAirArgs& a, uint32_t chain_len, uint32_t w4_recip) {
    constexpr int PTS = 8;
    static_assert(NT == 64 || NT == 128 || NT == 256, "one, two or four wavefronts per group");
    extern __shared__ uint32_t slots[];
    const uint32_t tid = threadIdx.x, W4 = a.width >> 2, lane = tid & 63u, wave = tid >> 6;
    const uint32_t mask = (1u << (a.log_n + a.log_qd)) - 1u;
    const uint32_t nq = 1u << a.log_qd;
    const uint32_t coset = blockIdx.x & (nq - 1u), blk0 = (blockIdx.x >> a.log_qd) * chain_len;
    const size_t lds_rows_words = (size_t)AIR_GP * (W4 + (AIR_SLOT_EXTRA + ((a.n_public + 3u) & ~3u)) / 4);
    uint32_t* red = slots + lds_rows_words;          // [NT / 64][32] cross-wave partial sums
    // the lane's prefetch slots: LDS word and (row, column group) packed
    uint32_t pf_rc[NPF];
#pragma unroll
    for (int k = 0; k < NPF; k++) {
        // slots past the end repeat the last one (two lanes then move the same 16 bytes to the same place): no lane-dependent
        // branch around a load, which the compiler would serialise behind a wait each
        const uint32_t raw = tid + (uint32_t)k * NT;
        const uint32_t idx = min(raw, 8u * W4 - 1u);
        const uint32_t r = __umulhi(idx, w4_recip);              // idx / W4 (exact below 2^16)
        const uint32_t cg = idx - r * W4;
        pf_rc[k] = (raw < 8u * W4 ? 0u : 0x80000000u) | ((r + 1u) << 16) | cg;      // top bit: a repeat, loaded but not stored
    }
    uint32_t e0 = air_group_e0(a, (blk0 << a.log_qd) | coset);
    // the first group: all 9 rows, as the unchained kernel
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const uint4* row = reinterpret_cast<const uint4*>(a.lde + (uint64_t)air_row_of(a, e0, r) * a.ld);
        for (uint32_t cg = tid; cg < W4; cg += NT) *reinterpret_cast<uint4*>(slots + AIR_GP * cg + 4u * r) = row[cg];
    }
    air_stage_extras<NT>(a, slots, e0, tid);
    const uint4* recs = reinterpret_cast<const uint4*>(a.recs);
    for (uint32_t i = 0; i < chain_len; i++) {
        const bool more = i + 1 < chain_len;
        const uint32_t e0n = air_group_e0(a, ((blk0 + i + 1u) << a.log_qd) | coset);
        // ---- next group's rows 1..8 and selector values: on their way while this group's terms run
        uint4 pf[NPF];
        uint4 exn = make_uint4(0, 0, 0, 0);
        if (more) {
#pragma unroll
            for (int k = 0; k < NPF; k++) {
                const uint32_t r = (pf_rc[k] >> 16) & 0xFFu, cg = pf_rc[k] & 0xFFFFu;
                pf[k] = ld_stream(a.lde + (uint64_t)air_row_of(a, e0n, r) * a.ld + 4u * cg);
            }
            if (tid < 8) {
                const uint32_t p = air_row_of(a, e0n, tid);
                exn = make_uint4(a.sel_first[p], a.sel_last[p], dsub(a.xs[p], a.wn_inv), MONTY_R1);
            }
        }
        __syncthreads();                               // this group's slots are staged
        uint64_t acc[PTS][4];
#pragma unroll
        for (int q = 0; q < PTS; q++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[q][c] = 0;
        auto product = [&](const uint4& o, uint32_t (&prod)[PTS]) {
            const uint32_t n = o.z >> 16;
            uint32_t v[PTS];
            air_load8(slots, o.x & 0xFFFFu, prod);
            const uint32_t offs[4] = {o.x >> 16, o.y & 0xFFFFu, o.y >> 16, o.z & 0xFFFFu};
            for (uint32_t k = 1; k < n; k++) {
                air_load8(slots, offs[k - 1], v);
#pragma unroll
                for (int q = 0; q < PTS; q++) prod[q] = dmul(prod[q], v[q]);
            }
        };
        for (uint32_t t = 2 * tid; t < a.n_terms; t += 2 * NT) {
            const uint4 ca = recs[2 * (size_t)t], oa = recs[2 * (size_t)t + 1], cb = recs[2 * (size_t)t + 2], ob = recs[2 * (size_t)t + 3];
            uint32_t pa[PTS], pb[PTS];
            product(oa, pa);
            product(ob, pb);
#pragma unroll
            for (int q = 0; q < PTS; q++) {
                dacc2(acc[q][0], ca.x, pa[q], cb.x, pb[q]); dacc2(acc[q][1], ca.y, pa[q], cb.y, pb[q]);
                dacc2(acc[q][2], ca.z, pa[q], cb.z, pb[q]); dacc2(acc[q][3], ca.w, pa[q], cb.w, pb[q]);
            }
        }
        // ---- 32 values per lane -> one per lane pair: reduce-scatter over the wavefront (value v = 4 q + c ends on lanes 2 v, 2 v + 1)
        uint32_t val[32];
#pragma unroll
        for (int q = 0; q < PTS; q++)
#pragma unroll
            for (int c = 0; c < 4; c++) val[4 * q + c] = dacc_finish(acc[q][c]);
#pragma unroll
        for (int half = 16; half >= 1; half >>= 1) {
            const uint32_t bit = 2u * (uint32_t)half;              // lane bit that decides which half a lane keeps: 32, 16, 8, 4, 2
            const bool upper = (lane & bit) != 0;
#pragma unroll
            for (int j = 0; j < half; j++) {
                const uint32_t give = upper ? val[j] : val[half + j];
                const uint32_t keep = upper ? val[half + j] : val[j];
                val[j] = dadd(keep, (uint32_t)__shfl_xor((int)give, (int)bit, 64));
            }
        }
        uint32_t total = dadd(val[0], (uint32_t)__shfl_xor((int)val[0], 1, 64));
        if (NT > 64) {
            if (wave != 0 && (lane & 1u) == 0) red[(wave - 1u) * 32u + (lane >> 1)] = total;
        }
        __syncthreads();                               // every lane is done with this group's slots; the other waves' sums are in LDS
        if (wave == 0 && (lane & 1u) == 0) {
            const uint32_t v = lane >> 1, q = v >> 2, c = v & 3u;
            if (NT > 64) {
#pragma unroll
                for (int w = 1; w < NT / 64; w++) total = dadd(total, red[(w - 1) * 32 + v]);
            }
            const int H = a.log_n + a.log_qd;
            const uint32_t e = (e0 + (q << a.log_qd)) & mask;
            const uint32_t p = __brev(e) >> (32 - H);
            if (a.addend) total = dadd(total, a.addend[4 * (uint64_t)p + c]);
            const uint32_t chunk = e & (nq - 1u);
            const uint32_t iz = chunk == 0 ? a.inv_zh[0] : (chunk == 1 ? a.inv_zh[1] : (chunk == 2 ? a.inv_zh[2] : a.inv_zh[3]));
            total = dmul(total, iz);
            a.out[((uint64_t)chunk * ((1ull << H) >> a.log_qd) + (e >> a.log_qd)) * 4 + c] = total;
            if (a.lde_out) a.lde_out[(uint64_t)p * a.lde_ld + 4u * chunk + c] = total;
        }
        if (more) {
            // row 8 -> row 0, the fetched rows -> rows 1..8, the selector values of the next group's points.  The lane that brings
            // the new row 8 of a column group moves the old one to row 0 first (no other lane touches those words: no barrier)
#pragma unroll
            for (int k = 0; k < NPF; k++) {
                const uint32_t r = (pf_rc[k] >> 16) & 0xFFu, cg = pf_rc[k] & 0xFFFFu;
                if (!(pf_rc[k] >> 31)) {
                    uint4* dst = reinterpret_cast<uint4*>(slots + AIR_GP * cg + 4u * r);
                    if (r == 8u) *reinterpret_cast<uint4*>(slots + AIR_GP * cg) = *dst;
                    *dst = pf[k];
                }
            }
            if (tid < 8) *reinterpret_cast<uint4*>(slots + AIR_GP * W4 + 4u * tid) = exn;
            e0 = e0n;
        }
    }
}
template <int NT, int NPF>
__global__ void __launch_bounds__(NT, 2) quotient_air_chain_kernel(QuotientAirArgs a, uint32_t chain_len, uint32_t w4_recip) { quotient_air_chain_kernel_body<NT, NPF>(a, chain_len, w4_recip); }
struct quotient_air_chain_kernel_bargs { QuotientAirArgs a; uint32_t chain_len; uint32_t w4_recip; static quotient_air_chain_kernel_bargs make(QuotientAirArgs a, uint32_t chain_len, uint32_t w4_recip) { return quotient_air_chain_kernel_bargs{a, chain_len, w4_recip}; } };
template <int NT, int NPF>
__global__ void __launch_bounds__(NT, 2) quotient_air_chain_kernel_batch(const quotient_air_chain_kernel_bargs* __restrict__ zk_arr) { const quotient_air_chain_kernel_bargs& zk_b = zk_arr[blockIdx.z]; quotient_air_chain_kernel_body<NT, NPF>(zk_b.a, zk_b.chain_len, zk_b.w4_recip); }

template <int NT, int NPF>
static hipError_t launch_chain(const QuotientAirArgs& a, uint32_t n_chains, uint32_t chain_len, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024) {
        static std::atomic<size_t> configured[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (lds > configured[dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute((const void*)quotient_air_chain_kernel<NT, NPF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute((const void*)quotient_air_chain_kernel_batch<NT, NPF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            configured[dev].store(lds, std::memory_order_release);
        }
    }
    const uint32_t W4 = a.width >> 2;
    const uint32_t recip = (uint32_t)(((1ull << 32) + W4 - 1) / W4);        // ceil(2^32 / W4): idx / W4 = umulhi(idx, recip) for idx < 2^16
    ZK_LAUNCH((quotient_air_chain_kernel<NT, NPF>), (quotient_air_chain_kernel_batch<NT, NPF>), quotient_air_chain_kernel_bargs, dim3(n_chains), dim3(NT), lds, s, a, chain_len, recip);
    return hipGetLastError();
}
template <int NT>
static hipError_t launch_chain_nt(const QuotientAirArgs& a, uint32_t n_chains, uint32_t chain_len, size_t lds, hipStream_t s) {
    const uint32_t need = (8u * (a.width >> 2) + NT - 1) / NT;               // prefetch loads per lane
    if (need <= 4) return launch_chain<NT, 4>(a, n_chains, chain_len, lds, s);
    if (need <= 8) return launch_chain<NT, 8>(a, n_chains, chain_len, lds, s);
    if (need <= 12) return launch_chain<NT, 12>(a, n_chains, chain_len, lds, s);
    if (need <= 16) return launch_chain<NT, 16>(a, n_chains, chain_len, lds, s);
    return hipErrorInvalidValue;
}
template <int NT>
static hipError_t launch_terms(const QuotientAirArgs& a, uint32_t n_groups, size_t lds_rows, hipStream_t s) {
    const size_t lds_red = (size_t)32 * (NT + 1) * 4, lds = lds_rows > lds_red ? lds_rows : lds_red;
    if (lds > 64 * 1024) {                              // beyond the default dynamic LDS limit: raise it once per device
        static std::atomic<size_t> configured[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (lds > configured[dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute((const void*)quotient_air_terms_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute((const void*)quotient_air_terms_kernel_batch<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            configured[dev].store(lds, std::memory_order_release);
        }
    }
    ZK_LAUNCH(quotient_air_terms_kernel<NT>, quotient_air_terms_kernel_batch<NT>, quotient_air_terms_kernel_bargs, dim3(n_groups), dim3(NT), lds, s, a);
    return hipGetLastError();
}
template <int NW>
static hipError_t launch_wide(const QuotientAirArgs& a, hipStream_t s) {
    const size_t lds_tile = (size_t)AIR_WIDE_PITCH * (a.width + AIR_SLOT_EXTRA + a.n_public) * 4, lds_red = (size_t)NW * 4 * 64 * 4, lds = lds_tile > lds_red ? lds_tile : lds_red;
    if (lds > 64 * 1024) {
        static std::atomic<size_t> configured[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (lds > configured[dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute((const void*)quotient_air_wide_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute((const void*)quotient_air_wide_kernel_batch<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            configured[dev].store(lds, std::memory_order_release);
        }
    }
    const uint32_t W4 = a.width >> 2, recip = (uint32_t)((((uint64_t)1 << 32) + W4 - 1) / W4);
    const uint32_t tiles = (uint32_t)(((uint64_t)1 << a.log_n) / AIR_WIDE_POINTS) << a.log_qd;
    ZK_LAUNCH(quotient_air_wide_kernel<NW>, quotient_air_wide_kernel_batch<NW>, quotient_air_wide_kernel_bargs, dim3(tiles), dim3(64 * NW), lds, s, a, recip);
    return hipGetLastError();
}
hipError_t launch_quotient_air(const QuotientAirArgs& a, hipStream_t s) {
    const uint64_t m = 1ull << (a.log_n + a.log_qd);
    constexpr int PTS = 8;
#ifdef ZKHIP_AB_HOOKS
    if (a.recs && a.wide) {
        static const int abl = [] { const char* e = getenv("ZKHIP_AIRQ_ABL"); return e ? atoi(e) : 0; }();
        if (abl) { QuotientAirArgs b = a; b.no_chain |= (uint32_t)abl; return launch_wide<16>(b, s); }
    }
#endif
    if (a.recs && a.wide) {                              // the host built the records for the wide form (air_wide_form said so)
        if (!air_wide_form(a.width, a.n_terms, a.log_n, a.n_public) || a.ld % 4 != 0 || (reinterpret_cast<uintptr_t>(a.lde) & 15u) != 0 || (a.n_terms & 1u))
            return hipErrorInvalidValue;
        return launch_wide<16>(a, s);
    }
    const uint32_t groups4 = (a.width >> 2) + (AIR_SLOT_EXTRA + ((a.n_public + 3u) & ~3u)) / 4;      // column groups + selector / public groups
    const size_t lds_rows = (size_t)AIR_GP * groups4 * 4;
    // the term-parallel kernel reads rows 16 bytes per lane: rows start on 16-byte boundaries, and a record addresses LDS words in 16 bits
    const bool fits = a.width % 4 == 0 && a.ld % 4 == 0 && (reinterpret_cast<uintptr_t>(a.lde) & 15u) == 0 && a.width <= 1024 &&
                      lds_rows <= 72 * 1024 && m >= ((uint64_t)PTS << a.log_qd);
    // Small programs over rows of at most 16 columns keep the row-per-lane interpreter: a whole row sits in one cache line, its gathers
    // are cheap, and the term-parallel kernel's fixed cost per group of 8 points would be all of its time (tools/airq_time.py).
    const bool small = a.n_terms <= 512 && a.width <= 16;
    if (a.recs && fits && !small && (a.n_terms & 1u) == 0) {
        const uint32_t n_groups = (uint32_t)(m / PTS);
        {   // the chained form: CHAIN consecutive groups of one coset per workgroup (every row fetched once, the next group's rows in
            // flight under the current group's terms).  Lanes per group: enough to keep the prefetch at <= 16 loads per lane, and by
            // the work per point as below.
            const uint32_t blocks = (uint32_t)((1ull << a.log_n) / PTS);        // groups per coset
            uint32_t chain_len = 32;
            while (chain_len > blocks) chain_len >>= 1;
            const uint32_t W4 = a.width >> 2;
            int nt = a.n_terms <= 512 ? 64 : (a.n_terms <= 8192 ? 128 : 256);
            while (nt < 256 && 8u * W4 > (nt == 64 ? 12u : 16u) * (uint32_t)nt) nt *= 2;      // one wavefront holds at most 12 prefetch loads per lane without spilling
            // Term-heavy programs keep the one-group-per-workgroup kernel: their time goes into the terms (record traffic from L2,
            // LDS reads with random banks), which want four waves per SIMD, and the chained form's prefetch registers cost half of
            // that (SHA-256 chip, 3 366 records: 6.3 ms chained against 5.7; Poseidon2 chip, 1 008 records: 2.5 against 2.2).  Light
            // programs are all staging: 8 terms over 608 columns 4.3 -> 1.37 ms per 2^21 points, 360 columns 0.78, 128 columns 0.44.
            const bool light = a.n_terms <= 512;
            if (light && chain_len >= 2 && 8u * W4 <= 16u * (uint32_t)nt && 8u * W4 < 65536u && !a.no_chain) {
                const uint32_t n_chains = (blocks / chain_len) << a.log_qd;
                const size_t lds = lds_rows + (size_t)(nt / 64) * 32 * 4;
                if (nt == 64) return launch_chain_nt<64>(a, n_chains, chain_len, lds, s);
                if (nt == 128) return launch_chain_nt<128>(a, n_chains, chain_len, lds, s);
                return launch_chain_nt<256>(a, n_chains, chain_len, lds, s);
            }
        }
        // lanes per group by the work per point: one wavefront keeps the most groups resident; more only for programs with many records
        // (same box, 2^21 points: the SHA-256 chip's 3 366 records 7.3 / 5.7 / 6.1 ms with 64 / 128 / 256 lanes, the Poseidon2 chip's 1 008
        // records 2.4 / 2.2 ms with 64 / 128; the first form of this kernel, 256 lanes and 16 staged rows: 7.4 and 3.9 ms)
        if (a.n_terms <= 512) return launch_terms<64>(a, n_groups, lds_rows, s);
        if (a.n_terms <= 8192) return launch_terms<128>(a, n_groups, lds_rows, s);
        return launch_terms<256>(a, n_groups, lds_rows, s);
    }
    ZK_LAUNCH(quotient_air_kernel, quotient_air_kernel_batch, quotient_air_kernel_bargs, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ 1 / (x_p - z)
// out[k][p] = 1 / (x_p - z_k) for p < count; optionally xw[k][p] = x_p / (x_p - z_k) for p < xw_count
// (the barycentric weights of the opening kernel, which sums over the first xw_count rows only)
__device__ __forceinline__ void inv_denominators_kernel_body(const uint32_t* xs, uint64_t count, const Ext& z0, const Ext& z1, int npoints, uint32_t* out, uint32_t* xw, uint64_t xw_count) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count) return;
    const uint32_t x = xs[p];
    const Ext d0 = ext_inv_dev(ext_neg(ext_sub_base(z0, x)));
    st_ext(out + 4 * p, d0);
    if (xw && p < xw_count) st_ext(xw + 4 * p, ext_mul_base_dev(d0, x));
    if (npoints > 1) {
        const Ext d1 = ext_inv_dev(ext_neg(ext_sub_base(z1, x)));
        st_ext(out + 4 * (count + p), d1);
        if (xw && p < xw_count) st_ext(xw + 4 * (xw_count + p), ext_mul_base_dev(d1, x));
    }
}
__global__ void __launch_bounds__(256) inv_denominators_kernel(const uint32_t* xs, uint64_t count, Ext z0, Ext z1, int npoints, uint32_t* out, uint32_t* xw, uint64_t xw_count) { inv_denominators_kernel_body(xs, count, z0, z1, npoints, out, xw, xw_count); }
struct inv_denominators_kernel_bargs { const uint32_t* xs; uint64_t count; Ext z0; Ext z1; int npoints; uint32_t* out; uint32_t* xw; uint64_t xw_count; static inv_denominators_kernel_bargs make(const uint32_t* xs, uint64_t count, Ext z0, Ext z1, int npoints, uint32_t* out, uint32_t* xw, uint64_t xw_count) { return inv_denominators_kernel_bargs{xs, count, z0, z1, npoints, out, xw, xw_count}; } };
__global__ void __launch_bounds__(256) inv_denominators_kernel_batch(const inv_denominators_kernel_bargs* __restrict__ zk_arr) { const inv_denominators_kernel_bargs& zk_b = zk_arr[blockIdx.z]; inv_denominators_kernel_body(zk_b.xs, zk_b.count, zk_b.z0, zk_b.z1, zk_b.npoints, zk_b.out, zk_b.xw, zk_b.xw_count); }

hipError_t launch_inv_denominators(const uint32_t* xs, uint64_t count, const Ext& z0, const Ext& z1, int npoints,
                                   uint32_t* out, uint32_t* xw, uint64_t xw_count, hipStream_t s) {
    ZK_LAUNCH(inv_denominators_kernel, inv_denominators_kernel_batch, inv_denominators_kernel_bargs, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, xs, count, z0, z1, npoints, out, xw, xw_count);
    return hipGetLastError();
}

// ------------------------------------------------------------------ barycentric opening
// partial[chunk][pt][col] = sum_{q in chunk} m[q][col] * xw_pt[q].  A thread owns one column and every
// TY-th row of the chunk; the four coefficients of each point are 64-bit running sums (dacc2: two
// products per conditional subtraction, one Montgomery reduction per chunk).
constexpr int OPEN_ROWS = 2048;   // rows per workgroup
constexpr int OPEN_ROWS4 = 1024;  // rows per workgroup of the four-columns-per-lane form
template <int NPTS>
__device__ __forceinline__ void open_partial_kernel_body(const OpenArgs& a) {
    __shared__ uint32_t red[256 * 4];
    const int TX = a.tx, TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const uint32_t col = blockIdx.y * TX + tx;
    const uint64_t r0 = (uint64_t)blockIdx.x * OPEN_ROWS;
    const bool active = col < a.width;
    uint64_t acc[NPTS][4];
#pragma unroll
    for (int k = 0; k < NPTS; k++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[k][i] = 0;
    const int nr = (int)((a.rows - r0) < (uint64_t)OPEN_ROWS ? (a.rows - r0) : (uint64_t)OPEN_ROWS);
    // four rows per trip (r, r + TY, r + 2 TY, r + 3 TY): eight independent loads in flight per lane
    for (int r = ty; r < nr; r += 4 * TY) {
        uint32_t v[4];
        uint64_t q[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool in = r + j * TY < nr;
            q[j] = in ? r0 + r + j * TY : r0 + r;                 // out-of-range slots re-read row r with weight 0
            v[j] = (active && in) ? a.mat[q[j] * a.ld + col] : 0u;
        }
#pragma unroll
        for (int k = 0; k < NPTS; k++) {
            Ext w[4];
#pragma unroll
            for (int j = 0; j < 4; j++) w[j] = ld_ext(a.xw + 4 * ((uint64_t)k * a.xw_stride + q[j]));
#pragma unroll
            for (int i = 0; i < 4; i++) {
                dacc2(acc[k][i], v[0], w[0].c[i], v[1], w[1].c[i]);
                dacc2(acc[k][i], v[2], w[2].c[i], v[3], w[3].c[i]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NPTS; k++) {
        const Ext mine = Ext{{dacc_finish(acc[k][0]), dacc_finish(acc[k][1]), dacc_finish(acc[k][2]), dacc_finish(acc[k][3])}};
        __syncthreads();
        for (int i = 0; i < 4; i++) red[threadIdx.x * 4 + i] = mine.c[i];
        __syncthreads();
        if (ty == 0 && active) {
            Ext sum = mine;
            for (int y = 1; y < TY; y++) {
                Ext o = Ext{{red[(y * TX + tx) * 4], red[(y * TX + tx) * 4 + 1], red[(y * TX + tx) * 4 + 2], red[(y * TX + tx) * 4 + 3]}};
                sum = ext_add(sum, o);
            }
            st_ext(a.partial + 4 * (((uint64_t)blockIdx.x * NPTS + k) * a.width + col), sum);
        }
    }
}
template <int NPTS>
__global__ void __launch_bounds__(256) open_partial_kernel(OpenArgs a) { open_partial_kernel_body<NPTS>(a); }
struct open_partial_kernel_bargs { OpenArgs a; static open_partial_kernel_bargs make(OpenArgs a) { return open_partial_kernel_bargs{a}; } };
template <int NPTS>
__global__ void __launch_bounds__(256) open_partial_kernel_batch(const open_partial_kernel_bargs* __restrict__ zk_arr) { const open_partial_kernel_bargs& zk_b = zk_arr[blockIdx.z]; open_partial_kernel_body<NPTS>(zk_b.a); }

// The same for matrices whose width and pitch are multiples of 4: a lane owns FOUR adjacent columns (16-byte loads: a wave
// covers a whole 1 KiB row of a 256-column matrix) and the weights of a row are fetched once per four columns.
template <int NPTS>
__device__ __forceinline__ void open_partial4_kernel_body(const OpenArgs& a) {
    __shared__ uint32_t red[256 * 16];
    const int TX = a.tx, TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const uint32_t col = (blockIdx.y * TX + tx) * 4;
    const uint64_t r0 = (uint64_t)blockIdx.x * OPEN_ROWS4;
    const bool active = col < a.width;
    uint64_t acc[NPTS][4][4];                       // [point][extension coefficient][column]
#pragma unroll
    for (int k = 0; k < NPTS; k++)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[k][i][c] = 0;
    const int nr = (int)((a.rows - r0) < (uint64_t)OPEN_ROWS4 ? (a.rows - r0) : (uint64_t)OPEN_ROWS4);
    for (int r = ty; r < nr; r += 4 * TY) {
        uint4 v[4];
        uint64_t q[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool in = r + j * TY < nr;
            q[j] = in ? r0 + r + j * TY : r0 + r;
            v[j] = (active && in) ? ld_stream(a.mat + q[j] * a.ld + col) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NPTS; k++) {
            Ext w[4];
#pragma unroll
            for (int j = 0; j < 4; j++) w[j] = ld_ext(a.xw + 4 * ((uint64_t)k * a.xw_stride + q[j]));
#pragma unroll
            for (int i = 0; i < 4; i++) {
                dacc2(acc[k][i][0], v[0].x, w[0].c[i], v[1].x, w[1].c[i]); dacc2(acc[k][i][0], v[2].x, w[2].c[i], v[3].x, w[3].c[i]);
                dacc2(acc[k][i][1], v[0].y, w[0].c[i], v[1].y, w[1].c[i]); dacc2(acc[k][i][1], v[2].y, w[2].c[i], v[3].y, w[3].c[i]);
                dacc2(acc[k][i][2], v[0].z, w[0].c[i], v[1].z, w[1].c[i]); dacc2(acc[k][i][2], v[2].z, w[2].c[i], v[3].z, w[3].c[i]);
                dacc2(acc[k][i][3], v[0].w, w[0].c[i], v[1].w, w[1].c[i]); dacc2(acc[k][i][3], v[2].w, w[2].c[i], v[3].w, w[3].c[i]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NPTS; k++) {
        uint32_t mine[16];                          // [column][coefficient]
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) mine[4 * c + i] = dacc_finish(acc[k][i][c]);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; e++) red[threadIdx.x * 16 + e] = mine[e];
        __syncthreads();
        if (ty == 0 && active) {
            for (int y = 1; y < TY; y++)
#pragma unroll
                for (int e = 0; e < 16; e++) mine[e] = dadd(mine[e], red[(y * TX + tx) * 16 + e]);
            uint4* dst = reinterpret_cast<uint4*>(a.partial + 4 * (((uint64_t)blockIdx.x * NPTS + k) * a.width + col));
#pragma unroll
            for (int c = 0; c < 4; c++) dst[c] = make_uint4(mine[4 * c], mine[4 * c + 1], mine[4 * c + 2], mine[4 * c + 3]);
        }
    }
}
template <int NPTS>
__global__ void __launch_bounds__(256) open_partial4_kernel(OpenArgs a) { open_partial4_kernel_body<NPTS>(a); }
struct open_partial4_kernel_bargs { OpenArgs a; static open_partial4_kernel_bargs make(OpenArgs a) { return open_partial4_kernel_bargs{a}; } };
template <int NPTS>
__global__ void __launch_bounds__(256) open_partial4_kernel_batch(const open_partial4_kernel_bargs* __restrict__ zk_arr) { const open_partial4_kernel_bargs& zk_b = zk_arr[blockIdx.z]; open_partial4_kernel_body<NPTS>(zk_b.a); }

// out[pt][col] = -scale_pt * sum_chunk partial[chunk][pt][col]; one wave per output, the chunks spread over its lanes
__device__ __forceinline__ void open_final_kernel_body(const uint32_t* partial, uint32_t nchunks, int npts, uint32_t width, const Ext& scale0, const Ext& scale1, uint32_t* out) {
    const uint32_t idx = blockIdx.x;
    const uint32_t k = idx / width, col = idx % width;
    Ext sum = ext_zero();
    for (uint32_t c = threadIdx.x; c < nchunks; c += 64) sum = ext_add(sum, ld_ext(partial + 4 * (((uint64_t)c * npts + k) * width + col)));
    sum = group_sum(sum, 64);
    if (threadIdx.x == 0) st_ext(out + 4 * (uint64_t)idx, ext_neg(ext_mul_dev(sum, k ? scale1 : scale0)));
}
__global__ void __launch_bounds__(64) open_final_kernel(const uint32_t* partial, uint32_t nchunks, int npts, uint32_t width, Ext scale0, Ext scale1, uint32_t* out) { open_final_kernel_body(partial, nchunks, npts, width, scale0, scale1, out); }
struct open_final_kernel_bargs { const uint32_t* partial; uint32_t nchunks; int npts; uint32_t width; Ext scale0; Ext scale1; uint32_t* out; static open_final_kernel_bargs make(const uint32_t* partial, uint32_t nchunks, int npts, uint32_t width, Ext scale0, Ext scale1, uint32_t* out) { return open_final_kernel_bargs{partial, nchunks, npts, width, scale0, scale1, out}; } };
__global__ void __launch_bounds__(64) open_final_kernel_batch(const open_final_kernel_bargs* __restrict__ zk_arr) { const open_final_kernel_bargs& zk_b = zk_arr[blockIdx.z]; open_final_kernel_body(zk_b.partial, zk_b.nchunks, zk_b.npts, zk_b.width, zk_b.scale0, zk_b.scale1, zk_b.out); }

bool open_uses_quads(uint32_t width, uint64_t ld, const uint32_t* mat) {
    return width >= 64 && width % 4 == 0 && ld % 4 == 0 && ((uintptr_t)mat & 15u) == 0;
}
size_t open_chunks(uint64_t rows, uint32_t width, uint64_t ld, const uint32_t* mat) {
    const uint64_t per = open_uses_quads(width, ld, mat) ? OPEN_ROWS4 : OPEN_ROWS;
    return (size_t)((rows + per - 1) / per);
}
hipError_t launch_open(const OpenArgs& a0, int npts, const Ext& scale0, const Ext& scale1, uint32_t* out, hipStream_t s) {
    OpenArgs a = a0;
    const bool quads = open_uses_quads(a.width, a.ld, a.mat);
    const uint32_t nchunks = (uint32_t)open_chunks(a.rows, a.width, a.ld, a.mat);
    if (quads) {
        a.tx = 1;
        while (a.tx < (int)(a.width / 4) && a.tx < 64) a.tx <<= 1;      // lanes per row = pow2ceil(width / 4), at most 64
        dim3 grid(nchunks, (a.width / 4 + a.tx - 1) / a.tx);
        if (npts == 1) ZK_LAUNCH(open_partial4_kernel<1>, open_partial4_kernel_batch<1>, open_partial4_kernel_bargs, grid, dim3(256), 0, s, a);
        else ZK_LAUNCH(open_partial4_kernel<2>, open_partial4_kernel_batch<2>, open_partial4_kernel_bargs, grid, dim3(256), 0, s, a);
    } else {
        dim3 grid(nchunks, (a.width + a.tx - 1) / a.tx);
        if (npts == 1) ZK_LAUNCH(open_partial_kernel<1>, open_partial_kernel_batch<1>, open_partial_kernel_bargs, grid, dim3(256), 0, s, a);
        else ZK_LAUNCH(open_partial_kernel<2>, open_partial_kernel_batch<2>, open_partial_kernel_bargs, grid, dim3(256), 0, s, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint32_t n = (uint32_t)npts * a.width;
    ZK_LAUNCH(open_final_kernel, open_final_kernel_batch, open_final_kernel_bargs, dim3(n), dim3(64), 0, s, a.partial, nchunks, npts, a.width, scale0, scale1, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ reduced openings (FRI input)
// ro[p] = d1 (A_T - y_loc) + off_next d2 (A_T - y_next) + off_q d1 (A_Q - y_q),
// A_T = sum_j alpha^j T[p][j], A_Q = sum_{j<8} alpha^j Q[p][j].
// Two launches: (1) A_T per row, a row spread over L lanes, products summed in pairs in 64
// bits before one Montgomery reduction; (2) one lane per row for the extension-field tail,
// so that the ~100 multiplications of the tail keep all 64 lanes busy.
// A row is spread over L <= 16 lanes of one DPP row (a wave covers 64 / L rows); each lane keeps four
// 64-bit running sums (dacc2) and reduces them once per row, and the lane partials are summed with
// DPP row rotations -- no LDS, no ds_bpermute.
__device__ __forceinline__ void rowdot_kernel_body(const uint32_t* __restrict__ mat, uint64_t ld, uint32_t width, uint64_t rows, int L, const uint32_t* __restrict__ alpha_pow, uint32_t* __restrict__ out_at) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p = gid / L;
    const int lane = (int)(gid % L);
    if (p >= rows) return;              // rows * L is a multiple of 64: whole waves exit together
    const uint32_t* row = mat + p * ld;
    uint64_t acc[4] = {0, 0, 0, 0};
    const uint32_t nq = width / 4;
    for (uint32_t q = lane; q < nq; q += L) {
        const uint4 v = ld_stream(row + 4 * q);
        const uint4* ap = reinterpret_cast<const uint4*>(alpha_pow + 16 * q);
        const uint4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
        dacc2(acc[0], a0.x, v.x, a1.x, v.y); dacc2(acc[0], a2.x, v.z, a3.x, v.w);
        dacc2(acc[1], a0.y, v.x, a1.y, v.y); dacc2(acc[1], a2.y, v.z, a3.y, v.w);
        dacc2(acc[2], a0.z, v.x, a1.z, v.y); dacc2(acc[2], a2.z, v.z, a3.z, v.w);
        dacc2(acc[3], a0.w, v.x, a1.w, v.y); dacc2(acc[3], a2.w, v.z, a3.w, v.w);
    }
    Ext r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.c[i] = row_group_sum(dacc_finish(acc[i]), L);
    if (lane == 0) st_ext(out_at + 4 * p, r);
}
__global__ void __launch_bounds__(256) rowdot_kernel(const uint32_t* __restrict__ mat, uint64_t ld, uint32_t width, uint64_t rows, int L, const uint32_t* __restrict__ alpha_pow, uint32_t* __restrict__ out_at) { rowdot_kernel_body(mat, ld, width, rows, L, alpha_pow, out_at); }
struct rowdot_kernel_bargs { const uint32_t* mat; uint64_t ld; uint32_t width; uint64_t rows; int L; const uint32_t* alpha_pow; uint32_t* out_at; static rowdot_kernel_bargs make(const uint32_t* mat, uint64_t ld, uint32_t width, uint64_t rows, int L, const uint32_t* alpha_pow, uint32_t* out_at) { return rowdot_kernel_bargs{mat, ld, width, rows, L, alpha_pow, out_at}; } };
__global__ void __launch_bounds__(256) rowdot_kernel_batch(const rowdot_kernel_bargs* __restrict__ zk_arr) { const rowdot_kernel_bargs& zk_b = zk_arr[blockIdx.z]; rowdot_kernel_body(zk_b.mat, zk_b.ld, zk_b.width, zk_b.rows, zk_b.L, zk_b.alpha_pow, zk_b.out_at); }

// The same dot product with the powers of alpha kept in REGISTERS: a lane owns the column quads q = lane + L k, k < NK, of EVERY row its
// wavefront visits (64 / L rows per trip, ROWDOT_TRIPS trips), so a row costs its own 16-byte loads only.  In the form above every
// 16 bytes of a row came with 64 bytes of powers from L1 -- five load instructions per payload load: the kernel was bound by the
// texture path at 4.0 TB/s of HBM traffic.  Used for widths up to 1024 / (16 / NK_MAX) = 256 columns per 16 lanes (NK <= 4).
constexpr int ROWDOT_TRIPS = 16;
template <int NK>
__device__ __forceinline__ void rowdot_regs_kernel_body(const uint32_t* __restrict__ mat, uint64_t ld, uint32_t width, uint64_t rows, int L, const uint32_t* __restrict__ alpha_pow, uint32_t* __restrict__ out_at) {
    const int lane = (int)(threadIdx.x % L);
    const uint32_t nq = width / 4;
    const uint64_t rows_per_trip = 256 / L;                       // rows a workgroup covers per trip
    const uint64_t r0 = (uint64_t)blockIdx.x * rows_per_trip * ROWDOT_TRIPS + threadIdx.x / L;
    uint4 a0[NK], a1[NK], a2[NK], a3[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const uint32_t q = lane + (uint32_t)L * k;
        const uint4* ap = reinterpret_cast<const uint4*>(alpha_pow + 16 * (q < nq ? q : 0));
        a0[k] = ap[0]; a1[k] = ap[1]; a2[k] = ap[2]; a3[k] = ap[3];
    }
    for (int t = 0; t < ROWDOT_TRIPS; t++) {
        const uint64_t p = r0 + (uint64_t)t * rows_per_trip;
        if (p >= rows) return;                                    // rows_per_trip divides the row count of every caller: whole waves leave together
        const uint32_t* row = mat + p * ld;
        uint4 v[NK];
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const uint32_t q = lane + (uint32_t)L * k;
            v[k] = q < nq ? ld_stream(row + 4 * q) : make_uint4(0, 0, 0, 0);
        }
        uint64_t acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < NK; k++) {
            dacc2(acc[0], a0[k].x, v[k].x, a1[k].x, v[k].y); dacc2(acc[0], a2[k].x, v[k].z, a3[k].x, v[k].w);
            dacc2(acc[1], a0[k].y, v[k].x, a1[k].y, v[k].y); dacc2(acc[1], a2[k].y, v[k].z, a3[k].y, v[k].w);
            dacc2(acc[2], a0[k].z, v[k].x, a1[k].z, v[k].y); dacc2(acc[2], a2[k].z, v[k].z, a3[k].z, v[k].w);
            dacc2(acc[3], a0[k].w, v[k].x, a1[k].w, v[k].y); dacc2(acc[3], a2[k].w, v[k].z, a3[k].w, v[k].w);
        }
        Ext r;
#pragma unroll
        for (int i = 0; i < 4; i++) r.c[i] = row_group_sum(dacc_finish(acc[i]), L);
        if (lane == 0) st_ext(out_at + 4 * p, r);
    }
}
template <int NK>
__global__ void __launch_bounds__(256) rowdot_regs_kernel(const uint32_t* __restrict__ mat, uint64_t ld, uint32_t width, uint64_t rows, int L, const uint32_t* __restrict__ alpha_pow, uint32_t* __restrict__ out_at) { rowdot_regs_kernel_body<NK>(mat, ld, width, rows, L, alpha_pow, out_at); }
struct rowdot_regs_kernel_bargs { const uint32_t* mat; uint64_t ld; uint32_t width; uint64_t rows; int L; const uint32_t* alpha_pow; uint32_t* out_at; static rowdot_regs_kernel_bargs make(const uint32_t* mat, uint64_t ld, uint32_t width, uint64_t rows, int L, const uint32_t* alpha_pow, uint32_t* out_at) { return rowdot_regs_kernel_bargs{mat, ld, width, rows, L, alpha_pow, out_at}; } };
template <int NK>
__global__ void __launch_bounds__(256) rowdot_regs_kernel_batch(const rowdot_regs_kernel_bargs* __restrict__ zk_arr) { const rowdot_regs_kernel_bargs& zk_b = zk_arr[blockIdx.z]; rowdot_regs_kernel_body<NK>(zk_b.mat, zk_b.ld, zk_b.width, zk_b.rows, zk_b.L, zk_b.alpha_pow, zk_b.out_at); }

static hipError_t launch_rowdot(const uint32_t* mat, uint64_t ld, uint32_t width, uint64_t rows, const uint32_t* alpha_pow, uint32_t* out_at, hipStream_t s);
__device__ __forceinline__ void reduced_combine_kernel_body(const ReducedArgs& a, const uint32_t* __restrict__ at_in, const uint32_t* __restrict__ ap_in) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.rows) return;
    const Ext at = ld_ext(at_in + 4 * p);
    const uint4* qrow = reinterpret_cast<const uint4*>(a.qlde + p * a.q_ld);
    Ext aq = ext_zero();
    for (uint32_t h = 0; h < a.q_width / 8; h++) {              // 8 columns (two quotient chunks) per trip: one trip, or two with four chunks
        const uint4 q0 = qrow[2 * h], q1 = qrow[2 * h + 1];
        const uint32_t qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int j = 0; j < 8; j++) aq = ext_add(aq, ext_mul_base_dev(ld_ext(a.alpha_pow + 4 * (8 * h + j)), qv[j]));
    }
    const Ext d1 = ld_ext(a.dinv + 4 * p), d2 = ld_ext(a.dinv + 4 * (a.rows + p));
    Ext r = ext_mul_dev(a.off_loc, ext_mul_dev(ext_sub(at, a.y_loc), d1));
    r = ext_add(r, ext_mul_dev(a.off_next, ext_mul_dev(ext_sub(at, a.y_next), d2)));
    if (a.p_width) {
        const Ext ap = ld_ext(ap_in + 4 * p);
        r = ext_add(r, ext_mul_dev(a.off_pl, ext_mul_dev(ext_sub(ap, a.y_pl), d1)));
        r = ext_add(r, ext_mul_dev(a.off_pn, ext_mul_dev(ext_sub(ap, a.y_pn), d2)));
    }
    r = ext_add(r, ext_mul_dev(a.off_q, ext_mul_dev(ext_sub(aq, a.y_q), d1)));
    if (a.accumulate) r = ext_add(r, ld_ext(a.out + 4 * p));
    st_ext(a.out + 4 * p, r);
}
__global__ void __launch_bounds__(256) reduced_combine_kernel(ReducedArgs a, const uint32_t* __restrict__ at_in, const uint32_t* __restrict__ ap_in) { reduced_combine_kernel_body(a, at_in, ap_in); }
struct reduced_combine_kernel_bargs { ReducedArgs a; const uint32_t* at_in; const uint32_t* ap_in; static reduced_combine_kernel_bargs make(ReducedArgs a, const uint32_t* at_in, const uint32_t* ap_in) { return reduced_combine_kernel_bargs{a, at_in, ap_in}; } };
__global__ void __launch_bounds__(256) reduced_combine_kernel_batch(const reduced_combine_kernel_bargs* __restrict__ zk_arr) { const reduced_combine_kernel_bargs& zk_b = zk_arr[blockIdx.z]; reduced_combine_kernel_body(zk_b.a, zk_b.at_in, zk_b.ap_in); }

static int lanes_for(uint32_t width) { int g = (int)(width / 4), l = 1; while (l < g && l < 16) l <<= 1; return l; }
static hipError_t launch_rowdot(const uint32_t* mat, uint64_t ld, uint32_t width, uint64_t rows, const uint32_t* alpha_pow, uint32_t* out_at, hipStream_t s) {
    const int L = lanes_for(width);
    const uint32_t nq = width / 4;
    const int nk = (int)((nq + L - 1) / L);
    const uint64_t rows_per_wg = (uint64_t)(256 / L) * ROWDOT_TRIPS;
    if (nk <= 4 && rows % (256 / L) == 0 && rows >= rows_per_wg) {
        const dim3 grid((unsigned)((rows + rows_per_wg - 1) / rows_per_wg)), block(256);
        switch (nk) {
            case 1: ZK_LAUNCH(rowdot_regs_kernel<1>, rowdot_regs_kernel_batch<1>, rowdot_regs_kernel_bargs, grid, block, 0, s, mat, ld, width, rows, L, alpha_pow, out_at); break;
            case 2: ZK_LAUNCH(rowdot_regs_kernel<2>, rowdot_regs_kernel_batch<2>, rowdot_regs_kernel_bargs, grid, block, 0, s, mat, ld, width, rows, L, alpha_pow, out_at); break;
            case 3: ZK_LAUNCH(rowdot_regs_kernel<3>, rowdot_regs_kernel_batch<3>, rowdot_regs_kernel_bargs, grid, block, 0, s, mat, ld, width, rows, L, alpha_pow, out_at); break;
            default: ZK_LAUNCH(rowdot_regs_kernel<4>, rowdot_regs_kernel_batch<4>, rowdot_regs_kernel_bargs, grid, block, 0, s, mat, ld, width, rows, L, alpha_pow, out_at); break;
        }
        return hipGetLastError();
    }
    const uint64_t threads = rows * L;
    ZK_LAUNCH(rowdot_kernel, rowdot_kernel_batch, rowdot_kernel_bargs, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, mat, ld, width, rows, L, alpha_pow, out_at);
    return hipGetLastError();
}
hipError_t launch_reduced_opening(const ReducedArgs& a, uint32_t* scratch_at, hipStream_t s) {
    hipError_t e = launch_rowdot(a.tlde, a.t_ld, a.width, a.rows, a.alpha_pow, scratch_at, s);
    if (e != hipSuccess) return e;
    uint32_t* scratch_ap = scratch_at + 4 * a.rows;
    if (a.p_width) {
        e = launch_rowdot(a.plde, a.p_ld, a.p_width, a.rows, a.alpha_pow, scratch_ap, s);
        if (e != hipSuccess) return e;
    }
    ZK_LAUNCH(reduced_combine_kernel, reduced_combine_kernel_batch, reduced_combine_kernel_bargs, dim3((unsigned)((a.rows + 255) / 256)), dim3(256), 0, s, a, scratch_at, scratch_ap);
    return hipGetLastError();
}

// ------------------------------------------------------------------ LogUp permutation trace
// phi_q[i] = 1/(gamma + a_s + beta b_s) - 1/(gamma + a_r + beta b_r), S = running sum of the row sums.
// (sp1-stark generate_permutation_trace, reference Cargo.lock:6172: per-row extension inverses,
// then a prefix sum.)  Three launches: rows + block-local scan, scan of the block totals, fix-up.
constexpr int PERM_BLOCK = 256;
__device__ __forceinline__ void perm_rows_kernel_body(const PermArgs& a, uint32_t* __restrict__ block_tot) {
    __shared__ uint32_t sh[PERM_BLOCK * 4];
    const uint64_t i = (uint64_t)blockIdx.x * PERM_BLOCK + threadIdx.x;
    Ext sum = ext_zero();
    if (i < a.rows) {
        const uint32_t* row = a.trace + i * a.ld;
        uint32_t* prow = a.out + i * a.out_ld;
        for (uint32_t q = 0; q < a.pairs; q++) {
            const uint4 vs = *reinterpret_cast<const uint4*>(row + 8 * q);
            const uint4 vr = *reinterpret_cast<const uint4*>(row + 8 * q + 4);
            const Ext ds = ext_add(ext_add_base(a.gamma, vs.x), ext_mul_base_dev(a.beta, vs.y));
            const Ext dr = ext_add(ext_add_base(a.gamma, vr.x), ext_mul_base_dev(a.beta, vr.y));
            // 1/ds - 1/dr = (dr - ds) / (ds dr): one extension inversion per pair instead of two (an inversion costs about six
            // extension products).  A zero denominator (1/0 = 0 by the protocol convention) takes the direct formula.
            const Ext d = ext_mul_dev(ds, dr);
            const Ext phi = ext_eq(d, ext_zero()) ? ext_sub(ext_inv_dev(ds), ext_inv_dev(dr)) : ext_mul_dev(ext_sub(dr, ds), ext_inv_dev(d));
            st_ext(prow + 4 * q, phi);
            sum = ext_add(sum, phi);
        }
    }
    // inclusive scan of the row sums inside the block (Hillis-Steele, exact modular adds)
    for (int k = 0; k < 4; k++) sh[threadIdx.x * 4 + k] = sum.c[k];
    __syncthreads();
    for (int off = 1; off < PERM_BLOCK; off <<= 1) {
        Ext o = ext_zero();
        if ((int)threadIdx.x >= off) o = Ext{{sh[(threadIdx.x - off) * 4], sh[(threadIdx.x - off) * 4 + 1], sh[(threadIdx.x - off) * 4 + 2], sh[(threadIdx.x - off) * 4 + 3]}};
        __syncthreads();
        sum = ext_add(sum, o);
        for (int k = 0; k < 4; k++) sh[threadIdx.x * 4 + k] = sum.c[k];
        __syncthreads();
    }
    if (i < a.rows) st_ext(a.out + i * a.out_ld + 4 * a.pairs, sum);
    if (threadIdx.x == PERM_BLOCK - 1) st_ext(block_tot + 4 * (uint64_t)blockIdx.x, sum);
}
__global__ void __launch_bounds__(PERM_BLOCK) perm_rows_kernel(PermArgs a, uint32_t* __restrict__ block_tot) { perm_rows_kernel_body(a, block_tot); }
struct perm_rows_kernel_bargs { PermArgs a; uint32_t* block_tot; static perm_rows_kernel_bargs make(PermArgs a, uint32_t* block_tot) { return perm_rows_kernel_bargs{a, block_tot}; } };
__global__ void __launch_bounds__(PERM_BLOCK) perm_rows_kernel_batch(const perm_rows_kernel_bargs* __restrict__ zk_arr) { const perm_rows_kernel_bargs& zk_b = zk_arr[blockIdx.z]; perm_rows_kernel_body(zk_b.a, zk_b.block_tot); }

// exclusive scan of the block totals, one workgroup (nblocks <= 65536)
__device__ __forceinline__ void perm_scan_blocks_kernel_body(uint32_t* block_tot, uint32_t nblocks) {
    __shared__ uint32_t sh[1024 * 4];
    const uint32_t per = (nblocks + 1023) / 1024;
    const uint32_t b0 = threadIdx.x * per;
    Ext sum = ext_zero();
    for (uint32_t k = 0; k < per && b0 + k < nblocks; k++) sum = ext_add(sum, ld_ext(block_tot + 4 * (uint64_t)(b0 + k)));
    for (int k = 0; k < 4; k++) sh[threadIdx.x * 4 + k] = sum.c[k];
    __syncthreads();
    Ext incl = sum;
    for (int off = 1; off < 1024; off <<= 1) {
        Ext o = ext_zero();
        if ((int)threadIdx.x >= off) o = Ext{{sh[(threadIdx.x - off) * 4], sh[(threadIdx.x - off) * 4 + 1], sh[(threadIdx.x - off) * 4 + 2], sh[(threadIdx.x - off) * 4 + 3]}};
        __syncthreads();
        incl = ext_add(incl, o);
        for (int k = 0; k < 4; k++) sh[threadIdx.x * 4 + k] = incl.c[k];
        __syncthreads();
    }
    Ext run = ext_sub(incl, sum);                  // exclusive prefix of this thread's chunk
    for (uint32_t k = 0; k < per && b0 + k < nblocks; k++) {
        const Ext t = ld_ext(block_tot + 4 * (uint64_t)(b0 + k));
        st_ext(block_tot + 4 * (uint64_t)(b0 + k), run);
        run = ext_add(run, t);
    }
}
__global__ void __launch_bounds__(1024) perm_scan_blocks_kernel(uint32_t* block_tot, uint32_t nblocks) { perm_scan_blocks_kernel_body(block_tot, nblocks); }
struct perm_scan_blocks_kernel_bargs { uint32_t* block_tot; uint32_t nblocks; static perm_scan_blocks_kernel_bargs make(uint32_t* block_tot, uint32_t nblocks) { return perm_scan_blocks_kernel_bargs{block_tot, nblocks}; } };
__global__ void __launch_bounds__(1024) perm_scan_blocks_kernel_batch(const perm_scan_blocks_kernel_bargs* __restrict__ zk_arr) { const perm_scan_blocks_kernel_bargs& zk_b = zk_arr[blockIdx.z]; perm_scan_blocks_kernel_body(zk_b.block_tot, zk_b.nblocks); }

__device__ __forceinline__ void perm_fixup_kernel_body(const PermArgs& a, const uint32_t* __restrict__ block_off) {
    const uint64_t i = (uint64_t)blockIdx.x * PERM_BLOCK + threadIdx.x;
    if (i >= a.rows) return;
    uint32_t* sp = a.out + i * a.out_ld + 4 * a.pairs;
    st_ext(sp, ext_add(ld_ext(sp), ld_ext(block_off + 4 * (uint64_t)blockIdx.x)));
}
__global__ void __launch_bounds__(PERM_BLOCK) perm_fixup_kernel(PermArgs a, const uint32_t* __restrict__ block_off) { perm_fixup_kernel_body(a, block_off); }
struct perm_fixup_kernel_bargs { PermArgs a; const uint32_t* block_off; static perm_fixup_kernel_bargs make(PermArgs a, const uint32_t* block_off) { return perm_fixup_kernel_bargs{a, block_off}; } };
__global__ void __launch_bounds__(PERM_BLOCK) perm_fixup_kernel_batch(const perm_fixup_kernel_bargs* __restrict__ zk_arr) { const perm_fixup_kernel_bargs& zk_b = zk_arr[blockIdx.z]; perm_fixup_kernel_body(zk_b.a, zk_b.block_off); }

hipError_t launch_perm_trace(const PermArgs& a, uint32_t* block_scratch, hipStream_t s) {
    const uint32_t nblocks = (uint32_t)((a.rows + PERM_BLOCK - 1) / PERM_BLOCK);
    ZK_LAUNCH(perm_rows_kernel, perm_rows_kernel_batch, perm_rows_kernel_bargs, dim3(nblocks), dim3(PERM_BLOCK), 0, s, a, block_scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    ZK_LAUNCH(perm_scan_blocks_kernel, perm_scan_blocks_kernel_batch, perm_scan_blocks_kernel_bargs, dim3(1), dim3(1024), 0, s, block_scratch, nblocks);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    ZK_LAUNCH(perm_fixup_kernel, perm_fixup_kernel_batch, perm_fixup_kernel_bargs, dim3(nblocks), dim3(PERM_BLOCK), 0, s, a, block_scratch);
    return hipGetLastError();
}

// ---- lookups as data (kernels.h, LookupArgs): the same three launches with the interactions read from a table
// (cm != nullptr: `row` is the lane's STAGED row in LDS and cm[column] the word's place in it -- lookup_stage_rows below)
ZK_D Ext lookup_fingerprint(const LookupArgs& lk, const uint32_t* rec, const uint32_t* row, const uint8_t* cm = nullptr) {
    Ext d = ext_add_base(lk.gamma, rec[2]);
    for (uint32_t v = 0; v < rec[3]; v++) d = ext_add(d, ext_mul_base_dev(lk.bpow[v + 1], row[cm ? cm[rec[4 + v]] : rec[4 + v]]));
    return d;
}
ZK_D uint32_t lookup_mult(const uint32_t* rec, const uint32_t* row, const uint8_t* cm = nullptr) {       // signed multiplicity, base field
    const uint32_t m = rec[1] == 0xFFFFFFFFu ? MONTY_R1 : row[cm ? cm[rec[1]] : rec[1]];
    return rec[0] ? (m ? P - m : 0u) : m;
}
// The rows [row0, row0 + 256) of a row-major matrix, the columns the interactions read, into an LDS tile (kernels.h LookupArgs: cmap, chunk_mask): a workgroup of 256
// lanes loads every touched 16-column chunk with 16-byte loads -- four lanes per row and chunk, 64 rows per instruction -- instead of every lane gathering its own
// row's words one scattered request at a time.  dyn: LOOKUP_MAX_COLS bytes of map, then 256 x pitch words; returns the calling lane's staged row.
ZK_D const uint32_t* lookup_stage_rows(const LookupArgs& lk, const uint32_t* mat, uint64_t ld, uint64_t row0, uint64_t rows, uint32_t* dyn, const uint8_t** cm_out,
                                       const uint32_t* pre = nullptr, uint64_t pre_ld = 0, uint32_t pre_w = 0) {
    uint8_t* cm = reinterpret_cast<uint8_t*>(dyn);
    uint32_t* tile = dyn + LOOKUP_MAX_COLS / 4;
    const uint32_t pitch = lk.n_used | 1u, tid = threadIdx.x;
    for (uint32_t i = tid; i < LOOKUP_MAX_COLS / 4; i += blockDim.x) dyn[i] = reinterpret_cast<const uint32_t*>(lk.cmap)[i];
    __syncthreads();
    const uint32_t q = tid & 3u;
    for (uint32_t m = lk.chunk_mask; m; m &= m - 1u) {
        const uint32_t col = 16u * (uint32_t)(__ffs((int)m) - 1) + 4u * q;
        const uint8_t k0 = cm[col], k1 = cm[col + 1], k2 = cm[col + 2], k3 = cm[col + 3];
        if ((k0 & k1 & k2 & k3) == 0xFF) continue;                           // (none of this lane's four words is read; the other lanes of the row go on)
        for (uint32_t r = tid >> 2; r < 256u; r += blockDim.x >> 2) {
            if (row0 + r >= rows) break;
            // (two sources: a quad of columns lies on one side of pre_w, a multiple of 4)
            const bool from_pre = pre != nullptr && col < pre_w;
            const uint32_t* src = from_pre ? pre + (row0 + r) * pre_ld + col : mat + (row0 + r) * ld + (col - (pre ? pre_w : 0u));
            uint32_t* dst = tile + r * pitch;
            if (from_pre || col - (pre ? pre_w : 0u) + 3 < ld) {
                const uint4 v = *reinterpret_cast<const uint4*>(src);
                if (k0 != 0xFF) dst[k0] = v.x;
                if (k1 != 0xFF) dst[k1] = v.y;
                if (k2 != 0xFF) dst[k2] = v.z;
                if (k3 != 0xFF) dst[k3] = v.w;
            } else {
                const uint32_t mc = col - (pre ? pre_w : 0u);
                if (k0 != 0xFF && mc < ld) dst[k0] = src[0];
                if (k1 != 0xFF && mc + 1 < ld) dst[k1] = src[1];
                if (k2 != 0xFF && mc + 2 < ld) dst[k2] = src[2];
            }
        }
    }
    __syncthreads();
    *cm_out = cm;
    return tile + tid * pitch;
}
ZK_D bool lookup_stageable(const LookupArgs& lk, const uint32_t* mat, uint64_t ld) { return lk.cmap != nullptr && (ld & 3u) == 0 && ((uintptr_t)mat & 15u) == 0; }
static size_t lookup_stage_bytes(const LookupArgs& lk, const uint32_t* mat, uint64_t ld) {
    return lk.cmap != nullptr && (ld & 3u) == 0 && ((uintptr_t)mat & 15u) == 0 ? LOOKUP_MAX_COLS + 256 * (size_t)(lk.n_used | 1u) * 4 : 0;
}
__device__ __forceinline__ void perm_rows_machine_kernel_body(const MachinePermArgs& a, uint32_t* __restrict__ block_tot) {
    __shared__ uint32_t sh[PERM_BLOCK * 4];
    const uint64_t i = (uint64_t)blockIdx.x * PERM_BLOCK + threadIdx.x;
    Ext sum = ext_zero();
    extern __shared__ uint32_t lookup_dyn[];
    const uint8_t* cm = nullptr;
    const uint32_t* row = a.trace + i * a.ld;
    if (lookup_stageable(a.lk, a.trace, a.ld)) row = lookup_stage_rows(a.lk, a.trace, a.ld, (uint64_t)blockIdx.x * PERM_BLOCK, a.rows, lookup_dyn, &cm, a.pre, a.pre_ld, a.pre_w);
    if (i < a.rows) {
        uint32_t* prow = a.out + i * a.out_ld;
        for (uint32_t j = 0; j < a.lk.cols; j++) {
            const uint32_t* ra = a.lk.table + (size_t)(2 * j) * LOOKUP_REC_WORDS;
            const Ext da = lookup_fingerprint(a.lk, ra, row, cm);
            const uint32_t ma = lookup_mult(ra, row, cm);
            Ext phi;
            if (2 * j + 1 < a.lk.ni) {
                const uint32_t* rb = ra + LOOKUP_REC_WORDS;
                const Ext db = lookup_fingerprint(a.lk, rb, row, cm);
                const uint32_t mb = lookup_mult(rb, row, cm);
                // m_a / d_a + m_b / d_b = (m_a d_b + m_b d_a) / (d_a d_b): one inversion per column; 1/0 = 0 takes the direct formula
                const Ext d = ext_mul_dev(da, db);
                phi = ext_eq(d, ext_zero()) ? ext_add(ext_mul_base_dev(ext_inv_dev(da), ma), ext_mul_base_dev(ext_inv_dev(db), mb))
                                            : ext_mul_dev(ext_add(ext_mul_base_dev(db, ma), ext_mul_base_dev(da, mb)), ext_inv_dev(d));
            } else phi = ext_mul_base_dev(ext_inv_dev(da), ma);
            st_ext(prow + 4 * j, phi);
            sum = ext_add(sum, phi);
        }
    }
    for (int k = 0; k < 4; k++) sh[threadIdx.x * 4 + k] = sum.c[k];
    __syncthreads();
    for (int off = 1; off < PERM_BLOCK; off <<= 1) {
        Ext o = ext_zero();
        if ((int)threadIdx.x >= off) o = Ext{{sh[(threadIdx.x - off) * 4], sh[(threadIdx.x - off) * 4 + 1], sh[(threadIdx.x - off) * 4 + 2], sh[(threadIdx.x - off) * 4 + 3]}};
        __syncthreads();
        sum = ext_add(sum, o);
        for (int k = 0; k < 4; k++) sh[threadIdx.x * 4 + k] = sum.c[k];
        __syncthreads();
    }
    if (i < a.rows) st_ext(a.out + i * a.out_ld + 4 * a.lk.cols, sum);
    if (threadIdx.x == PERM_BLOCK - 1) st_ext(block_tot + 4 * (uint64_t)blockIdx.x, sum);
}
__global__ void __launch_bounds__(PERM_BLOCK) perm_rows_machine_kernel(MachinePermArgs a, uint32_t* __restrict__ block_tot) { perm_rows_machine_kernel_body(a, block_tot); }
struct perm_rows_machine_kernel_bargs { MachinePermArgs a; uint32_t* block_tot; static perm_rows_machine_kernel_bargs make(MachinePermArgs a, uint32_t* block_tot) { return perm_rows_machine_kernel_bargs{a, block_tot}; } };
__global__ void __launch_bounds__(PERM_BLOCK) perm_rows_machine_kernel_batch(const perm_rows_machine_kernel_bargs* __restrict__ zk_arr) { const perm_rows_machine_kernel_bargs& zk_b = zk_arr[blockIdx.z]; perm_rows_machine_kernel_body(zk_b.a, zk_b.block_tot); }

bool lookup_perm_two_sources_ok(const MachinePermArgs& a) {
    return lookup_stage_bytes(a.lk, a.trace, a.ld) != 0 && a.pre != nullptr && (a.pre_w & 3u) == 0 && (a.pre_ld & 3u) == 0 && ((uintptr_t)a.pre & 15u) == 0;
}
hipError_t launch_perm_trace_machine(const MachinePermArgs& a, uint32_t* block_scratch, hipStream_t s) {
    if (a.pre && !lookup_perm_two_sources_ok(a)) return hipErrorInvalidValue;         // (the caller puts the two side by side first when they cannot be read as they lie)
    const uint32_t nblocks = (uint32_t)((a.rows + PERM_BLOCK - 1) / PERM_BLOCK);
    static_assert(PERM_BLOCK == 256, "lookup_stage_rows stages 256 rows per workgroup");
    ZK_LAUNCH(perm_rows_machine_kernel, perm_rows_machine_kernel_batch, perm_rows_machine_kernel_bargs, dim3(nblocks), dim3(PERM_BLOCK), lookup_stage_bytes(a.lk, a.trace, a.ld), s, a, block_scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    ZK_LAUNCH(perm_scan_blocks_kernel, perm_scan_blocks_kernel_batch, perm_scan_blocks_kernel_bargs, dim3(1), dim3(1024), 0, s, block_scratch, nblocks);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    PermArgs fix{};                                   // the fix-up only needs where the running-sum column lives
    fix.rows = a.rows; fix.pairs = a.lk.cols; fix.out = a.out; fix.out_ld = a.out_ld;
    ZK_LAUNCH(perm_fixup_kernel, perm_fixup_kernel_batch, perm_fixup_kernel_bargs, dim3(nblocks), dim3(PERM_BLOCK), 0, s, fix, block_scratch);
    return hipGetLastError();
}
__device__ __forceinline__ void lookup_addend_kernel_body(const MachineQuotArgs& a) {
    const int H = a.log_n + a.log_qd;
    const uint32_t m = 1u << H;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    extern __shared__ uint32_t lookup_dyn[];
    const uint8_t* cm = nullptr;
    const uint32_t* row = a.lde + (uint64_t)p * a.ld;
    if (lookup_stageable(a.lk, a.lde, a.ld)) row = lookup_stage_rows(a.lk, a.lde, a.ld, (uint64_t)blockIdx.x * 256u, m, lookup_dyn, &cm);
    if (p >= m) return;
    const uint32_t e = __brev(p) >> (32 - H);
    const uint32_t pn = __brev((e + (1u << a.log_qd)) & (m - 1)) >> (32 - H);
    const uint32_t* prow = a.perm + (uint64_t)p * a.perm_ld;
    const uint32_t* pnrow = a.perm + (uint64_t)pn * a.perm_ld;
    Ext r = ext_zero(), sphi = ext_zero(), sphin = ext_zero();
    for (uint32_t j = 0; j < a.lk.cols; j++) {
        const uint32_t* ra = a.lk.table + (size_t)(2 * j) * LOOKUP_REC_WORDS;
        const Ext da = lookup_fingerprint(a.lk, ra, row, cm);
        const uint32_t ma = lookup_mult(ra, row, cm);
        const Ext phi = ld_ext(prow + 4 * j);
        Ext c;
        if (2 * j + 1 < a.lk.ni) {
            const uint32_t* rb = ra + LOOKUP_REC_WORDS;
            const Ext db = lookup_fingerprint(a.lk, rb, row, cm);
            const uint32_t mb = lookup_mult(rb, row, cm);
            c = ext_sub(ext_mul_dev(ext_mul_dev(phi, da), db), ext_add(ext_mul_base_dev(db, ma), ext_mul_base_dev(da, mb)));
        } else c = ext_sub_base(ext_mul_dev(phi, da), ma);
        r = ext_add(r, ext_mul_dev(c, ld_ext(a.weights + 4 * j)));
        sphi = ext_add(sphi, phi);
        sphin = ext_add(sphin, ld_ext(pnrow + 4 * j));
    }
    const Ext S = ld_ext(prow + 4 * a.lk.cols), Sn = ld_ext(pnrow + 4 * a.lk.cols);
    const uint32_t sel_trans = dsub(a.xs[p], a.wn_inv);
    const Ext F1 = ext_mul_base_dev(ld_ext(a.weights + 4 * a.lk.cols), a.sel_first[p]);
    const Ext F2 = ext_mul_base_dev(ld_ext(a.weights + 4 * (a.lk.cols + 1)), sel_trans);
    const Ext F3 = ext_mul_base_dev(ld_ext(a.weights + 4 * (a.lk.cols + 2)), a.sel_last[p]);
    r = ext_add(r, ext_mul_dev(F1, ext_sub(S, sphi)));
    r = ext_add(r, ext_mul_dev(F2, ext_sub(ext_sub(Sn, S), sphin)));
    r = ext_add(r, ext_mul_dev(F3, ext_sub(S, a.cumsum)));
    st_ext(a.addend + 4 * (uint64_t)p, r);
}
__global__ void __launch_bounds__(256) lookup_addend_kernel(MachineQuotArgs a) { lookup_addend_kernel_body(a); }
struct lookup_addend_kernel_bargs { MachineQuotArgs a; static lookup_addend_kernel_bargs make(MachineQuotArgs a) { return lookup_addend_kernel_bargs{a}; } };
__global__ void __launch_bounds__(256) lookup_addend_kernel_batch(const lookup_addend_kernel_bargs* __restrict__ zk_arr) { const lookup_addend_kernel_bargs& zk_b = zk_arr[blockIdx.z]; lookup_addend_kernel_body(zk_b.a); }

hipError_t launch_lookup_addend(const MachineQuotArgs& a, hipStream_t s) {
    const uint64_t m = 1ull << (a.log_n + a.log_qd);
    ZK_LAUNCH(lookup_addend_kernel, lookup_addend_kernel_batch, lookup_addend_kernel_bargs, dim3((unsigned)((m + 255) / 256)), dim3(256), lookup_stage_bytes(a.lk, a.lde, a.ld), s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ FRI fold (arity 2)
// out[i] = (e0 + e1)/2 + beta (e0 - e1) / (2 x_i),  itw[i] = 1 / (2 x_i)
__device__ __forceinline__ void fri_fold_kernel_body(const uint32_t* in, uint32_t* out, const uint32_t* itw, uint64_t half, const Ext& beta) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const Ext e0 = ld_ext(in + 8 * i), e1 = ld_ext(in + 8 * i + 4);
    const Ext s = ext_mul_base_dev(ext_add(e0, e1), MONTY_INV2);
    const Ext d = ext_mul_base_dev(ext_sub(e0, e1), itw[i]);
    st_ext(out + 4 * i, ext_add(s, ext_mul_dev(beta, d)));
}
__global__ void __launch_bounds__(256) fri_fold_kernel(const uint32_t* in, uint32_t* out, const uint32_t* itw, uint64_t half, Ext beta) { fri_fold_kernel_body(in, out, itw, half, beta); }
struct fri_fold_kernel_bargs { const uint32_t* in; uint32_t* out; const uint32_t* itw; uint64_t half; Ext beta; static fri_fold_kernel_bargs make(const uint32_t* in, uint32_t* out, const uint32_t* itw, uint64_t half, Ext beta) { return fri_fold_kernel_bargs{in, out, itw, half, beta}; } };
__global__ void __launch_bounds__(256) fri_fold_kernel_batch(const fri_fold_kernel_bargs* __restrict__ zk_arr) { const fri_fold_kernel_bargs& zk_b = zk_arr[blockIdx.z]; fri_fold_kernel_body(zk_b.in, zk_b.out, zk_b.itw, zk_b.half, zk_b.beta); }

hipError_t launch_fri_fold(const uint32_t* in, uint32_t* out, const uint32_t* itw, uint64_t half, const Ext& beta, hipStream_t s) {
    ZK_LAUNCH(fri_fold_kernel, fri_fold_kernel_batch, fri_fold_kernel_bargs, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, s, in, out, itw, half, beta);
    return hipGetLastError();
}

__device__ __forceinline__ void fri_fold_dev_kernel_body(const uint32_t* in, uint32_t* out, const uint32_t* itw, uint64_t half, const uint32_t* beta_ptr, int squarings) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    Ext beta = ld_ext(beta_ptr);
    for (int j = 0; j < squarings; j++) beta = ext_mul_dev(beta, beta);
    const Ext e0 = ld_ext(in + 8 * i), e1 = ld_ext(in + 8 * i + 4);
    const Ext s = ext_mul_base_dev(ext_add(e0, e1), MONTY_INV2);
    const Ext d = ext_mul_base_dev(ext_sub(e0, e1), itw[i]);
    st_ext(out + 4 * i, ext_add(s, ext_mul_dev(beta, d)));
}
__global__ void __launch_bounds__(256) fri_fold_dev_kernel(const uint32_t* in, uint32_t* out, const uint32_t* itw, uint64_t half, const uint32_t* beta_ptr, int squarings) { fri_fold_dev_kernel_body(in, out, itw, half, beta_ptr, squarings); }
struct fri_fold_dev_kernel_bargs { const uint32_t* in; uint32_t* out; const uint32_t* itw; uint64_t half; const uint32_t* beta_ptr; int squarings; static fri_fold_dev_kernel_bargs make(const uint32_t* in, uint32_t* out, const uint32_t* itw, uint64_t half, const uint32_t* beta_ptr, int squarings) { return fri_fold_dev_kernel_bargs{in, out, itw, half, beta_ptr, squarings}; } };
__global__ void __launch_bounds__(256) fri_fold_dev_kernel_batch(const fri_fold_dev_kernel_bargs* __restrict__ zk_arr) { const fri_fold_dev_kernel_bargs& zk_b = zk_arr[blockIdx.z]; fri_fold_dev_kernel_body(zk_b.in, zk_b.out, zk_b.itw, zk_b.half, zk_b.beta_ptr, zk_b.squarings); }

hipError_t launch_fri_fold_dev(const uint32_t* in, uint32_t* out, const uint32_t* itw, uint64_t half, const uint32_t* beta_ptr,
                               int squarings, hipStream_t s) {
    ZK_LAUNCH(fri_fold_dev_kernel, fri_fold_dev_kernel_batch, fri_fold_dev_kernel_bargs, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, s, in, out, itw, half, beta_ptr, squarings);
    return hipGetLastError();
}

// dst[i] += src[i] over `count` extension elements (a shorter chip's reduced openings joining the FRI vector)
__device__ __forceinline__ void ext_add_kernel_body(uint32_t* dst, const uint32_t* src, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    st_ext(dst + 4 * i, ext_add(ld_ext(dst + 4 * i), ld_ext(src + 4 * i)));
}
__global__ void __launch_bounds__(256) ext_add_kernel(uint32_t* dst, const uint32_t* src, uint64_t count) { ext_add_kernel_body(dst, src, count); }
struct ext_add_kernel_bargs { uint32_t* dst; const uint32_t* src; uint64_t count; static ext_add_kernel_bargs make(uint32_t* dst, const uint32_t* src, uint64_t count) { return ext_add_kernel_bargs{dst, src, count}; } };
__global__ void __launch_bounds__(256) ext_add_kernel_batch(const ext_add_kernel_bargs* __restrict__ zk_arr) { const ext_add_kernel_bargs& zk_b = zk_arr[blockIdx.z]; ext_add_kernel_body(zk_b.dst, zk_b.src, zk_b.count); }

hipError_t launch_ext_add(uint32_t* dst, const uint32_t* src, uint64_t count, hipStream_t s) {
    ZK_LAUNCH(ext_add_kernel, ext_add_kernel_batch, ext_add_kernel_bargs, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, dst, src, count);
    return hipGetLastError();
}

// ------------------------------------------------------------------ proof-of-work search
// state: sponge state with the pending inputs already written to words [0, slot);
// candidate w goes to word `slot`; hit when canonical(permute(state)[7]) & mask == 0.
__device__ __forceinline__ void grind_kernel_body(const GrindArgs& a, uint32_t base, uint32_t* result) {
    const uint32_t w0 = base + blockIdx.x * blockDim.x, w = w0 + threadIdx.x;
    // the smallest witness is wanted and candidates are scanned in order: a block whose candidates all lie above a hit already
    // recorded cannot lower it (workgroups start roughly in index order, so most of a launch ends here once a hit is in)
    if (__atomic_load_n(result, __ATOMIC_RELAXED) <= w0) return;
    if (w >= P) return;
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = a.state[i];
    const uint32_t wm = fmul(w, MONTY_R2);
#pragma unroll
    for (int i = 0; i < 8; i++) if (i == a.slot) s[i] = wm;
    p2_permute_dev(s);
    if ((from_monty(s[7]) & a.mask) == 0) atomicMin(result, w);
}
__global__ void __launch_bounds__(256) grind_kernel(GrindArgs a, uint32_t base, uint32_t* result) { grind_kernel_body(a, base, result); }
struct grind_kernel_bargs { GrindArgs a; uint32_t base; uint32_t* result; static grind_kernel_bargs make(GrindArgs a, uint32_t base, uint32_t* result) { return grind_kernel_bargs{a, base, result}; } };
__global__ void __launch_bounds__(256) grind_kernel_batch(const grind_kernel_bargs* __restrict__ zk_arr) { const grind_kernel_bargs& zk_b = zk_arr[blockIdx.z]; grind_kernel_body(zk_b.a, zk_b.base, zk_b.result); }

hipError_t launch_grind(const GrindArgs& a, uint32_t base, uint32_t count, uint32_t* result, hipStream_t s) {
    ZK_LAUNCH(grind_kernel, grind_kernel_batch, grind_kernel_bargs, dim3((count + 255) / 256), dim3(256), 0, s, a, base, result);
    return hipGetLastError();
}

// ------------------------------------------------------------------ query gather
// one wave per descriptor: copy nwords from src, Montgomery -> canonical, to dst + off
__device__ __forceinline__ void gather_kernel_body(const GatherDesc* descs, uint32_t ndesc, uint32_t* dst) {
    const uint32_t d = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (d >= ndesc) return;
    const GatherDesc g = descs[d];
    for (uint32_t i = lane; i < g.nwords; i += 64) dst[g.dst_off + i] = from_monty(g.src[i]);
}
__global__ void __launch_bounds__(256) gather_kernel(const GatherDesc* descs, uint32_t ndesc, uint32_t* dst) { gather_kernel_body(descs, ndesc, dst); }
struct gather_kernel_bargs { const GatherDesc* descs; uint32_t ndesc; uint32_t* dst; static gather_kernel_bargs make(const GatherDesc* descs, uint32_t ndesc, uint32_t* dst) { return gather_kernel_bargs{descs, ndesc, dst}; } };
__global__ void __launch_bounds__(256) gather_kernel_batch(const gather_kernel_bargs* __restrict__ zk_arr) { const gather_kernel_bargs& zk_b = zk_arr[blockIdx.z]; gather_kernel_body(zk_b.descs, zk_b.ndesc, zk_b.dst); }

hipError_t launch_gather(const GatherDesc* descs, uint32_t ndesc, uint32_t* dst, hipStream_t s) {
    if (ndesc == 0) return hipSuccess;
    const uint64_t threads = (uint64_t)ndesc * 64;
    ZK_LAUNCH(gather_kernel, gather_kernel_batch, gather_kernel_bargs, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, descs, ndesc, dst);
    return hipGetLastError();
}

}  // namespace zk
