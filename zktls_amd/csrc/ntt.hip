// ntt.hip -- batched radix-2 NTT passes over the columns of a row-major BabyBear matrix,
// hand-written for gfx950 (wave64, 160 KiB LDS/CU).
//
// Replaces p3-dft 0.2.1-succinct Radix2DitParallel::{dft_batch, coset_lde_batch}
// (reference Cargo.lock:3903) on the path below crates/guest-prover-sp1/src/sp1.rs:116.
//
// Design.  A column transform of N = M1*M2 points runs as two
// launches of this one kernel (four-step factorisation); a launch moves every element
// HBM -> registers -> HBM exactly once (8 B/element algorithmic traffic).  A workgroup
// owns a tile of M = 32*P rows x C columns (P = 2^b threads per column, b <= 5):
//   load      thread (u, c) pulls x[u + P*n1], n1 = 0..31 straight into 32 VGPRs; the
//             64 lanes of a wave cover 4 consecutive tile rows x 16 columns (4 x 64 B);
//   phase A   32-point DIF in registers, twiddles are compile-time constants (centred, scalar registers);
//   twiddle   * w_M^(u*k1), table staged in LDS (broadcast reads);
//   exchange  one pass through LDS, row pitch (P+1)*C words -> conflict-free both ways;
//   phase B   the last b stages of the same 32-point network = P-point DIFs;
//   store     * post[tile][k] (inter-pass twiddle / coset shift / 1/N, staged in LDS),
//             written natural or bit-reversed inside the tile.
// Launch side (launch_ntt_pass): cache policy (non-temporal loads and stores except on a strided pass run in place) and tile order
// (XCD-aware; strided passes of 2^20-row transforms rotate the tile index by 4 bits) are chosen per launch.
// No MFMA: a 31-bit modular butterfly is not a dense contraction.
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "babybear.cuh"
#include "kernels.h"
#include "batch.h"
#include "ntt_bfly.cuh"

namespace zk {

constexpr int MAX_DEVICES = 64;     // device ordinals a process can hold contexts on

// CPT = columns per thread.  CPT = 2 doubles the row chunk a wave touches to 128 B (16 lanes
// x 8 B), which the HBM likes much better than 64 B (tools/microbench: 5.1-5.5 vs 3.8-4.8
// TB/s for the same tile shapes), and halves the twiddle-table LDS reads per element.  The
// LDS exchange then runs once per column, reusing one 64 KiB buffer, so two workgroups
// still fit a CU.
// BFIX >= 0 fixes the tile height at compile time (M = 32 << BFIX): every LDS / store offset of the
// exchange and of phase B then folds into an instruction immediate instead of a VGPR.
// NT != 0: non-temporal cache policy on the tile's loads and stores (chosen per launch, see launch_ntt_pass).
template <int LOG_C, bool INV, int CPT, int BFIX = -1, int NT = 0>
__device__ __forceinline__ void ntt_pass_kernel_body(const NttPassArgs& a) {
    // NT = 1 and NT = 2 are the same code: two names, so that a profile tells the passes apart.  NT = 3 / 4 are again the same
    // code as 1 / 2: the names under which zkhip_ntt_pass (the roofline hook) launches, so that a rocprofv3 summary of bench.py
    // lists the isolated, event-timed launches apart from the in-proof ones (which overlap with the other shards in flight and
    // are stretched by that).  NT >= 256 (policy sweep builds
    // only, -DNTT_POLICY_SWEEP): load policy in bits 8..15, store policy in bits 16..23 (sc0 = 1, nt = 2, sc1 = 16).
    constexpr int AUX = NT >= 256 ? ((NT >> 8) & 0xff) : (NT ? 2 : 0);
    constexpr int ST_AUX = NT >= 256 ? ((NT >> 16) & 0xff) : (NT ? 2 : 0);
    extern __shared__ uint32_t lds[];
    constexpr int C = 1 << LOG_C;          // lanes along the row chunk
    constexpr int TC = C * CPT;            // tile columns
    const int b = BFIX >= 0 ? BFIX : (int)a.log_m - 5;
    const int Pn = 1 << b;
    const int M = 32 << b;
    const int pitch = (Pn + 1) * C;
    uint32_t* sdata = lds;
    uint32_t* stw = lds + 32 * pitch;
    uint32_t* spost = stw + M;
    uint32_t* spre = spost + M;

    constexpr bool SD = BFIX == 5;                   // signed differences out of the lazy stages (1024-row tiles only)
    int64_t bias = (int64_t)((uint64_t)P << 32);     // dbfly_mul's addend: kept opaque so that it stays the addend of a mad
    asm volatile("" : "+v"(bias));
    const int tid = threadIdx.x;
    const int c = tid & (C - 1);
    const int u = tid >> LOG_C;

    const uint32_t ncg = (a.ncols + TC - 1) / TC;
    uint32_t tile, cg;
    if (a.map_mode == 1 || a.map_mode >= 2) {
        // blocks b and b+8 share an XCD (round-robin dispatch): keep the column groups
        // of one tile on one XCD so that they share L2 lines and DRAM pages
        const uint32_t xcd = blockIdx.x & 7u, l = blockIdx.x >> 3;
        cg = l % ncg;
        tile = (l / ncg) * 8u + xcd;
        if (a.map_mode >= 2) {
            // A/B: spread the tiles that are resident together over the whole tile range (tile index rotated by map_mode
            // bits inside log2(num_tiles) bits; num_tiles is a power of two here)
            const uint32_t lt = 31u - __clz(a.num_tiles), r = a.map_mode;
            if (lt > r) tile = ((tile << r) | (tile >> (lt - r))) & (a.num_tiles - 1u);
        }
#ifdef ZKHIP_AB_HOOKS
        if (a.tile_perm) {
            const uint32_t lt = 31u - __clz(a.num_tiles);
            uint32_t t2 = 0;
            for (uint32_t b = 0; b < lt; b++) t2 |= ((tile >> b) & 1u) << ((a.tile_perm >> (4u * b)) & 15u);
            tile = t2;
        }
#endif
    } else {
        cg = blockIdx.x % ncg;
        tile = blockIdx.x / ncg;
    }
    const uint32_t col = cg * TC + c * CPT;
    const bool active = col < a.ncols;      // CPT = 2 requires an even width: both or none
    const uint32_t lcol = active ? col : 0u;   // inactive lanes load a valid address, never store

    // ---- global -> registers (32 independent loads in flight per lane).  Buffer addressing:
    // wave-uniform descriptor + one 32-bit lane offset + a scalar offset per row group, so the
    // 64 address VGPRs of flat addressing are free for data (launch_ntt_pass checks the 4 GiB span).
    uint32_t x[CPT][32];
    {
        const uint32_t* ib = a.in + (uint64_t)tile * a.in_tile_mul * a.in_ld;
#ifdef ZKHIP_AB_HOOKS
        // timing-only knob of A/B builds (tools/archive/ab_ntt2.sh): a zero-record descriptor drops the loads / stores
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(ib), 0, (a.debug_flags & 1u) ? 0u : 0xFFFFFFFFu, 0x00020000);
#else
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(ib), 0, 0xFFFFFFFFu, 0x00020000);
#endif
        const uint32_t voff = 4u * ((uint32_t)((uint64_t)u * a.in_stride * a.in_ld) + lcol);
        const uint32_t istep_b = (uint32_t)(4u * (uint64_t)Pn * a.in_stride * a.in_ld);
#pragma unroll
        for (int n1 = 0; n1 < 32; n1++) {
            if (CPT == 2) {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, n1 * istep_b, AUX);
                x[0][n1] = v.x; x[CPT - 1][n1] = v.y;
            } else {
                x[0][n1] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, n1 * istep_b, 0);
            }
        }
    }
    // ---- twiddle tables -> LDS while the loads fly
    const bool has_post = a.post != nullptr, has_pre = a.pre != nullptr;
    {
        const int sh = 5 - b;
        for (int i = tid; i < M; i += blockDim.x) {
            stw[i] = a.w1024[i << sh];
            if (has_post) spost[i] = a.post[(uint64_t)tile * M + i];
            if (has_pre) spre[i] = a.pre[i];
        }
    }
    __syncthreads();
#ifdef ZKHIP_AB_HOOKS
    if (BFIX == 5 && (a.debug_flags & 4u)) {
        // timing-only: the same loads and stores with no transform in between (the memory floor of this access pattern)
        if (active) {
            uint32_t* ob = a.out + (uint64_t)tile * a.out_tile_mul * a.out_ld;
            const auto ors = __builtin_amdgcn_make_buffer_rsrc(ob, 0, 0xFFFFFFFFu, 0x00020000);
            const uint32_t ostep_b = (uint32_t)(4u * a.out_stride * a.out_ld);
            const uint32_t out_off = (a.bitrev_out ? 32u * (__brev((uint32_t)u) >> 27) : (uint32_t)u) * ostep_b + 4u * col;
#pragma unroll
            for (int rho = 0; rho < 32; rho++) {
                const uint32_t ro = a.bitrev_out ? (uint32_t)rho : (uint32_t)(32 * rev5(rho));
                if (CPT == 2) { u32x2 v; v.x = x[0][rho]; v.y = x[CPT - 1][rho]; __builtin_amdgcn_raw_buffer_store_b64(v, ors, out_off, ro * ostep_b, AUX); }
                else __builtin_amdgcn_raw_buffer_store_b32(x[0][rho], ors, out_off, ro * ostep_b, 0);
            }
        }
        return;
    }
#endif
    if (has_pre) {
#pragma unroll
        for (int n1 = 0; n1 < 32; n1++) {
            const uint32_t w = spre[u + Pn * n1];
#pragma unroll
            for (int cc = 0; cc < CPT; cc++) x[cc][n1] = dmul(x[cc][n1], w);
        }
    }

    // ---- phase A: 32-point DIF over n1; x[r] <- A[rev5(r)], then the tile twiddle w_M^(u*k1).
    // One column at a time (scheduling barrier in between): interleaving the two columns of
    // CPT = 2 for ILP is what pushed the register allocation over 128 and into scratch.
#pragma unroll
    for (int cc = 0; cc < CPT; cc++) {
        dif_stage<INV, 0>(x[cc], bias);
        dif_stage<INV, 1>(x[cc], bias);
        dif_stage<INV, 2>(x[cc], bias);
        dif_stage<INV, 3>(x[cc], bias);
        dif_stage<INV, 4, true, SD>(x[cc], bias);    // lazy: every output is multiplied (or reduced) next
#pragma unroll
        for (int r = 1; r < 32; r++) x[cc][r] = (SD && (r & 1)) ? dmul_sd(x[cc][r], stw[u * rev5(r)], bias) : dmul(x[cc][r], stw[u * rev5(r)]);
        x[cc][0] = dred(x[cc][0]);
        if (CPT > 1) __builtin_amdgcn_sched_barrier(0);
    }

    // ---- exchange through LDS, one column per round
#pragma unroll
    for (int cc = 0; cc < CPT; cc++) {
        if (cc) __syncthreads();           // previous round's reads are done
        uint32_t* wp = sdata + u * C + c;
#pragma unroll
        for (int r = 0; r < 32; r++) wp[rev5(r) * pitch] = x[cc][r];
        __syncthreads();
        const uint32_t* rp = sdata + c;
#pragma unroll
        for (int rho = 0; rho < 32; rho++) {
            const int t = rho & (Pn - 1);
            const int k1 = (rho & ~(Pn - 1)) + u;
            x[cc][rho] = rp[k1 * pitch + t * C];
        }
    }

    // ---- phase B: P-point DIFs = the last b stages of the 32-point network
#pragma unroll
    for (int cc = 0; cc < CPT; cc++) {
        if (b >= 5) dif_stage<INV, 0>(x[cc], bias);
        if (b >= 4) dif_stage<INV, 1>(x[cc], bias);
        if (b >= 3) dif_stage<INV, 2>(x[cc], bias);
        if (b >= 2) dif_stage<INV, 3>(x[cc], bias);
        if (b >= 1) {
            if (has_post) dif_stage<INV, 4, true, SD>(x[cc], bias);   // outputs go straight into the post multiplication
            else dif_stage<INV, 4>(x[cc], bias);
        }
        if (CPT > 1) __builtin_amdgcn_sched_barrier(0);
    }

    // ---- post multiply and store
    if (BFIX == 5) {
        // M = 1024: element k = 32 rev5(rho) + u goes to tile row o(k) = k, or bitrev10(k) = 32 rev5(u) + rho;
        // either way o * ostep splits into a per-lane 32-bit offset and a wave-uniform (scalar) part
        if (has_post) {
#pragma unroll
            for (int rho = 0; rho < 32; rho++) {
                const uint32_t w = spost[32 * rev5(rho) + u];
#pragma unroll
                for (int cc = 0; cc < CPT; cc++) x[cc][rho] = (SD && (rho & 1)) ? dmul_sd(x[cc][rho], w, bias) : dmul(x[cc][rho], w);
            }
        }
        if (active) {
            uint32_t* ob = a.out + (uint64_t)tile * a.out_tile_mul * a.out_ld;
#ifdef ZKHIP_AB_HOOKS
            const auto ors = __builtin_amdgcn_make_buffer_rsrc(ob, 0, (a.debug_flags & 2u) ? 0u : 0xFFFFFFFFu, 0x00020000);
#else
            const auto ors = __builtin_amdgcn_make_buffer_rsrc(ob, 0, 0xFFFFFFFFu, 0x00020000);
#endif
            const uint32_t ostep_b = (uint32_t)(4u * a.out_stride * a.out_ld);
            const uint32_t out_off = (a.bitrev_out ? 32u * (__brev((uint32_t)u) >> 27) : (uint32_t)u) * ostep_b + 4u * col;
#pragma unroll
            for (int rho = 0; rho < 32; rho++) {
                const uint32_t ro = a.bitrev_out ? (uint32_t)rho : (uint32_t)(32 * rev5(rho));
                if (CPT == 2) { u32x2 v; v.x = x[0][rho]; v.y = x[CPT - 1][rho]; __builtin_amdgcn_raw_buffer_store_b64(v, ors, out_off, ro * ostep_b, ST_AUX); }
                else __builtin_amdgcn_raw_buffer_store_b32(x[0][rho], ors, out_off, ro * ostep_b, 0);
            }
        }
        return;
    }
    // ---- post multiply and store
    if (has_post) {
#pragma unroll
        for (int rho = 0; rho < 32; rho++) {
            const uint32_t t = rho & (Pn - 1);
            const uint32_t k1 = (rho & ~(Pn - 1)) + u;
            const uint32_t k0 = b ? (__brev(t) >> (32 - b)) : 0u;
            const uint32_t w = spost[32u * k0 + k1];
#pragma unroll
            for (int cc = 0; cc < CPT; cc++) x[cc][rho] = dmul(x[cc][rho], w);
        }
    }
    if (active) {
        const int m = (int)a.log_m;
        uint32_t* ob = a.out + (uint64_t)tile * a.out_tile_mul * a.out_ld;
#ifdef ZKHIP_AB_HOOKS
        const auto ors = __builtin_amdgcn_make_buffer_rsrc(ob, 0, (a.debug_flags & 2u) ? 0u : 0xFFFFFFFFu, 0x00020000);
#else
        const auto ors = __builtin_amdgcn_make_buffer_rsrc(ob, 0, 0xFFFFFFFFu, 0x00020000);
#endif
        const uint32_t ostep_b = (uint32_t)(4u * a.out_stride * a.out_ld);
#pragma unroll
        for (int rho = 0; rho < 32; rho++) {
            const uint32_t t = rho & (Pn - 1);
            const uint32_t k1 = (rho & ~(Pn - 1)) + u;
            const uint32_t k0 = b ? (__brev(t) >> (32 - b)) : 0u;
            const uint32_t k = 32u * k0 + k1;
            const uint32_t o = a.bitrev_out ? (__brev(k) >> (32 - m)) : k;
            const uint32_t off = o * ostep_b + 4u * col;
            if (CPT == 2) { u32x2 v; v.x = x[0][rho]; v.y = x[CPT - 1][rho]; __builtin_amdgcn_raw_buffer_store_b64(v, ors, off, 0, 0); }
            else __builtin_amdgcn_raw_buffer_store_b32(x[0][rho], ors, off, 0, 0);
        }
    }
}
template <int LOG_C, bool INV, int CPT, int BFIX = -1, int NT = 0>
__global__ void __launch_bounds__(32 << LOG_C, 4) ntt_pass_kernel(NttPassArgs a) { ntt_pass_kernel_body<LOG_C, INV, CPT, BFIX, NT>(a); }
struct ntt_pass_kernel_bargs { NttPassArgs a; static ntt_pass_kernel_bargs make(NttPassArgs a) { return ntt_pass_kernel_bargs{a}; } };
template <int LOG_C, bool INV, int CPT, int BFIX = -1, int NT = 0>
__global__ void __launch_bounds__(32 << LOG_C, 4) ntt_pass_kernel_batch(const ntt_pass_kernel_bargs* __restrict__ zk_arr) { const ntt_pass_kernel_bargs& zk_b = zk_arr[blockIdx.z]; ntt_pass_kernel_body<LOG_C, INV, CPT, BFIX, NT>(zk_b.a); }



#ifdef ZKHIP_AB_HOOKS
// ------------------------------------------------------------------ persistent fast path
// M = 1024 rows x 32 columns per tile, one 1024-thread workgroup per CU that walks its tiles:
//   * 128-byte row chunks (a wave = 2 rows x 32 columns) -- the HBM likes them far better than
//     the 64-byte chunks of the generic kernel (tools/microbench: 5.1-5.5 vs 3.8-4.8 TB/s);
//   * the 32 loads of the NEXT tile are issued before the current tile is transformed and stay
//     in flight behind counted waits (raw s_barrier + lgkmcnt(0) only, never vmcnt(0)), so HBM
//     latency hides under ~9 us of butterflies instead of being exposed at every tile start;
//   * stores are fire-and-forget.
// LDS: exchange buffer 32 x 33 x 32 words + three 1024-word tables = 147 456 B (one per CU).
ZK_D void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct ItemPos { uint32_t tile, cg; };
ZK_D ItemPos item_pos(uint32_t item, uint32_t ncg, uint32_t map_mode) {
    ItemPos p;
    if (map_mode == 1) {
        const uint32_t xcd = item & 7u, l = item >> 3;
        p.cg = l % ncg;
        p.tile = (l / ncg) * 8u + xcd;
    } else {
        p.cg = item % ncg;
        p.tile = item / ncg;
    }
    return p;
}

// Two columns per lane, 512 threads (2 waves per SIMD, up to 256 VGPRs): the register file
// then holds this tile (64 values) AND the prefetched next tile (64 values) without spills,
// loads/stores are 8 bytes per lane (a wave = 4 rows x 128 B) and the LDS exchange moves
// both columns with one ds_write_b64 / ds_read_b64.
template <bool INV>
__global__ void __launch_bounds__(512) ntt_pass1024x2_kernel(NttPassArgs a, uint32_t total_items) {
    int64_t bias = (int64_t)((uint64_t)P << 32);
    asm volatile("" : "+v"(bias));
    extern __shared__ uint32_t lds[];
    constexpr int C = 32, C2 = 16, Pn = 32, M = 1024, pitch2 = (Pn + 1) * C2;   // pitch in column pairs
    uint2* sdata = reinterpret_cast<uint2*>(lds);
    uint32_t* stw = lds + 2 * 32 * pitch2;
    uint32_t* spost = stw + M;
    uint32_t* spre = spost + M;
    const int tid = threadIdx.x;
    const int c2 = tid & (C2 - 1);
    const int u = tid >> 4;
    const uint32_t ncg = a.ncols / C;
    const bool has_post = a.post != nullptr, has_pre = a.pre != nullptr;
    const uint32_t istep_b = (uint32_t)(4u * (uint64_t)Pn * a.in_stride * a.in_ld);
    const uint32_t ostep_b = (uint32_t)(4u * a.out_stride * a.out_ld);
    const uint32_t in_off = 4u * (uint32_t)((uint64_t)u * a.in_stride * a.in_ld) + 8u * (uint32_t)c2;
    // timing-only knobs (tools/ab_ntt.sh): a zero-record descriptor drops the loads / stores
    const uint32_t in_rec = (a.debug_flags & 1u) ? 0u : 0xFFFFFFFFu, out_rec = (a.debug_flags & 2u) ? 0u : 0xFFFFFFFFu;

    stw[tid] = a.w1024[tid];
    stw[tid + 512] = a.w1024[tid + 512];
    if (has_pre) { spre[tid] = a.pre[tid]; spre[tid + 512] = a.pre[tid + 512]; }

    uint32_t n0[32], n1[32];
    uint32_t item = blockIdx.x;
    {
        const ItemPos p = item_pos(item, ncg, a.map_mode);
        const uint32_t* ib = a.in + (uint64_t)p.tile * a.in_tile_mul * a.in_ld + p.cg * C;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(ib), 0, in_rec, 0x00020000);
#pragma unroll
        for (int k = 0; k < 32; k++) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, in_off, k * istep_b, 0);
            n0[k] = v.x; n1[k] = v.y;
        }
    }
    wg_barrier();
    for (; item < total_items; item += gridDim.x) {
        const ItemPos cur = item_pos(item, ncg, a.map_mode);
        uint32_t pv0 = 0, pv1 = 0;
        if (has_post) { pv0 = a.post[(uint64_t)cur.tile * M + tid]; pv1 = a.post[(uint64_t)cur.tile * M + tid + 512]; }
        uint32_t x0[32], x1[32];
#pragma unroll
        for (int k = 0; k < 32; k++) { x0[k] = n0[k]; x1[k] = n1[k]; }
        const uint32_t nitem = item + gridDim.x;
        if (nitem < total_items) {
            const ItemPos p = item_pos(nitem, ncg, a.map_mode);
            const uint32_t* ib = a.in + (uint64_t)p.tile * a.in_tile_mul * a.in_ld + p.cg * C;
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(ib), 0, in_rec, 0x00020000);
#pragma unroll
            for (int k = 0; k < 32; k++) {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, in_off, k * istep_b, 0);
                n0[k] = v.x; n1[k] = v.y;
            }
        }
        if (has_post) { spost[tid] = pv0; spost[tid + 512] = pv1; }

        if (has_pre) {
#pragma unroll
            for (int k = 0; k < 32; k++) {
                const uint32_t w = spre[u + Pn * k];
                x0[k] = dmul(x0[k], w); x1[k] = dmul(x1[k], w);
            }
        }
        dif_stage<INV, 0>(x0, bias); dif_stage<INV, 0>(x1, bias);
        dif_stage<INV, 1>(x0, bias); dif_stage<INV, 1>(x1, bias);
        dif_stage<INV, 2>(x0, bias); dif_stage<INV, 2>(x1, bias);
        dif_stage<INV, 3>(x0, bias); dif_stage<INV, 3>(x1, bias);
        dif_stage<INV, 4, true>(x0, bias); dif_stage<INV, 4, true>(x1, bias);
        {
            uint2* wp = sdata + u * C2 + c2;
#pragma unroll
            for (int r = 0; r < 32; r++) {
                const int k1 = rev5(r);
                if (k1 == 0) wp[0] = make_uint2(dred(x0[r]), dred(x1[r]));
                else { const uint32_t w = stw[u * k1]; wp[k1 * pitch2] = make_uint2(dmul(x0[r], w), dmul(x1[r], w)); }
            }
        }
        wg_barrier();
        {
            const uint2* rp = sdata + u * pitch2 + c2;       // P = 32: k1 = u, t = rho
#pragma unroll
            for (int rho = 0; rho < 32; rho++) { const uint2 v = rp[rho * C2]; x0[rho] = v.x; x1[rho] = v.y; }
        }
        dif_stage<INV, 0>(x0, bias); dif_stage<INV, 0>(x1, bias);
        dif_stage<INV, 1>(x0, bias); dif_stage<INV, 1>(x1, bias);
        dif_stage<INV, 2>(x0, bias); dif_stage<INV, 2>(x1, bias);
        dif_stage<INV, 3>(x0, bias); dif_stage<INV, 3>(x1, bias);
        uint32_t* ob = a.out + (uint64_t)cur.tile * a.out_tile_mul * a.out_ld + cur.cg * C;
        const auto ors = __builtin_amdgcn_make_buffer_rsrc(ob, 0, out_rec, 0x00020000);
        const uint32_t out_off = (a.bitrev_out ? 32u * (__brev((uint32_t)u) >> 27) : (uint32_t)u) * ostep_b + 8u * (uint32_t)c2;
        if (has_post) {
            dif_stage<INV, 4, true>(x0, bias); dif_stage<INV, 4, true>(x1, bias);
#pragma unroll
            for (int rho = 0; rho < 32; rho++) {
                const uint32_t w = spost[32 * rev5(rho) + u];
                const uint32_t ro = a.bitrev_out ? (uint32_t)rho : (uint32_t)(32 * rev5(rho));
                u32x2 v; v.x = dmul(x0[rho], w); v.y = dmul(x1[rho], w);
                __builtin_amdgcn_raw_buffer_store_b64(v, ors, out_off, ro * ostep_b, 0);
            }
        } else {
            dif_stage<INV, 4>(x0, bias); dif_stage<INV, 4>(x1, bias);
#pragma unroll
            for (int rho = 0; rho < 32; rho++) {
                const uint32_t ro = a.bitrev_out ? (uint32_t)rho : (uint32_t)(32 * rev5(rho));
                u32x2 v; v.x = x0[rho]; v.y = x1[rho];
                __builtin_amdgcn_raw_buffer_store_b64(v, ors, out_off, ro * ostep_b, 0);
            }
        }
        wg_barrier();
    }
}

static int cu_count() {
    static const int n = [] {         // initialised once, thread-safe
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

template <bool INV>
static hipError_t launch_ntt1024x2(const NttPassArgs& a, hipStream_t s) {
    const uint32_t total = a.num_tiles * (a.ncols / 32);
    const uint32_t grid = total < (uint32_t)cu_count() ? total : (uint32_t)cu_count();
    const size_t lds = (size_t)(2 * 32 * 33 * 16 + 3 * 1024) * sizeof(uint32_t);
    static std::atomic<bool> configured[MAX_DEVICES] = {};       // several host threads (one context each) launch concurrently
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    if (!configured[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)ntt_pass1024x2_kernel<INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((ntt_pass1024x2_kernel<INV>), dim3(grid), dim3(512), lds, s, a, total);
    return hipGetLastError();
}

#endif  // ZKHIP_AB_HOOKS (persistent A/B kernel)

static size_t ntt_lds_bytes(int log_m, int log_c) {
    int b = log_m - 5, Pn = 1 << b, M = 32 << b, C = 1 << log_c;
    return (size_t)(32 * (Pn + 1) * C + 3 * M) * sizeof(uint32_t);
}

template <int LOG_C, bool INV, int CPT, int BFIX = -1, int NT = 0>
static hipError_t launch_ntt_k(const NttPassArgs& a, hipStream_t s) {
    const int TC = (1 << LOG_C) * CPT;
    const uint32_t ncg = (a.ncols + TC - 1) / TC;
    const int b = (int)a.log_m - 5;
    dim3 grid(a.num_tiles * ncg), block((1 << b) << LOG_C);
    const size_t lds = ntt_lds_bytes((int)a.log_m, LOG_C);
    // raise the dynamic-LDS cap once per instantiation AND per device (function attributes are per device; a process may hold
    // contexts on every GPU of the node: zkhip_prove_shards_multi); thread-safe, several host threads launch concurrently
    static std::atomic<size_t> configured[MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    if (lds > configured[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)ntt_pass_kernel<LOG_C, INV, CPT, BFIX, NT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)ntt_pass_kernel_batch<LOG_C, INV, CPT, BFIX, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[dev].store(lds, std::memory_order_release);
    }
    ZK_LAUNCH((ntt_pass_kernel<LOG_C, INV, CPT, BFIX, NT>), (ntt_pass_kernel_batch<LOG_C, INV, CPT, BFIX, NT>), ntt_pass_kernel_bargs, grid, block, lds, s, a);
    return hipGetLastError();
}

hipError_t launch_ntt_pass(const NttPassArgs& a_, bool inverse, hipStream_t s) {
    NttPassArgs a = a_;
    if (a.log_m < 5 || a.log_m > 10) return hipErrorInvalidValue;
    {   // buffer addressing: every byte offset inside one tile must fit 32 bits
        const uint64_t M = 1ull << a.log_m;
        const uint64_t in_span = 4ull * ((M - 1) * a.in_stride * a.in_ld + a.ncols);
        const uint64_t out_span = 4ull * ((M - 1) * a.out_stride * a.out_ld + a.ncols);
        if (in_span >= (1ull << 32) || out_span >= (1ull << 32)) return hipErrorInvalidValue;
    }
    // Block -> tile order.  255 = automatic: XCD-aware (1) for block passes; for a strided pass additionally the tile index rotated
    // by 4 bits (4), so that the ~64 tiles resident at any moment are spread over the whole 1 MiB row period instead of being 64
    // neighbouring rows.  With neighbours the reads and the writes of the chip sit in one narrow address window at a time, and when source
    // and destination happen to map that window onto the same channels the pass drops from 0.48 to 0.53 ms (two classes of buffers,
    // slow iff both are in the same class: tools/archive/ntt_buffer_quality.py); spread out, both cases run at 0.486-0.494 ms (tools/archive/ntt_map_ab.py).
    const bool pow2_tiles = (a.num_tiles & (a.num_tiles - 1u)) == 0;
    if (a.map_mode >= 100u && a.map_mode < 116u) {           // A/B: 100 + r = rotation r for block-in / strided-out passes only, automatic otherwise
        const uint32_t r = a.map_mode - 100u;
        a.map_mode = (a.in_stride == 1 && a.out_stride != 1) ? (r == 0 ? 1u : r) : 255u;
    }
    // A pass that reads strided but writes every tile as ONE block (the LDE's first pass in front of the fused launch) is the exception: its
    // writes are streams of their own, the rotation only scatters its reads -- 0.45 ms with plain XCD order on every (trace, workspace) pair
    // against 0.46 ... 0.52 ms rotated (tools/archive/i1_map_sweep.py); strided -> strided keeps the rotation (0.50 against 0.52), block -> strided is indifferent.
    const bool strided_in_block_out = a.in_stride != 1 && a.out_stride == 1;
    if (a.map_mode == 255u) a.map_mode = (a.in_stride != 1 || a.out_stride != 1) && !strided_in_block_out && pow2_tiles && a.num_tiles >= 1024u ? 4u : 1u;     // smaller transforms lose 3-8 % with it
    if (a.map_mode == 1 && (a.num_tiles % 8u) != 0) a.map_mode = 0;
    if (a.map_mode >= 2 && ((a.num_tiles % 8u) != 0 || !pow2_tiles)) a.map_mode = 0;
    // A/B only (fast_path 1): the persistent 1024 x 32 kernel (one workgroup per CU, next tile prefetched into
    // registers).  Measured on 2^20 x 256 it loses to the two-workgroups-per-CU kernel below once that one has
    // compile-time tile offsets (0.62 / 0.55 ms against 0.53 / 0.46 ms for the strided / contiguous pass); so does a
    // persistent form of the two-workgroup kernel itself (0.55 / 0.50 ms: workgroups that walk tile lists stay
    // phase-locked across the chip, freshly dispatched ones drift apart and keep the memory pipe fed).
#ifdef ZKHIP_AB_HOOKS
    const bool al8 = a.in_ld % 2 == 0 && a.out_ld % 2 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 7) == 0 &&
                     (reinterpret_cast<uintptr_t>(a.out) & 7) == 0;
    if (a.log_m == 10 && a.ncols >= 32 && a.ncols % 32 == 0 && a.fast_path == 1 && al8) {
        if (a.map_mode != 1 || (a.num_tiles % 8u) != 0 || (cu_count() % 8) != 0) a.map_mode = 0;
        return inverse ? launch_ntt1024x2<true>(a, s) : launch_ntt1024x2<false>(a, s);
    }
#endif
    // two columns per lane (128-byte row chunks per 16 lanes) need 8-byte aligned row chunks.  Default for
    // 1024-row tiles, where the compile-time tile height keeps it at 116 VGPRs (two workgroups per CU);
    // with a run-time tile height it spills, so there it stays opt-in (cols_per_thread = 2).
    const bool pair_ok = a.ncols >= 32 && a.ncols % 2 == 0 && a.in_ld % 2 == 0 && a.out_ld % 2 == 0 &&
                         (reinterpret_cast<uintptr_t>(a.in) & 7) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 7) == 0;
#ifdef ZKHIP_AB_HOOKS
    {   // A/B (tools/archive/i1_wide_ab.py): the LDE's first pass with 64-column tiles -- 32 lanes x 8 B = 256-byte row chunks, 1024 lanes, one workgroup per CU
        static const bool wide = getenv("ZKHIP_I1_WIDE") != nullptr;
        if (wide && pair_ok && a.log_m == 10 && strided_in_block_out && a.ncols % 64 == 0 && inverse) return launch_ntt_k<5, true, 2, 5, 2>(a, s);
        static const bool wide_f2 = getenv("ZKHIP_F2_WIDE") != nullptr;
        if (wide_f2 && pair_ok && a.log_m == 10 && a.in_stride == 1 && a.out_stride == 1 && a.ncols % 64 == 0 && !inverse) return launch_ntt_k<5, false, 2, 5, 1>(a, s);
    }
#endif
    if (pair_ok && a.log_m == 10 && a.cols_per_thread != 1) {
        // non-temporal loads AND stores: the tile is read once and written once, its lines need not stay in L2 / MALL.
        // Pays on every pass except a strided one run in place (+6 %), which keeps the default policy; tools/archive/ntt_policy_sweep.sh
        // walks the sc0 / sc1 / nt combinations (strided pass 0.537 -> 0.487 ms, block pass 0.457 -> 0.447 ms on the same box).
        const bool contiguous = a.in_stride == 1 && a.out_stride == 1;
        const bool in_place = a.in == a.out;
#ifdef NTT_POLICY_SWEEP
        if (const char* pol = std::getenv("ZKHIP_NTT_POL")) {       // "<load>,<store>" policy bits, forward passes only
            int l = 0, st = 0;
            std::sscanf(pol, "%d,%d", &l, &st);
#define POLCASE(L, S) if (l == L && st == S && !inverse) return launch_ntt_k<4, false, 2, 5, 256 | (L << 8) | (S << 16)>(a, s);
            POLCASE(0, 2) POLCASE(0, 3) POLCASE(0, 18) POLCASE(0, 17) POLCASE(0, 19) POLCASE(0, 1) POLCASE(0, 16)
            POLCASE(2, 2) POLCASE(2, 3) POLCASE(2, 18) POLCASE(2, 17) POLCASE(2, 19) POLCASE(2, 1) POLCASE(2, 16)
            POLCASE(3, 0) POLCASE(18, 0) POLCASE(19, 0) POLCASE(17, 0) POLCASE(1, 0) POLCASE(16, 0)
            POLCASE(3, 2) POLCASE(18, 2) POLCASE(18, 18) POLCASE(3, 3)
#undef POLCASE
        }
#endif
#ifdef ZKHIP_AB_HOOKS
        if (!(a.debug_flags & 8u))
#endif
        {
            if (a.bench_tag) {
                if (contiguous) return inverse ? launch_ntt_k<4, true, 2, 5, 3>(a, s) : launch_ntt_k<4, false, 2, 5, 3>(a, s);
                if (!in_place) return inverse ? launch_ntt_k<4, true, 2, 5, 4>(a, s) : launch_ntt_k<4, false, 2, 5, 4>(a, s);
            }
            if (contiguous) return inverse ? launch_ntt_k<4, true, 2, 5, 1>(a, s) : launch_ntt_k<4, false, 2, 5, 1>(a, s);
            if (!in_place) return inverse ? launch_ntt_k<4, true, 2, 5, 2>(a, s) : launch_ntt_k<4, false, 2, 5, 2>(a, s);
        }
        return inverse ? launch_ntt_k<4, true, 2, 5>(a, s) : launch_ntt_k<4, false, 2, 5>(a, s);
    }
    // 64 .. 512-row tiles (the two passes of a 2^11 .. 2^18-row transform): the same two-columns-per-lane form with the tile height fixed at
    // compile time (round 6).  These heights used to fall to the one-column-per-lane kernel with a run-time height -- 64-byte row chunks, offsets in
    // VGPRs --, at 0.8 - 0.9 TB/s per pass on the 2^14 x 640 traces of BASELINE configs[2] (profiles/r05_compress64_kernel_stats.md).
#ifdef ZKHIP_AB_HOOKS
    static const bool tile_fix_off = getenv("ZKHIP_TILE_FIX") != nullptr && atoi(getenv("ZKHIP_TILE_FIX")) == 0;
#else
    constexpr bool tile_fix_off = false;
#endif
    if (!tile_fix_off && pair_ok && a.cols_per_thread != 1 && a.log_m >= 6 && a.log_m <= 9) {
        const bool contiguous = a.in_stride == 1 && a.out_stride == 1;
        const bool in_place = a.in == a.out;
        const int pol = contiguous ? 1 : (!in_place ? 2 : 0);
#define ZK_TILE_CASE(LM, BF)                                                                                                                         \
        if (a.log_m == LM) {                                                                                                                         \
            if (pol == 1) return inverse ? launch_ntt_k<4, true, 2, BF, 1>(a, s) : launch_ntt_k<4, false, 2, BF, 1>(a, s);                          \
            if (pol == 2) return inverse ? launch_ntt_k<4, true, 2, BF, 2>(a, s) : launch_ntt_k<4, false, 2, BF, 2>(a, s);                          \
            return inverse ? launch_ntt_k<4, true, 2, BF>(a, s) : launch_ntt_k<4, false, 2, BF>(a, s);                                              \
        }
        ZK_TILE_CASE(6, 1) ZK_TILE_CASE(7, 2) ZK_TILE_CASE(8, 3) ZK_TILE_CASE(9, 4)
#undef ZK_TILE_CASE
    }
    if (pair_ok && a.cols_per_thread == 2) return inverse ? launch_ntt_k<4, true, 2>(a, s) : launch_ntt_k<4, false, 2>(a, s);
    // narrow matrices use narrower tiles so that lanes are not wasted on masked columns
    if (a.ncols <= 4) return inverse ? launch_ntt_k<2, true, 1>(a, s) : launch_ntt_k<2, false, 1>(a, s);
    if (a.ncols <= 8) return inverse ? launch_ntt_k<3, true, 1>(a, s) : launch_ntt_k<3, false, 1>(a, s);
    if (a.log_m == 10) return inverse ? launch_ntt_k<4, true, 1, 5>(a, s) : launch_ntt_k<4, false, 1, 5>(a, s);
    return inverse ? launch_ntt_k<4, true, 1>(a, s) : launch_ntt_k<4, false, 1>(a, s);
}

// ------------------------------------------------------------------ native passes over contiguous vectors (column-major polynomials)
// The tile arithmetic is that of ntt_pass_kernel<4, INV, 2, 5>: 512 threads, thread (u, c) holds rows u + 32 n1 of two columns,
// 32-point DIF in registers, tile twiddle, one LDS exchange per column, 32-point DIFs again.  What differs is how a tile reaches the
// registers and leaves them (kernels.h, ColPassArgs): on a transposed side the 16 columns of one round are moved as sixteen 4 KiB lines
// between global memory and the exchange buffer (skewed line pitch, col_lidx),
// lanes along the line; the per-element inter-pass twiddle is a full 1024 x 1024 table read like the data (4 MiB, shared by every
// polynomial of the batch, so it lives in L2 / MALL after the first).
// word index of element kk of line cs in the transposed staging buffer: line pitch 1028 and one extra word per 256 elements keep
// BOTH access patterns of a wave (4 consecutive kk x 16 lines, and the bit-reversed form: 4 kk that are 256 apart x 16 lines) at the
// two-way bank conflict a 64-lane access cannot avoid (pitch 1028 without the skew: 8-way for the bit-reversed form)
ZK_D uint32_t col_lidx(uint32_t cs, uint32_t kk) { return cs * 1028u + kk + (kk >> 8); }
ZK_D uint32_t brev10(uint32_t x) { return __brev(x) >> 22; }
template <bool INV>
__global__ void __launch_bounds__(512, 4) ntt_colpass_kernel(ColPassArgs a) {
    extern __shared__ uint32_t lds[];
    constexpr int C = 16, Pn = 32, M = 1024, pitch = (Pn + 1) * C;
    uint32_t* sdata = lds;
    uint32_t* stw = lds + 32 * pitch;
    uint32_t* spre = stw + M;
    int64_t bias = (int64_t)((uint64_t)P << 32);
    asm volatile("" : "+v"(bias));
    const int tid = threadIdx.x, c = tid & 15, u = tid >> 4;
    const uint32_t cg = blockIdx.x, poly = blockIdx.y;
    const uint32_t* in = a.in + (uint64_t)poly * a.in_batch;
    uint32_t* out = a.out + (uint64_t)poly * a.out_batch;
    const bool has_pre = a.pre != nullptr;
    for (int i = tid; i < M; i += 512) { stw[i] = a.w1024[i]; if (has_pre) spre[i] = a.pre[i]; }

    uint32_t x[2][32];
    if (!a.tload) {
        const uint32_t* ib = in + cg * 32 + 2 * c;
#pragma unroll
        for (int n1 = 0; n1 < 32; n1++) {
            const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(ib + (uint64_t)(u + 32 * n1) * 1024));
            x[0][n1] = v.x; x[1][n1] = v.y;
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            __syncthreads();
            // sixteen lines of 1024 words -> staging buffer, lanes along the line (8 bytes per lane)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint32_t idx = i * 512 + tid, cs = idx >> 9, kp = idx & 511;
                const uint32_t col = cg * 32 + 2 * cs + cc;
                const uint32_t line = a.in_brev ? brev10(col) : col;
                const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(in + (uint64_t)line * 1024 + 2 * kp));
                const uint32_t li = col_lidx(cs, 2 * kp);          // 2 kp and 2 kp + 1 share their skew
                sdata[li] = v.x; sdata[li + 1] = v.y;
            }
            __syncthreads();
#pragma unroll
            for (int n1 = 0; n1 < 32; n1++) {
                const uint32_t n = u + 32 * n1;
                x[cc][n1] = sdata[col_lidx(c, a.in_brev ? brev10(n) : n)];
            }
        }
        __syncthreads();
    }
    if (has_pre) {
#pragma unroll
        for (int n1 = 0; n1 < 32; n1++) {
            const uint32_t w = spre[u + Pn * n1];
            x[0][n1] = dmul(x[0][n1], w); x[1][n1] = dmul(x[1][n1], w);
        }
    }
#pragma unroll
    for (int cc = 0; cc < 2; cc++) {
        dif_stage<INV, 0>(x[cc], bias);
        dif_stage<INV, 1>(x[cc], bias);
        dif_stage<INV, 2>(x[cc], bias);
        dif_stage<INV, 3>(x[cc], bias);
        dif_stage<INV, 4, true, true>(x[cc], bias);
#pragma unroll
        for (int r = 1; r < 32; r++) x[cc][r] = (r & 1) ? dmul_sd(x[cc][r], stw[u * rev5(r)], bias) : dmul(x[cc][r], stw[u * rev5(r)]);
        x[cc][0] = dred(x[cc][0]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int cc = 0; cc < 2; cc++) {
        __syncthreads();
        uint32_t* wp = sdata + u * C + c;
#pragma unroll
        for (int r = 0; r < 32; r++) wp[rev5(r) * pitch] = x[cc][r];
        __syncthreads();
        const uint32_t* rp = sdata + c;
#pragma unroll
        for (int rho = 0; rho < 32; rho++) x[cc][rho] = rp[u * pitch + rho * C];      // P = 32: k1 = u, t = rho
    }
    const bool has_post = a.post2d != nullptr;
#pragma unroll
    for (int cc = 0; cc < 2; cc++) {
        dif_stage<INV, 0>(x[cc], bias);
        dif_stage<INV, 1>(x[cc], bias);
        dif_stage<INV, 2>(x[cc], bias);
        dif_stage<INV, 3>(x[cc], bias);
        if (has_post) dif_stage<INV, 4, true, true>(x[cc], bias);
        else dif_stage<INV, 4>(x[cc], bias);
        __builtin_amdgcn_sched_barrier(0);
    }
    // element k = 32 rev5(rho) + u of column (cg 32 + 2 c + cc) now sits in x[cc][rho]
    if (has_post) {
        const uint32_t* pb = a.post2d + cg * 32 + 2 * c;
#pragma unroll
        for (int rho = 0; rho < 32; rho++) {
            const uint32_t k = 32 * rev5(rho) + u;
            const u32x2 w = *reinterpret_cast<const u32x2*>(pb + (uint64_t)k * 1024);
            x[0][rho] = (rho & 1) ? dmul_sd(x[0][rho], w.x, bias) : dmul(x[0][rho], w.x);
            x[1][rho] = (rho & 1) ? dmul_sd(x[1][rho], w.y, bias) : dmul(x[1][rho], w.y);
        }
    }
    if (!a.tstore) {
        uint32_t* ob = out + cg * 32 + 2 * c;
#pragma unroll
        for (int rho = 0; rho < 32; rho++) {
            const uint32_t k = 32 * rev5(rho) + u;
            u32x2 v; v.x = x[0][rho]; v.y = x[1][rho];
            __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(ob + (uint64_t)k * 1024));
        }
    } else {
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            __syncthreads();
#pragma unroll
            for (int rho = 0; rho < 32; rho++) {
                const uint32_t k = 32 * rev5(rho) + u;
                sdata[col_lidx(c, a.out_brev ? brev10(k) : k)] = x[cc][rho];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint32_t idx = i * 512 + tid, cs = idx >> 9, kp = idx & 511;
                const uint32_t col = cg * 32 + 2 * cs + cc;
                const uint32_t line = a.out_brev ? brev10(col) : col;
                const uint32_t li = col_lidx(cs, 2 * kp);
                u32x2 v; v.x = sdata[li]; v.y = sdata[li + 1];
                __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(out + (uint64_t)line * 1024 + 2 * kp));
            }
        }
    }
}
hipError_t launch_ntt_colpass(const ColPassArgs& a, bool inverse, hipStream_t s) {
    if (a.count == 0) return hipSuccess;
    const size_t lds = ntt_lds_bytes(10, 4);
    static std::atomic<bool> configured[2][MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    if (!configured[inverse][dev].load(std::memory_order_acquire)) {
        hipError_t e = inverse ? hipFuncSetAttribute((const void*)ntt_colpass_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                               : hipFuncSetAttribute((const void*)ntt_colpass_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[inverse][dev].store(true, std::memory_order_release);
    }
    const dim3 grid(32, a.count), block(512);
    if (inverse) hipLaunchKernelGGL((ntt_colpass_kernel<true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((ntt_colpass_kernel<false>), grid, block, lds, s, a);
    return hipGetLastError();
}
__global__ void post2d_table_kernel(uint32_t* out, uint32_t w, uint32_t shift, uint32_t scale) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;      // 2^20 entries
    const uint32_t k = idx >> 10, c = idx & 1023;
    out[idx] = fmul(scale, fmul(fpow(w, (uint64_t)k * c), fpow(shift, c)));
}
hipError_t launch_post2d_table(uint32_t* out, uint32_t w, uint32_t shift, uint32_t scale, hipStream_t s) {
    hipLaunchKernelGGL(post2d_table_kernel, dim3(4096), dim3(256), 0, s, out, w, shift, scale);
    return hipGetLastError();
}

// ------------------------------------------------------------------ radix-2 / radix-4 combine pass (2^21, 2^22 rows)
// Streaming: a thread owns VEC adjacent columns of one group of R rows; the twiddles of a group are row-uniform.  HBM-bound,
// 8 B/element; arithmetic is R - 1 (forward) or R (inverse) Montgomery products per element group.
template <int LOG_R, int VEC>
__device__ __forceinline__ void ntt_combine_kernel_body(const CombineArgs& a) {
    constexpr int R = 1 << LOG_R;
    typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
    const uint32_t cv = (a.ncols + VEC - 1) / VEC;                    // column vectors per row
    // blockIdx.x walks the groups in runs of blockDim.y rows, blockIdx.y / threadIdx.x the column vectors: no division per thread,
    // and a wavefront stays inside one row whenever a row has 64 vectors or more
    const uint32_t cx = blockIdx.y * blockDim.x + threadIdx.x;
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.y + threadIdx.y;
    const uint32_t c = cx * VEC;
    if (g >= a.groups || cx >= cv) return;
    uint32_t x[R][VEC];
#pragma unroll
    for (int e = 0; e < R; e++) {
        const uint32_t* row = a.in + (g * a.in_group_mul + (uint64_t)e * a.in_elem_mul) * a.in_ld + c;
        // streamed once (a 2^21 x 256 matrix is 2 GiB): non-temporal both ways
        if (VEC == 4) { const u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(row)); x[e][0] = v.x; x[e][1] = v.y; x[e][2] = v.z; x[e][VEC - 1] = v.w; }
        else x[e][0] = row[0];
    }
    uint32_t tw[R];
#pragma unroll
    for (int j = 0; j < R; j++) tw[j] = (j == 0 && !a.inverse) ? MONTY_R1 : a.tw[(uint64_t)j * a.groups + g];
    constexpr uint32_t W4 = two_adic_generator(2);                    // primitive 4th root of unity
    const uint32_t im = a.inverse ? fneg(W4) : W4;                    // w_4^(+-1)
    uint32_t y[R][VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) {
        uint32_t t[R];
#pragma unroll
        for (int e = 0; e < R; e++) t[e] = (!a.inverse && e > 0) ? fmul(x[e][v], tw[e]) : x[e][v];
        if (R == 2) {
            y[0][v] = fadd(t[0], t[1]);
            y[1][v] = fsub(t[0], t[1]);
        } else {
            const uint32_t s02 = fadd(t[0], t[2]), d02 = fsub(t[0], t[2]);
            const uint32_t s13 = fadd(t[1], t[R - 1]), d13 = fmul(fsub(t[1], t[R - 1]), im);
            y[0][v] = fadd(s02, s13);
            y[1][v] = fadd(d02, d13);
            y[2 % R][v] = fsub(s02, s13);
            y[R - 1][v] = fsub(d02, d13);
        }
        if (a.inverse) {
#pragma unroll
            for (int e = 0; e < R; e++) y[e][v] = fmul(y[e][v], tw[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < R; e++) {
        const uint32_t slot = a.bitrev_out ? (LOG_R == 2 ? (uint32_t)(((e & 1) << 1) | (e >> 1)) : (uint32_t)e) : (uint32_t)e;
        uint32_t* row = a.out + (g * a.out_group_mul + (uint64_t)slot * a.out_elem_mul) * a.out_ld + c;
        if (VEC == 4) { const u32x4_nt v = {y[e][0], y[e][1], y[e][2], y[e][VEC - 1]}; __builtin_nontemporal_store(v, reinterpret_cast<u32x4_nt*>(row)); }
        else row[0] = y[e][0];
    }
}
template <int LOG_R, int VEC>
__global__ void __launch_bounds__(256) ntt_combine_kernel(CombineArgs a) { ntt_combine_kernel_body<LOG_R, VEC>(a); }
struct ntt_combine_kernel_bargs { CombineArgs a; static ntt_combine_kernel_bargs make(CombineArgs a) { return ntt_combine_kernel_bargs{a}; } };
template <int LOG_R, int VEC>
__global__ void __launch_bounds__(256) ntt_combine_kernel_batch(const ntt_combine_kernel_bargs* __restrict__ zk_arr) { const ntt_combine_kernel_bargs& zk_b = zk_arr[blockIdx.z]; ntt_combine_kernel_body<LOG_R, VEC>(zk_b.a); }

hipError_t launch_ntt_combine(const CombineArgs& a, hipStream_t s) {
    if (a.log_r != 1 && a.log_r != 2) return hipErrorInvalidValue;
    if (a.groups == 0 || a.ncols == 0) return hipSuccess;
    const bool vec = a.ncols % 4 == 0 && a.in_ld % 4 == 0 && a.out_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
    const uint32_t cv = vec ? a.ncols / 4 : a.ncols;
    uint32_t bx = 1;
    while (bx < 256u && bx < cv) bx <<= 1;                            // 256 threads = bx column vectors x (256 / bx) groups
    const uint32_t by = 256u / bx;
    const uint64_t gy = (a.groups + by - 1) / by;
    if (gy > 0x7FFFFFFFull || (cv + bx - 1) / bx > 65535u) return hipErrorInvalidValue;
    const dim3 grid((unsigned)gy, (cv + bx - 1) / bx), block(bx, by);
    if (a.log_r == 1) {
        if (vec) ZK_LAUNCH((ntt_combine_kernel<1, 4>), (ntt_combine_kernel_batch<1, 4>), ntt_combine_kernel_bargs, grid, block, 0, s, a);
        else ZK_LAUNCH((ntt_combine_kernel<1, 1>), (ntt_combine_kernel_batch<1, 1>), ntt_combine_kernel_bargs, grid, block, 0, s, a);
    } else {
        if (vec) ZK_LAUNCH((ntt_combine_kernel<2, 4>), (ntt_combine_kernel_batch<2, 4>), ntt_combine_kernel_bargs, grid, block, 0, s, a);
        else ZK_LAUNCH((ntt_combine_kernel<2, 1>), (ntt_combine_kernel_batch<2, 1>), ntt_combine_kernel_bargs, grid, block, 0, s, a);
    }
    return hipGetLastError();
}
__global__ void combine_table_kernel(uint32_t* out, uint32_t rows, uint64_t groups, uint32_t w, uint32_t shift, uint32_t scale, int bits) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint64_t)rows * groups) return;
    const uint32_t j = (uint32_t)(idx / groups);
    const uint64_t g = idx % groups;
    const uint64_t e = bits > 0 ? (uint64_t)(__brev((uint32_t)g) >> (32 - bits)) : g;
    out[idx] = fmul(scale, fmul(fpow(shift, j), fpow(w, (uint64_t)j * e)));
}
hipError_t launch_combine_table(uint32_t* out, uint32_t rows, uint64_t groups, uint32_t w, uint32_t shift, uint32_t scale, int bits, hipStream_t s) {
    const uint64_t n = (uint64_t)rows * groups;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(combine_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, rows, groups, w, shift, scale, bits);
    return hipGetLastError();
}

// ------------------------------------------------------------------ table generators
__global__ void pow_table_kernel(uint32_t* out, size_t n, uint32_t base, uint32_t scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fmul(scale, fpow(base, i));
}
hipError_t launch_pow_table(uint32_t* out, size_t n, uint32_t base, uint32_t scale, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pow_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n, base, scale);
    return hipGetLastError();
}

__global__ void post_table_kernel(uint32_t* out, uint32_t rows, uint32_t cols, uint32_t omega,
                                  uint32_t shift, uint32_t scale) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * cols) return;
    uint32_t i = (uint32_t)(idx / cols), k = (uint32_t)(idx % cols);
    uint32_t v = fmul(fpow(omega, (uint64_t)i * k), fpow(shift, i));
    out[idx] = fmul(v, scale);
}
hipError_t launch_post_table(uint32_t* out, uint32_t rows, uint32_t cols, uint32_t omega,
                             uint32_t shift, uint32_t scale, hipStream_t s) {
    size_t n = (size_t)rows * cols;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(post_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, rows, cols, omega, shift, scale);
    return hipGetLastError();
}

}  // namespace zk
