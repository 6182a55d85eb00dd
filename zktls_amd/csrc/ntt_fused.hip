// ntt_fused.hip -- the middle of a two-pass coset LDE as ONE launch: second inverse pass + first forward pass of every coset.
//
// Replaces, with ntt.hip, p3-dft 0.2.1-succinct Radix2DitParallel::coset_lde_batch (reference Cargo.lock:3903) on the path below
// crates/guest-prover-sp1/src/sp1.rs:116.
//
// Why.  A 2^20-row LDE factors as N = 1024 x 1024.  The second inverse pass (I2) and the first forward
// pass of a coset (F1) both work on the SAME tile -- 1024 elements that differ in the second index -- and a thread that finishes
// I2 holds exactly the rows { u + 32 j } that F1's first 32-point network starts from.  So the coefficients never have to reach
// memory: one workgroup reads a tile once, finishes the inverse transform in registers, keeps the (lazily reduced) coefficients
// in 64 VGPRs, and for every coset multiplies them by the coset's powers and runs the forward tile transform into a second set of
// 64 VGPRs.  Traffic for blowup 2: read 4 B + write 8 B per trace cell instead of 24 B over three launches; the launch is then
// bound by its butterflies (three tile transforms per 12 B), not by HBM.
//
// Shape: 1024 threads (thread (u, c) owns rows u + 32 n1 of one column; a wave touches 2 rows x 128 B per access), ONE workgroup
// per CU (the whole LDS: one 132 KiB exchange buffer + the twiddle tables), 16 waves at <= 128 VGPRs.  Because no second
// workgroup is resident to hide memory latency, the kernel is persistent: a workgroup walks a list of tiles, and the 32 loads of
// its NEXT tile are issued as soon as the last coset has consumed the coefficients -- a whole forward transform ahead of their
// use.  Stores are fire-and-forget.  The strided sides of the LDE sit here (rows 1024 apart on the way in, bit-reversed rows on
// the way out), where there is slack, so that the memory-bound passes on either side stream contiguous blocks.
// No MFMA: a 31-bit modular butterfly is not a dense contraction.
#include <atomic>
#include <type_traits>

#include "babybear.cuh"
#include "kernels.h"
#include "ntt_bfly.cuh"

namespace zk {

namespace {

constexpr int FUSED_MAX_DEVICES = 64;
constexpr int TC = 32;                       // tile columns = lanes along a row chunk (128 B)
constexpr int PITCH = 33 * TC;               // exchange row pitch in words: conflict-free both ways
constexpr int XCHG_WORDS = 32 * PITCH;
constexpr size_t FUSED_LDS_BYTES = (size_t)XCHG_WORDS * 4 + (size_t)(2 + 4) * 1024 * 4;

// LDS-only workgroup barrier: outstanding global loads (the prefetched tile) and stores stay in flight across it
ZK_D void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16 table words of thread u (half H of its row of 32, in register order) as four 16-byte LDS loads; all lanes of a half-wave read
// the same words.  Halves, with a scheduling barrier between them, keep the table at 16 live registers instead of 32.
template <int H>
ZK_D void load_half16(uint32_t (&t)[16], const uint32_t* table, int u) {
    const uint4* tp = reinterpret_cast<const uint4*>(table + 32 * u + 16 * H);
#pragma unroll
    for (int i = 0; i < 4; i++) { const uint4 v = tp[i]; t[4 * i] = v.x; t[4 * i + 1] = v.y; t[4 * i + 2] = v.z; t[4 * i + 3] = v.w; }
}
// first five stages over n1 and the tile twiddle w_1024^(+-u k1) (stw in thread order: stw[32 u + r] = w^(u rev5(r)));
// x[r] then holds A[k1 = rev5(r)]
template <bool INV>
ZK_D void tile_phase_a(uint32_t (&x)[32], const uint32_t* stw, int u, int64_t bias) {
    dif_stage<INV, 0>(x, bias);
    dif_stage<INV, 1>(x, bias);
    dif_stage<INV, 2>(x, bias);
    dif_stage<INV, 3>(x, bias);
    dif_stage<INV, 4, true, true>(x, bias);
    __builtin_amdgcn_sched_barrier(0);       // the table words are fetched here, not above the butterflies (32 live registers)
    {
        uint32_t tw[16];
        load_half16<0>(tw, stw, u);
#pragma unroll
        for (int r = 1; r < 16; r++) x[r] = (r & 1) ? dmul_sd(x[r], tw[r], bias) : dmul(x[r], tw[r]);
        x[0] = dred(x[0]);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        uint32_t tw[16];
        load_half16<1>(tw, stw, u);
#pragma unroll
        for (int r = 16; r < 32; r++) x[r] = (r & 1) ? dmul_sd(x[r], tw[r - 16], bias) : dmul(x[r], tw[r - 16]);
    }
}
// last five stages; outputs lazy (sums in [0, 2P), odd slots signed differences): a multiplication follows.
// Element k = 32 rev5(rho) + u of the transform ends in x[rho].
template <bool INV>
ZK_D void tile_phase_b(uint32_t (&x)[32], int64_t bias) {
    dif_stage<INV, 0>(x, bias);
    dif_stage<INV, 1>(x, bias);
    dif_stage<INV, 2>(x, bias);
    dif_stage<INV, 3>(x, bias);
    dif_stage<INV, 4, true, true>(x, bias);
}
ZK_D void exchange(uint32_t (&x)[32], uint32_t* sdata, int u, int c) {
    uint32_t* wp = sdata + u * TC + c;
#pragma unroll
    for (int r = 0; r < 32; r++) wp[rev5(r) * PITCH] = x[r];
    lds_barrier();
    const uint32_t* rp = sdata + u * PITCH + c;
#pragma unroll
    for (int rho = 0; rho < 32; rho++) x[rho] = rp[rho * TC];
}

struct TilePos { uint32_t tile, cg; };
// item j of an XCD's list -> (tile, column group): the column groups of a tile are consecutive items (they share rows, hence L2
// lines and DRAM pages), tiles are dealt round-robin over the XCDs, and the tile index may be rotated so that the tiles in
// flight at one time are spread over the row period instead of being neighbours
ZK_D TilePos item_pos(uint32_t j, uint32_t xcd, uint32_t ncg, uint32_t log_tiles, uint32_t rot) {
    TilePos p;
    p.cg = j % ncg;
    uint32_t tile = (j / ncg) * 8u + xcd;
    if (rot && log_tiles > rot) tile = ((tile << rot) | (tile >> (log_tiles - rot))) & ((1u << log_tiles) - 1u);
    p.tile = tile;
    return p;
}

// 1024 threads: thread (u, c) owns rows u + 32 n1 of ONE column, so a wave covers 2 rows x 128 B per access and a CU holds 16
// waves (4 per SIMD) at <= 128 VGPRs.  One wave alone issues a vector instruction every 4 cycles, a SIMD can take one every 2:
// with only two waves per SIMD (the 512-thread, two-columns-per-lane form of this kernel: 1.32 ms) every stall of one wave
// halves the SIMD's rate.
}  // namespace

template <int TAG>
__global__ void __launch_bounds__(1024, 4) lde_fused_kernel(LdeFusedArgs a, uint32_t items_per_xcd, uint32_t wgs_per_xcd, uint32_t log_tiles) {
    extern __shared__ uint32_t lds[];
    uint32_t* sdata = lds;
    uint32_t* stw_inv = lds + XCHG_WORDS;
    uint32_t* stw_fwd = stw_inv + 1024;
    uint32_t* slots = stw_fwd + 1024;            // [2][pre 1024 | post 1024]
    int64_t bias = (int64_t)((uint64_t)P << 32);
    asm volatile("" : "+v"(bias));
    const int tid = threadIdx.x;
    const int c = tid & (TC - 1);
    const int u = tid >> 5;
    const uint32_t xcd = blockIdx.x & 7u, l = blockIdx.x >> 3;
    const uint32_t ncg = a.ncols / 32u;
    constexpr uint32_t B = 2;
    const uint32_t istep_b = (uint32_t)(4u * 32u * a.in_stride * a.in_ld);
    const uint32_t in_off = 4u * (uint32_t)((uint64_t)u * a.in_stride * a.in_ld) + 4u * (uint32_t)c;
    const uint32_t ostep_b = (uint32_t)(4u * a.out_stride * a.out_ld);
    const uint32_t out_off = 32u * (__brev((uint32_t)u) >> 27) * ostep_b + 4u * (uint32_t)c;

    uint32_t j = l;
    if (j >= items_per_xcd) return;
    uint32_t nx[32];                             // the tile as it arrives; carried across the loop (prefetched)
    TilePos cur = item_pos(j, xcd, ncg, log_tiles, a.map_rot);
    {
        const uint32_t* ib = a.in + (uint64_t)cur.tile * a.in_tile_mul * a.in_ld + cur.cg * 32u;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(ib), 0, 0xFFFFFFFFu, 0x00020000);
#pragma unroll
        for (int n1 = 0; n1 < 32; n1++) nx[n1] = __builtin_amdgcn_raw_buffer_load_b32(rs, in_off, n1 * istep_b, 2);
    }
    stw_inv[tid] = a.w1024_inv[tid];
    stw_fwd[tid] = a.w1024_fwd[tid];
    // tables of the first coset of the first tile
    slots[tid] = a.pre[0][tid];
    slots[1024 + tid] = a.post[0][(uint64_t)cur.tile * 1024u + tid];
    // the first tile is waited for HERE: if its loads were still pending at the loop header, the compiler would merge "32 loads
    // pending, nothing behind them" with the loop's "32 loads pending, 32 stores behind them" into counted waits that also
    // drain the previous tile's stores at the top of every iteration
#pragma unroll
    for (int n1 = 0; n1 < 32; n1++) asm volatile("" : "+v"(nx[n1]));
    lds_barrier();
    uint32_t q = 0;                              // coset iterations done: tables of iteration q live in slot q & 1

    for (; j < items_per_xcd; j += wgs_per_xcd) {
        const uint32_t jn = j + wgs_per_xcd;
        const bool has_next = jn < items_per_xcd;
        const TilePos nxt = item_pos(has_next ? jn : j, xcd, ncg, log_tiles, a.map_rot);

        // ---- second inverse pass; no table: the first pass carried w^-(i1 k2) / N
        uint32_t s[32];                          // the tile, then its coefficients (lazy)
#pragma unroll
        for (int n1 = 0; n1 < 32; n1++) s[n1] = nx[n1];
        tile_phase_a<true>(s, stw_inv, u, bias);
        exchange(s, sdata, u, c);
        tile_phase_b<true>(s, bias);
        lds_barrier();                           // exchange buffer free again
        // coefficient c[k], k = 32 rev5(rho) + u, now sits (lazy) in s[rho]

        auto coset = [&](uint32_t t, auto prefetch_tag) {
            constexpr bool PREFETCH = decltype(prefetch_tag)::value;
            const uint32_t* slot = slots + (q & 1u) * 2048u;
            // tables of the NEXT coset iteration: loads now, into the other slot after this iteration's first barrier
            const bool last = t + 1 == B;
            const bool stage_next = !last || has_next;
            uint32_t g0, g1;
            {
                const uint32_t tn = last ? 0u : t + 1u;
                const uint32_t* pp = a.pre[tn];
                const uint32_t* po = a.post[tn] + (uint64_t)(last ? nxt.tile : cur.tile) * 1024u;
                const auto prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(pp), 0, (uint32_t)__builtin_amdgcn_readfirstlane(stage_next ? 4096 : 0), 0x00020000);
                const auto pos = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(po), 0, (uint32_t)__builtin_amdgcn_readfirstlane(stage_next ? 4096 : 0), 0x00020000);
                g0 = __builtin_amdgcn_raw_buffer_load_b32(prs, 4u * tid, 0, 0);
                g1 = __builtin_amdgcn_raw_buffer_load_b32(pos, 4u * tid, 0, 0);
            }
            // forward input n = u + 32 n1 is coefficient k = n: register rev5(n1); times the coset's power pre_t[n]
            uint32_t w[32];
            {
                uint32_t pw[16];                 // pre_t[u + 32 n1] at slot[32 u + n1]
                load_half16<0>(pw, slot, u);
#pragma unroll
                for (int n1 = 0; n1 < 16; n1++) {
                    const int r = rev5(n1);
                    w[n1] = (r & 1) ? dmul_sd(s[r], pw[n1], bias) : dmul(s[r], pw[n1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                uint32_t pw[16];
                load_half16<1>(pw, slot, u);
#pragma unroll
                for (int n1 = 16; n1 < 32; n1++) {
                    const int r = rev5(n1);
                    w[n1] = (r & 1) ? dmul_sd(s[r], pw[n1 - 16], bias) : dmul(s[r], pw[n1 - 16]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // the loads below must not move up beside the coefficients they replace
            if (PREFETCH) {
                // the coefficients are consumed: the next tile's loads fly under this transform.  Unconditional (a zero-record
                // descriptor turns them into no-ops behind the last tile), so that the old tile need not stay live beside them
                const uint32_t* ib = a.in + (uint64_t)nxt.tile * a.in_tile_mul * a.in_ld + nxt.cg * 32u;
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(ib), 0, (uint32_t)__builtin_amdgcn_readfirstlane(has_next ? -1 : 0), 0x00020000);
                uint32_t istep = istep_b;        // opaque: the 32 scalar row offsets are recomputed here (scalar multiplies) instead of
                asm volatile("" : "+s"(istep));  // living in 32 SGPRs across the loop -- they were being spilled to vector lanes
#pragma unroll
                for (int n1 = 0; n1 < 32; n1++) nx[n1] = __builtin_amdgcn_raw_buffer_load_b32(rs, in_off, n1 * istep, 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            tile_phase_a<false>(w, stw_fwd, u, bias);
            exchange(w, sdata, u, c);
            if (stage_next) {                    // every wave is past this iteration's first barrier: the other slot is idle
                uint32_t* ns = slots + ((q + 1u) & 1u) * 2048u;
                ns[tid] = g0; ns[1024 + tid] = g1;
            }
            tile_phase_b<false>(w, bias);
            // post = inter-pass twiddle x coset power, then 1024 apart bit-reversed rows of the coset's block
            uint32_t* ob = a.out[t] + (uint64_t)cur.tile * a.out_tile_mul * a.out_ld + cur.cg * 32u;
            const auto ors = __builtin_amdgcn_make_buffer_rsrc(ob, 0, 0xFFFFFFFFu, 0x00020000);
            uint32_t ostep = ostep_b;
            asm volatile("" : "+s"(ostep));
            __builtin_amdgcn_sched_barrier(0);
            uint32_t qw[16];                     // post_t[tile][32 rev5(rho) + u] at slot[1024 + 32 u + rho]
            load_half16<0>(qw, slot + 1024, u);
#pragma unroll
            for (int rho = 0; rho < 16; rho++) {
                const uint32_t v = (rho & 1) ? dmul_sd(w[rho], qw[rho], bias) : dmul(w[rho], qw[rho]);
                __builtin_amdgcn_raw_buffer_store_b32(v, ors, out_off, rho * ostep, 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            load_half16<1>(qw, slot + 1024, u);
#pragma unroll
            for (int rho = 16; rho < 32; rho++) {
                const uint32_t v = (rho & 1) ? dmul_sd(w[rho], qw[rho - 16], bias) : dmul(w[rho], qw[rho - 16]);
                __builtin_amdgcn_raw_buffer_store_b32(v, ors, out_off, rho * ostep, 2);
            }
            lds_barrier();                       // exchange buffer free, the other slot's tables visible
            q++;
        };
        // exactly two cosets per launch, spelled out: with a run-time loop here the compiler keeps the prefetch registers live
        // across it and spills (a blowup of 4 is two launches)
        coset(0u, std::false_type{});
        coset(1u, std::true_type{});
        cur = nxt;
    }
}

namespace {

__global__ void fused_table_kernel(uint32_t* out, const uint32_t* in, int mode) {
    const uint32_t b = blockIdx.x, p = threadIdx.x;          // 1024 threads: one block of 1024 words
    const uint32_t u = p >> 5, r = p & 31u;
    const uint32_t src = mode == 0 ? u * (uint32_t)rev5((int)r) : mode == 1 ? u + 32u * r : 32u * (uint32_t)rev5((int)r) + u;
    out[(uint64_t)b * 1024u + p] = in[(uint64_t)b * 1024u + src];
}

int cu_count_of(int dev) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
}

template <int TAG>
hipError_t launch_tagged(const LdeFusedArgs& a, uint32_t grid, uint32_t items_per_xcd, uint32_t log_tiles, hipStream_t s) {
    static std::atomic<bool> configured[FUSED_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FUSED_MAX_DEVICES) return hipErrorInvalidDevice;
    if (!configured[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)lde_fused_kernel<TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS_BYTES);
        if (e != hipSuccess) return e;
        configured[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((lde_fused_kernel<TAG>), dim3(grid), dim3(1024), FUSED_LDS_BYTES, s, a, items_per_xcd, grid / 8u, log_tiles);
    return hipGetLastError();
}

}  // namespace

bool lde_fused_supported(const LdeFusedArgs& a) {
    if (a.ncols == 0 || a.ncols % 32u != 0 || false) return false;
    if (a.num_tiles < 8 || (a.num_tiles & (a.num_tiles - 1u)) != 0) return false;
    for (uint32_t t = 0; t < (uint32_t)FUSED_COSETS; t++)
        if (!a.out[t] || !a.pre[t] || !a.post[t]) return false;
    // buffer addressing: every byte offset inside one tile must fit 32 bits
    const uint64_t in_span = 4ull * (1023ull * a.in_stride * a.in_ld + a.ncols);
    const uint64_t out_span = 4ull * (1023ull * a.out_stride * a.out_ld + a.ncols);
    return in_span < (1ull << 32) && out_span < (1ull << 32);
}

hipError_t launch_fused_table(uint32_t* out, const uint32_t* in, uint32_t blocks, int mode, hipStream_t s) {
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(fused_table_kernel, dim3(blocks), dim3(1024), 0, s, out, in, mode);
    return hipGetLastError();
}

hipError_t launch_lde_fused(const LdeFusedArgs& a, hipStream_t s) {
    if (!lde_fused_supported(a)) return hipErrorInvalidValue;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    const uint32_t ncg = a.ncols / 32u;
    const uint32_t items = a.num_tiles * ncg;
    const uint32_t items_per_xcd = items / 8u;
    uint32_t grid = a.grid ? a.grid : (uint32_t)cu_count_of(dev);
    grid = (grid / 8u) * 8u;
    if (grid < 8u) grid = 8u;
    if (grid > items) grid = items;
    const uint32_t log_tiles = 31u - (uint32_t)__builtin_clz(a.num_tiles);
    return a.bench_tag ? launch_tagged<1>(a, grid, items_per_xcd, log_tiles, s) : launch_tagged<0>(a, grid, items_per_xcd, log_tiles, s);
}

}  // namespace zk
