// hal.hip -- the RISC Zero `Hal` operator set at operator level, hand-written for gfx950:
// element-wise operators, zk_shift, mix_poly_coeffs, batch_evaluate_any, gather_sample, scatter, prefix_products and the
// SHA-256 variant of hash_rows / hash_fold.  Kernels and their C-ABI entries (include/zkhip.h) live together here.
//
// Replaces the CUDA / C++ kernels of risc0-sys 1.2.5 behind risc0-zkp 1.2.5 `hal::Hal` (reference Cargo.lock:5045, 5057),
// which the reference reaches through crates/guest-prover-r0/src/prover.rs:90.  Layout as RISC Zero's Hal holds it:
// polynomials / columns are contiguous vectors (column-major [count][size]), base elements in Montgomery form, extension
// elements 4 consecutive words.  The extension field is a TEMPLATE PARAMETER: x^4 = 11 (Plonky3 / SP1) or x^4 = -11
// (RISC Zero's x^4 + 11).  All of these are HBM-bound streaming or gather kernels (no MFMA, nothing to tile).  The streaming ones
// move 16 bytes per lane and instruction where alignment allows (add, zeroize, zk_shift, sum), mix_poly_coeffs keeps its
// accumulators in registers and reads the (wave-uniform) mix powers from a small table, batch_evaluate_any reads its coefficients
// coalesced (lane t takes coefficients t mod 1024).  Measured per operator at RISC Zero's sizes: profiles/r05_hal_ops.md.
#include "context.h"

namespace zk {

// ---- extension arithmetic with the non-residue as a template parameter (W in Montgomery form)
constexpr uint32_t MONTY_W_SP1 = to_monty(EXT_W);          // x^4 = 11
constexpr uint32_t MONTY_W_R0 = to_monty(P - EXT_W);       // x^4 = -11
template <uint32_t W>
ZK_D Ext ext_mul_t(const Ext& a, const Ext& b) {
    const uint32_t w1 = dmul(b.c[1], W), w2 = dmul(b.c[2], W), w3 = dmul(b.c[3], W);
    uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    dacc2(s0, a.c[0], b.c[0], a.c[1], w3); dacc2(s0, a.c[2], w2, a.c[3], w1);
    dacc2(s1, a.c[0], b.c[1], a.c[1], b.c[0]); dacc2(s1, a.c[2], w3, a.c[3], w2);
    dacc2(s2, a.c[0], b.c[2], a.c[1], b.c[1]); dacc2(s2, a.c[2], b.c[0], a.c[3], w3);
    dacc2(s3, a.c[0], b.c[3], a.c[1], b.c[2]); dacc2(s3, a.c[2], b.c[1], a.c[3], b.c[0]);
    return Ext{{dacc_finish(s0), dacc_finish(s1), dacc_finish(s2), dacc_finish(s3)}};
}
ZK_D Ext ext_add_d(const Ext& a, const Ext& b) { return Ext{{dadd(a.c[0], b.c[0]), dadd(a.c[1], b.c[1]), dadd(a.c[2], b.c[2]), dadd(a.c[3], b.c[3])}}; }
ZK_D Ext ld_ext(const uint32_t* p) { const uint4 v = *reinterpret_cast<const uint4*>(p); return Ext{{v.x, v.y, v.z, v.w}}; }
ZK_D void st_ext(uint32_t* p, const Ext& e) { *reinterpret_cast<uint4*>(p) = make_uint4(e.c[0], e.c[1], e.c[2], e.c[3]); }

typedef uint32_t hal_u32x4 __attribute__((ext_vector_type(4)));
// ---- element-wise (grid-stride; 16 B per lane when the vectors are 16-byte aligned, the tail and unaligned vectors 4 B per lane)
__global__ void hal_add_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint64_t n, int vec) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    uint64_t done = 0;
    if (vec) {
        const uint64_t n4 = n / 4;
        const hal_u32x4* a4 = reinterpret_cast<const hal_u32x4*>(a);
        const hal_u32x4* b4 = reinterpret_cast<const hal_u32x4*>(b);
        hal_u32x4* o4 = reinterpret_cast<hal_u32x4*>(out);
        for (uint64_t i = tid; i < n4; i += nth) {
            const hal_u32x4 x = __builtin_nontemporal_load(a4 + i), y = __builtin_nontemporal_load(b4 + i);
            hal_u32x4 r;
            r.x = dadd(x.x, y.x); r.y = dadd(x.y, y.y); r.z = dadd(x.z, y.z); r.w = dadd(x.w, y.w);
            __builtin_nontemporal_store(r, o4 + i);
        }
        done = n4 * 4;
    }
    for (uint64_t i = done + tid; i < n; i += nth) out[i] = dadd(a[i], b[i]);
}
__global__ void hal_zeroize_kernel(uint32_t* __restrict__ io, uint64_t n, int vec) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    uint64_t done = 0;
    if (vec) {
        const uint64_t n4 = n / 4;
        uint4* p = reinterpret_cast<uint4*>(io);
        for (uint64_t i = tid; i < n4; i += nth) {
            uint4 v = p[i];
            const bool hit = v.x == 0xFFFFFFFFu || v.y == 0xFFFFFFFFu || v.z == 0xFFFFFFFFu || v.w == 0xFFFFFFFFu;
            if (hit) {                                         // (rare: only touched lines are written back)
                if (v.x == 0xFFFFFFFFu) v.x = 0u;
                if (v.y == 0xFFFFFFFFu) v.y = 0u;
                if (v.z == 0xFFFFFFFFu) v.z = 0u;
                if (v.w == 0xFFFFFFFFu) v.w = 0u;
                p[i] = v;
            }
        }
        done = n4 * 4;
    }
    for (uint64_t i = done + tid; i < n; i += nth)
        if (io[i] == 0xFFFFFFFFu) io[i] = 0u;
}
// out[i] = sum_j in[j * count + i] (extension elements): lanes along i, the j loop walks `to_add` coalesced rows
__global__ void hal_sum_ext_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in, uint64_t count, uint64_t to_add) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Ext acc = ext_zero();
    for (uint64_t j = 0; j < to_add; j++) acc = ext_add_d(acc, ld_ext(in + 4 * (j * count + i)));
    st_ext(out + 4 * i, acc);
}
// coefficient i of every polynomial times shift^i.  blockIdx.y = the polynomial; a lane owns FOUR consecutive coefficients (one
// 16-byte load and store, the lanes of a wave 1 KiB of consecutive memory) and walks the polynomial with the grid's stride: one
// fpow at its first position, then its running power moves on by shift^(4 * threads) -- `step`, the same for every lane, with
// s1 .. s3 = shift^1 .. shift^3 computed once on the host.  Polynomials shorter than 4 coefficients: one lane each.
__global__ void hal_zk_shift_kernel(uint32_t* __restrict__ io, int log_size, uint32_t shift, uint32_t s1, uint32_t s2, uint32_t s3, uint32_t step) {
    const uint64_t n = (uint64_t)1 << log_size;
    uint32_t* v = io + (uint64_t)blockIdx.y * n;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    if (n < 4) {
        if (tid == 0) { uint32_t s = MONTY_R1; for (uint64_t k = 0; k < n; k++) { v[k] = dmul(v[k], s); s = dmul(s, shift); } }
        return;
    }
    uint64_t i = 4 * tid;
    if (i >= n) return;
    uint32_t s = fpow(shift, i);
    uint4* p = reinterpret_cast<uint4*>(v);
    for (; i < n; i += 4 * nth) {
        uint4 x = p[i / 4];
        x.x = dmul(x.x, s); x.y = dmul(x.y, dmul(s, s1)); x.z = dmul(x.z, dmul(s, s2)); x.w = dmul(x.w, dmul(s, s3));
        p[i / 4] = x;
        s = dmul(s, step);
    }
}
// the same for a vector that is NOT 16-byte aligned (a Hal slice at a 4-byte offset inside a larger buffer): one coefficient per lane and
// step, 4-byte accesses, the running power advanced by shift^(threads of the grid row)
__global__ void hal_zk_shift_scalar_kernel(uint32_t* __restrict__ io, int log_size, uint32_t shift, uint32_t step) {
    const uint64_t n = (uint64_t)1 << log_size;
    uint32_t* v = io + (uint64_t)blockIdx.y * n;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    if (tid >= n) return;
    uint32_t s = fpow(shift, tid);
    for (uint64_t i = tid; i < n; i += nth) { v[i] = dmul(v[i], s); s = dmul(s, step); }
}
// mix_poly_coeffs: out[combos[i] * count + idx] += mix_start * mix^i * in[i * count + idx].
// (1) the powers mix_start * mix^i are the same for every idx: one small launch writes them to a table (a wave's scan: lane l takes
//     i = l, l + 64, ...; its start mix^l by square-and-multiply, its stride mix^64);
// (2) a lane owns one idx and walks the inputs (reads coalesced along idx, 4 B per lane, the power of the step from the table through
//     scalar loads -- the index is wave-uniform).  Up to MIX_REGS combos accumulate in REGISTERS (the combo number is wave-uniform: a
//     scalar branch picks the accumulator) and reach the output once, at the end; combos beyond that are added in memory per term.
constexpr int MIX_REGS = 8;
template <uint32_t W>
__global__ void __launch_bounds__(64) hal_mix_powers_kernel(uint32_t* __restrict__ table, Ext mix_start, Ext mix, uint64_t input_size) {
    const uint32_t l = threadIdx.x;
    Ext pw = ext_one(), b = mix;
    for (uint32_t k = l; k; k >>= 1) { if (k & 1) pw = ext_mul_t<W>(pw, b); b = ext_mul_t<W>(b, b); }
    Ext stride = mix;                                          // mix^64
    for (int k = 0; k < 6; k++) stride = ext_mul_t<W>(stride, stride);
    Ext cur = ext_mul_t<W>(mix_start, pw);
    for (uint64_t i = l; i < input_size; i += 64) { st_ext(table + 4 * i, cur); cur = ext_mul_t<W>(cur, stride); }
}
__global__ void __launch_bounds__(256) hal_mix_poly_coeffs_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ powers, const uint32_t* __restrict__ in,
                                                                  const uint32_t* __restrict__ combos, uint64_t input_size, uint64_t count) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    Ext acc[MIX_REGS];
#pragma unroll
    for (int c = 0; c < MIX_REGS; c++) acc[c] = ext_zero();
    uint32_t touched = 0;
    constexpr int AHEAD = 8;                                    // input words in flight per lane: the loop is latency-bound otherwise
    uint32_t vin[AHEAD];
    for (uint64_t i0 = 0; i0 < input_size; i0 += AHEAD) {
#pragma unroll
      for (int u = 0; u < AHEAD; u++) vin[u] = i0 + u < input_size ? __builtin_nontemporal_load(in + (i0 + u) * count + idx) : 0u;
#pragma unroll
      for (int u = 0; u < AHEAD; u++) {
        const uint64_t i = i0 + u;
        if (i >= input_size) break;
        const uint32_t combo = __builtin_amdgcn_readfirstlane(combos[i]);
        const uint4 pv = *reinterpret_cast<const uint4*>(powers + 4 * i);          // uniform address: scalar loads
        const Ext cur{{(uint32_t)__builtin_amdgcn_readfirstlane(pv.x), (uint32_t)__builtin_amdgcn_readfirstlane(pv.y),
                       (uint32_t)__builtin_amdgcn_readfirstlane(pv.z), (uint32_t)__builtin_amdgcn_readfirstlane(pv.w)}};
        const Ext term = ext_mul_base_dev(cur, vin[u]);
        if (combo < (uint32_t)MIX_REGS) {
            touched |= 1u << combo;
            switch (combo) {                                   // wave-uniform: one scalar branch, statically indexed registers
                case 0: acc[0] = ext_add_d(acc[0], term); break;
                case 1: acc[1] = ext_add_d(acc[1], term); break;
                case 2: acc[2] = ext_add_d(acc[2], term); break;
                case 3: acc[3] = ext_add_d(acc[3], term); break;
                case 4: acc[4] = ext_add_d(acc[4], term); break;
                case 5: acc[5] = ext_add_d(acc[5], term); break;
                case 6: acc[6] = ext_add_d(acc[6], term); break;
                default: acc[7] = ext_add_d(acc[7], term); break;
            }
        } else {
            uint32_t* o = out + 4 * ((uint64_t)combo * count + idx);
            st_ext(o, ext_add_d(ld_ext(o), term));
        }
      }
    }
#pragma unroll
    for (int c = 0; c < MIX_REGS; c++)
        if (touched & (1u << c)) { uint32_t* o = out + 4 * ((uint64_t)c * count + idx); st_ext(o, ext_add_d(ld_ext(o), acc[c])); }
}
// ---- the same operator with the terms GROUPED BY COMBO (round 5).  A plan kernel (one workgroup) ranks the inputs by (combo, index) and
// writes, in that order, the input's row number, its mix power and its combo, plus for the first term of every combo where the combo's
// run ends.  The streaming kernel then walks whole runs: ONE set of four 64-bit running sums per run (dacc2: two multiply-adds and one
// conditional subtraction per pair of terms and component, no Montgomery reduction until the run ends), eight input words per lane in
// flight, the powers and row numbers wave-uniform (scalar loads), one read-modify-write of the output per (combo, idx) -- for ANY number
// of combos (the register form above spills to per-term read-modify-writes beyond eight).  The sum is exact field arithmetic, so the
// order of the terms does not show in the result.
constexpr uint32_t MIX_PLAN_MAX = 4096;                          // inputs the plan kernel ranks in LDS; longer lists keep the form above
__global__ void __launch_bounds__(256) hal_mix_plan_kernel(const uint32_t* __restrict__ combos, const uint32_t* __restrict__ powers, uint32_t n,
                                                           uint32_t* __restrict__ spow, uint32_t* __restrict__ srow, uint32_t* __restrict__ scombo, uint32_t* __restrict__ send) {
    __shared__ uint32_t cb[MIX_PLAN_MAX], sc[MIX_PLAN_MAX];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) cb[i] = combos[i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t c = cb[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; j++) rank += (cb[j] < c || (cb[j] == c && j < i)) ? 1u : 0u;
        srow[rank] = i; scombo[rank] = c; sc[rank] = c;
        st_ext(spow + 4 * (uint64_t)rank, ld_ext(powers + 4 * (uint64_t)i));
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < n; r += blockDim.x) {
        uint32_t e = 0;
        if (r == 0 || sc[r - 1] != sc[r]) { e = r + 1; while (e < n && sc[e] == sc[r]) e++; }
        send[r] = e;                                           // end of the run that starts here (0: not a start)
    }
}
__global__ void __launch_bounds__(256) hal_mix_sorted_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in, const uint32_t* __restrict__ spow,
                                                             const uint32_t* __restrict__ srow, const uint32_t* __restrict__ scombo, const uint32_t* __restrict__ send,
                                                             uint32_t n, uint64_t count) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const uint32_t* col = in + idx;
    uint32_t r = 0;
    while (r < n) {
        const uint32_t combo = __builtin_amdgcn_readfirstlane(scombo[r]), end = __builtin_amdgcn_readfirstlane(send[r]);
        uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        constexpr int AHEAD = 8;
        for (; r + AHEAD <= end; r += AHEAD) {
            uint32_t v[AHEAD];
#pragma unroll
            for (int u = 0; u < AHEAD; u++) v[u] = __builtin_nontemporal_load(col + (uint64_t)__builtin_amdgcn_readfirstlane(srow[r + u]) * count);
#pragma unroll
            for (int u = 0; u < AHEAD; u += 2) {
                const uint4 pa = *reinterpret_cast<const uint4*>(spow + 4 * (uint64_t)(r + u)), pb = *reinterpret_cast<const uint4*>(spow + 4 * (uint64_t)(r + u + 1));
                const uint32_t a0 = __builtin_amdgcn_readfirstlane(pa.x), a1 = __builtin_amdgcn_readfirstlane(pa.y), a2 = __builtin_amdgcn_readfirstlane(pa.z), a3 = __builtin_amdgcn_readfirstlane(pa.w);
                const uint32_t b0 = __builtin_amdgcn_readfirstlane(pb.x), b1 = __builtin_amdgcn_readfirstlane(pb.y), b2 = __builtin_amdgcn_readfirstlane(pb.z), b3 = __builtin_amdgcn_readfirstlane(pb.w);
                dacc2(s0, a0, v[u], b0, v[u + 1]); dacc2(s1, a1, v[u], b1, v[u + 1]);
                dacc2(s2, a2, v[u], b2, v[u + 1]); dacc2(s3, a3, v[u], b3, v[u + 1]);
            }
        }
        for (; r < end; r++) {
            const uint32_t v = __builtin_nontemporal_load(col + (uint64_t)__builtin_amdgcn_readfirstlane(srow[r]) * count);
            const uint4 pa = *reinterpret_cast<const uint4*>(spow + 4 * (uint64_t)r);
            dacc2(s0, (uint32_t)__builtin_amdgcn_readfirstlane(pa.x), v, 0u, 0u); dacc2(s1, (uint32_t)__builtin_amdgcn_readfirstlane(pa.y), v, 0u, 0u);
            dacc2(s2, (uint32_t)__builtin_amdgcn_readfirstlane(pa.z), v, 0u, 0u); dacc2(s3, (uint32_t)__builtin_amdgcn_readfirstlane(pa.w), v, 0u, 0u);
        }
        uint32_t* o = out + 4 * ((uint64_t)combo * count + idx);
        st_ext(o, ext_add_d(ld_ext(o), Ext{{dacc_finish(s0), dacc_finish(s1), dacc_finish(s2), dacc_finish(s3)}}));
    }
}
// out[e] = polynomial which[e] at xs[e].  One workgroup of 1024 lanes per evaluation; lane t takes the coefficients i = t mod 1024 (a wave
// reads 256 consecutive bytes per step) and evaluates sum_k c[t + 1024 k] y^k with y = x^1024, which is WORKGROUP-UNIFORM: the powers
// y^0 .. y^31 come from a small table (a tables kernel writes it per evaluation; scalar loads here), so a coefficient costs FOUR
// multiply-adds into 64-bit running sums (base x extension, dacc2: no Montgomery reduction until a block of 32 coefficients ends) and
// one extension product by y^32 per block -- 5 products per 4 bytes instead of the 19 of a Horner step in the extension field (round 4),
// which made the operator arithmetic-bound at 0.22 of the HBM peak.  The partial value is scaled by x^t = x^(t mod 64) (x^64)^(t div 64)
// (two more table rows) and the lanes are summed in LDS.
constexpr uint32_t EVAL_T = 1024, EVAL_BLK = 32;
constexpr uint32_t EVAL_TAB = 64 + 16 + EVAL_BLK + 1;             // per evaluation: x^l (l < 64), x^(64 w) (w < 16), y^j (j <= 32), extension elements
template <uint32_t W>
__global__ void __launch_bounds__(64) hal_eval_tables_kernel(const uint32_t* __restrict__ xs, uint32_t* __restrict__ tab) {
    const uint32_t e = blockIdx.x, l = threadIdx.x;
    Ext sq[16];                                                // x^(2^k)
    sq[0] = ld_ext(xs + 4 * (uint64_t)e);
#pragma unroll
    for (int k = 1; k < 16; k++) sq[k] = ext_mul_t<W>(sq[k - 1], sq[k - 1]);
    uint32_t* t = tab + 4 * (uint64_t)EVAL_TAB * e;
    Ext a = ext_one(), b = ext_one(), c = ext_one();
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if ((l >> k) & 1u) { a = ext_mul_t<W>(a, sq[k]); c = ext_mul_t<W>(c, sq[10 + k]); }
        if (k < 4 && ((l >> k) & 1u)) b = ext_mul_t<W>(b, sq[6 + k]);
    }
    st_ext(t + 4 * l, a);
    if (l < 16) st_ext(t + 4 * (64 + l), b);
    if (l < EVAL_BLK) st_ext(t + 4 * (80 + l), c);
    if (l == 0) st_ext(t + 4 * (80 + EVAL_BLK), sq[15]);       // y^32 = x^32768
}
template <uint32_t W>
__global__ void __launch_bounds__(EVAL_T) hal_batch_evaluate_any_kernel(const uint32_t* __restrict__ coeffs, int log_size, const uint32_t* __restrict__ which,
                                                                        const uint32_t* __restrict__ tab, uint32_t* __restrict__ out) {
    __shared__ uint32_t part[EVAL_T][4];
    const uint64_t n = (uint64_t)1 << log_size;
    const uint32_t e = blockIdx.x, t = threadIdx.x;
    const uint32_t* c = coeffs + (uint64_t)which[e] * n + t;
    const uint32_t* tb = tab + 4 * (uint64_t)EVAL_TAB * e;
    const uint32_t* yp = tb + 4 * 80;
    const uint64_t m = n >= EVAL_T ? n / EVAL_T : 1;           // coefficients per lane (lanes t >= n of a short polynomial hold none)
    const bool live = t < n;
    Ext acc = ext_zero();
    const uint4 yt = *reinterpret_cast<const uint4*>(yp + 4 * EVAL_BLK);
    const Ext y32{{(uint32_t)__builtin_amdgcn_readfirstlane(yt.x), (uint32_t)__builtin_amdgcn_readfirstlane(yt.y),
                   (uint32_t)__builtin_amdgcn_readfirstlane(yt.z), (uint32_t)__builtin_amdgcn_readfirstlane(yt.w)}};
    for (uint64_t b = (m + EVAL_BLK - 1) / EVAL_BLK; b-- > 0;) {
        uint32_t v[EVAL_BLK];
#pragma unroll
        for (uint32_t j = 0; j < EVAL_BLK; j++) {
            const uint64_t k = b * EVAL_BLK + j;
            v[j] = (live && k < m) ? __builtin_nontemporal_load(c + k * EVAL_T) : 0u;
        }
        acc = ext_mul_t<W>(acc, y32);
        uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for (uint32_t j = 0; j < EVAL_BLK; j += 2) {
            const uint4 pa = *reinterpret_cast<const uint4*>(yp + 4 * j), pb = *reinterpret_cast<const uint4*>(yp + 4 * (j + 1));
            const uint32_t a0 = __builtin_amdgcn_readfirstlane(pa.x), a1 = __builtin_amdgcn_readfirstlane(pa.y), a2 = __builtin_amdgcn_readfirstlane(pa.z), a3 = __builtin_amdgcn_readfirstlane(pa.w);
            const uint32_t b0 = __builtin_amdgcn_readfirstlane(pb.x), b1 = __builtin_amdgcn_readfirstlane(pb.y), b2 = __builtin_amdgcn_readfirstlane(pb.z), b3 = __builtin_amdgcn_readfirstlane(pb.w);
            dacc2(s0, a0, v[j], b0, v[j + 1]); dacc2(s1, a1, v[j], b1, v[j + 1]);
            dacc2(s2, a2, v[j], b2, v[j + 1]); dacc2(s3, a3, v[j], b3, v[j + 1]);
        }
        acc = ext_add_d(acc, Ext{{dacc_finish(s0), dacc_finish(s1), dacc_finish(s2), dacc_finish(s3)}});
    }
    {   // x^t = x^(t mod 64) (x^64)^(t div 64)
        const uint4 wv = *reinterpret_cast<const uint4*>(tb + 4 * (64 + (t >> 6)));
        const Ext xw{{(uint32_t)__builtin_amdgcn_readfirstlane(wv.x), (uint32_t)__builtin_amdgcn_readfirstlane(wv.y),
                      (uint32_t)__builtin_amdgcn_readfirstlane(wv.z), (uint32_t)__builtin_amdgcn_readfirstlane(wv.w)}};
        acc = ext_mul_t<W>(ext_mul_t<W>(acc, ld_ext(tb + 4 * (t & 63u))), xw);
    }
    for (int k = 0; k < 4; k++) part[t][k] = acc.c[k];
    __syncthreads();
    for (int s = EVAL_T / 2; s > 0; s >>= 1) {
        if ((int)t < s) for (int k = 0; k < 4; k++) part[t][k] = dadd(part[t][k], part[t + s][k]);
        __syncthreads();
    }
    if (t < 4) out[4 * (uint64_t)e + t] = part[0][t];
}
__global__ void hal_gather_sample_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint64_t idx, uint64_t size, uint64_t stride) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < size) dst[g] = src[g * stride + idx];
}
__global__ void hal_scatter_kernel(uint32_t* __restrict__ into, const uint32_t* __restrict__ index, const uint32_t* __restrict__ offsets,
                                   const uint32_t* __restrict__ values, uint64_t rows) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    for (uint32_t k = index[r]; k < index[r + 1]; k++) into[offsets[k]] = values[k];
}
// inclusive prefix products of extension elements in three launches: (1) every thread multiplies its chunk of CH consecutive
// elements and the workgroup scans the 256 chunk products in LDS, leaving per-chunk exclusive prefixes and the block total;
// (2) one workgroup scans the block totals; (3) every thread replays its chunk from (block prefix * chunk prefix).
constexpr int SCAN_CH = 8;
template <uint32_t W>
__global__ void __launch_bounds__(256) hal_scan_blocks_kernel(const uint32_t* __restrict__ io, uint64_t n, uint32_t* __restrict__ chunk_pre, uint32_t* __restrict__ block_tot) {
    __shared__ uint32_t sh[256][4];
    const uint32_t t = threadIdx.x;
    const uint64_t c = (uint64_t)blockIdx.x * 256 + t, i0 = c * SCAN_CH;
    Ext prod = ext_one();
    for (int k = 0; k < SCAN_CH; k++) if (i0 + k < n) prod = ext_mul_t<W>(prod, ld_ext(io + 4 * (i0 + k)));
    for (int k = 0; k < 4; k++) sh[t][k] = prod.c[k];
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {                      // Hillis-Steele inclusive scan (the product is not commutative-sensitive)
        Ext v = Ext{{sh[t][0], sh[t][1], sh[t][2], sh[t][3]}};
        Ext u = ext_one();
        if ((int)t >= d) u = Ext{{sh[t - d][0], sh[t - d][1], sh[t - d][2], sh[t - d][3]}};
        __syncthreads();
        if ((int)t >= d) { v = ext_mul_t<W>(u, v); for (int k = 0; k < 4; k++) sh[t][k] = v.c[k]; }
        __syncthreads();
    }
    const Ext excl = t ? Ext{{sh[t - 1][0], sh[t - 1][1], sh[t - 1][2], sh[t - 1][3]}} : ext_one();
    st_ext(chunk_pre + 4 * c, excl);
    if (t == 255) st_ext(block_tot + 4 * (uint64_t)blockIdx.x, Ext{{sh[255][0], sh[255][1], sh[255][2], sh[255][3]}});
}
template <uint32_t W>
__global__ void __launch_bounds__(256) hal_scan_totals_kernel(uint32_t* __restrict__ block_tot, uint64_t nblocks) {
    // exclusive scan of the block totals, in place; one workgroup, sequential over tiles of 256
    __shared__ uint32_t sh[256][4];
    __shared__ uint32_t carry[4];
    const uint32_t t = threadIdx.x;
    if (t < 4) carry[t] = t == 0 ? MONTY_R1 : 0u;
    __syncthreads();
    for (uint64_t base = 0; base < nblocks; base += 256) {
        const uint64_t i = base + t;
        Ext v = i < nblocks ? ld_ext(block_tot + 4 * i) : ext_one();
        for (int k = 0; k < 4; k++) sh[t][k] = v.c[k];
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            Ext cur = Ext{{sh[t][0], sh[t][1], sh[t][2], sh[t][3]}};
            Ext u = ext_one();
            if ((int)t >= d) u = Ext{{sh[t - d][0], sh[t - d][1], sh[t - d][2], sh[t - d][3]}};
            __syncthreads();
            if ((int)t >= d) { cur = ext_mul_t<W>(u, cur); for (int k = 0; k < 4; k++) sh[t][k] = cur.c[k]; }
            __syncthreads();
        }
        const Ext c0 = Ext{{carry[0], carry[1], carry[2], carry[3]}};
        const Ext excl = t ? Ext{{sh[t - 1][0], sh[t - 1][1], sh[t - 1][2], sh[t - 1][3]}} : ext_one();
        if (i < nblocks) st_ext(block_tot + 4 * i, ext_mul_t<W>(c0, excl));
        __syncthreads();
        if (t == 0) { const Ext nc = ext_mul_t<W>(c0, Ext{{sh[255][0], sh[255][1], sh[255][2], sh[255][3]}}); for (int k = 0; k < 4; k++) carry[k] = nc.c[k]; }
        __syncthreads();
    }
}
template <uint32_t W>
__global__ void __launch_bounds__(256) hal_scan_apply_kernel(uint32_t* __restrict__ io, uint64_t n, const uint32_t* __restrict__ chunk_pre, const uint32_t* __restrict__ block_pre) {
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x, i0 = c * SCAN_CH;
    if (i0 >= n) return;
    Ext acc = ext_mul_t<W>(ld_ext(block_pre + 4 * (uint64_t)blockIdx.x), ld_ext(chunk_pre + 4 * c));
    for (int k = 0; k < SCAN_CH; k++)
        if (i0 + k < n) { acc = ext_mul_t<W>(acc, ld_ext(io + 4 * (i0 + k))); st_ext(io + 4 * (i0 + k), acc); }
}

// ---- SHA-256 (FIPS 180-4) over canonical words, big-endian; one row / one node per lane.  Integer-ALU bound (64 rounds of
// rotates and adds per 64-byte block), the column-major reads are coalesced (a wave reads 64 consecutive words of a column).
__constant__ uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
ZK_D uint32_t rotr32(uint32_t x, int n) { return __builtin_rotateright32(x, n); }
ZK_D void sha256_block(uint32_t h[8], uint32_t w[16]) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
            const uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        const uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25), ch = (e & f) ^ (~e & g);
        const uint32_t t1 = hh + S1 + ch + K256[i] + w[i & 15];
        const uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
        const uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
ZK_D void sha256_init(uint32_t h[8]) {
    h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a; h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
}
// leaf r = SHA-256 over the canonical words of row r of a column-major [cols][rows] matrix (Montgomery in memory)
__global__ void __launch_bounds__(256) hal_hash_rows_sha256_kernel(const uint32_t* __restrict__ mat, uint64_t cols, uint64_t rows, uint32_t* __restrict__ digests) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    uint32_t h[8], w[16];
    sha256_init(h);
    uint64_t i = 0;
    for (; i + 16 <= cols; i += 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) w[k] = from_monty(mat[(i + k) * rows + r]);
        sha256_block(h, w);
    }
    int n = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { w[k] = 0u; if (i + k < cols) { w[k] = from_monty(mat[(i + k) * rows + r]); n = k + 1; } }
    // padding: 0x80 byte, zeros, 64-bit message length in bits
#pragma unroll
    for (int k = 0; k < 16; k++) if (k == n) w[k] = 0x80000000u;
    if (n > 13) {
        sha256_block(h, w);
#pragma unroll
        for (int k = 0; k < 16; k++) w[k] = 0u;
    }
    const uint64_t bits = cols * 32;
    w[14] = (uint32_t)(bits >> 32); w[15] = (uint32_t)bits;
    sha256_block(h, w);
    uint4* d = reinterpret_cast<uint4*>(digests + 8 * r);
    d[0] = make_uint4(h[0], h[1], h[2], h[3]);
    d[1] = make_uint4(h[4], h[5], h[6], h[7]);
}
// parents[i] = SHA-256(children[2i] || children[2i+1]): one 64-byte data block + the padding block
__global__ void __launch_bounds__(256) hal_hash_fold_sha256_kernel(const uint32_t* __restrict__ children, uint32_t* __restrict__ parents, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint4* cp = reinterpret_cast<const uint4*>(children + 16 * i);
    const uint4 v0 = cp[0], v1 = cp[1], v2 = cp[2], v3 = cp[3];
    uint32_t w[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    uint32_t h[8];
    sha256_init(h);
    sha256_block(h, w);
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = 0u;
    w[0] = 0x80000000u; w[15] = 512u;
    sha256_block(h, w);
    uint4* d = reinterpret_cast<uint4*>(parents + 8 * i);
    d[0] = make_uint4(h[0], h[1], h[2], h[3]);
    d[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

static unsigned grid_for(uint64_t n, unsigned cap = 65536) { const uint64_t b = (n + 255) / 256; return (unsigned)(b < cap ? (b ? b : 1) : cap); }

}  // namespace zk

using namespace zk;

#define CHECK_CTX(ctx)                                                  \
    do {                                                                \
        if (!(ctx)) return fail(ZKHIP_ERR_INVALID, "null context");     \
        ZK_HIP(hipSetDevice((ctx)->device));                            \
    } while (0)
#define LAUNCHED() ZK_HIP(hipGetLastError())

extern "C" {

int zkhip_eltwise_add(zkhip_ctx* ctx, uint32_t* d_out, const uint32_t* d_a, const uint32_t* d_b, size_t n) {
    CHECK_CTX(ctx);
    if (n && (!d_out || !d_a || !d_b)) return fail(ZKHIP_ERR_INVALID, "eltwise_add: null pointer");
    if (!n) return ZKHIP_OK;
    const int vec = ((reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(d_a) | reinterpret_cast<uintptr_t>(d_b)) & 15) == 0;
    hipLaunchKernelGGL(hal_add_kernel, dim3(grid_for(vec ? (n + 3) / 4 : n, 8192)), dim3(256), 0, ctx->stream, d_out, d_a, d_b, (uint64_t)n, vec);
    LAUNCHED();
    return ZKHIP_OK;
}
int zkhip_eltwise_copy(zkhip_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t n) {
    CHECK_CTX(ctx);
    if (n && (!d_out || !d_in)) return fail(ZKHIP_ERR_INVALID, "eltwise_copy: null pointer");
    if (n) ZK_HIP(hipMemcpyAsync(d_out, d_in, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return ZKHIP_OK;
}
int zkhip_eltwise_zeroize(zkhip_ctx* ctx, uint32_t* d_io, size_t n) {
    CHECK_CTX(ctx);
    if (n && !d_io) return fail(ZKHIP_ERR_INVALID, "eltwise_zeroize: null pointer");
    if (!n) return ZKHIP_OK;
    const int vec = (reinterpret_cast<uintptr_t>(d_io) & 15) == 0;
    hipLaunchKernelGGL(hal_zeroize_kernel, dim3(grid_for(vec ? (n + 3) / 4 : n, 8192)), dim3(256), 0, ctx->stream, d_io, (uint64_t)n, vec);
    LAUNCHED();
    return ZKHIP_OK;
}
int zkhip_eltwise_sum_ext(zkhip_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t count, size_t to_add) {
    CHECK_CTX(ctx);
    if (count && (!d_out || !d_in)) return fail(ZKHIP_ERR_INVALID, "eltwise_sum_ext: null pointer");
    if ((reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(d_in)) & 15) return fail(ZKHIP_ERR_INVALID, "eltwise_sum_ext: extension vectors must be 16-byte aligned");
    if (!count) return ZKHIP_OK;
    hipLaunchKernelGGL(hal_sum_ext_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, d_out, d_in, (uint64_t)count, (uint64_t)to_add);
    LAUNCHED();
    return ZKHIP_OK;
}
int zkhip_zk_shift(zkhip_ctx* ctx, uint32_t* d_io, size_t count, int log_size, uint32_t shift) {
    CHECK_CTX(ctx);
    if (!d_io || log_size < 0 || log_size > 30 || shift == 0 || shift >= P) return fail(ZKHIP_ERR_INVALID, "zk_shift: bad arguments (shift canonical, non-zero)");
    if (!count) return ZKHIP_OK;
    // a lane owns 4 coefficients per step; ~16 steps per lane amortise its fpow; whole polynomials per grid row (at most 65535 rows per
    // launch: more polynomials take several launches)
    const bool vec = log_size < 2 || (reinterpret_cast<uintptr_t>(d_io) & 15) == 0;
    const uint64_t n = (uint64_t)1 << log_size;
    const uint64_t quads = log_size >= 2 ? n / 4 : 1;
    uint64_t blocks = vec ? (quads + 256 * 16 - 1) / (256 * 16) : (n + 256 * 16 - 1) / (256 * 16);
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    const uint32_t sm = to_monty(shift), s2 = fmul(sm, sm), s3 = fmul(s2, sm), step = fpow(sm, (vec ? 4 : 1) * blocks * 256);
    for (size_t done = 0; done < count; done += 65535) {
        const size_t rows = count - done < 65535 ? count - done : 65535;
        uint32_t* base = d_io + done * n;
        if (vec) hipLaunchKernelGGL(hal_zk_shift_kernel, dim3((unsigned)blocks, (unsigned)rows), dim3(256), 0, ctx->stream, base, log_size, sm, sm, s2, s3, step);
        else hipLaunchKernelGGL(hal_zk_shift_scalar_kernel, dim3((unsigned)blocks, (unsigned)rows), dim3(256), 0, ctx->stream, base, log_size, sm, step);
        LAUNCHED();
    }
    return ZKHIP_OK;
}
static int ext_field_ok(int ext) { return ext == ZKHIP_EXT_X4_MINUS_11 || ext == ZKHIP_EXT_X4_PLUS_11; }
int zkhip_mix_poly_coeffs(zkhip_ctx* ctx, uint32_t* d_out, const uint32_t mix_start[4], const uint32_t mix[4], const uint32_t* d_in,
                          const uint32_t* d_combos, size_t input_size, size_t count, int ext_field) {
    CHECK_CTX(ctx);
    if (!d_out || !mix_start || !mix || !d_in || !d_combos || !ext_field_ok(ext_field) || (reinterpret_cast<uintptr_t>(d_out) & 15))
        return fail(ZKHIP_ERR_INVALID, "mix_poly_coeffs: bad arguments");
    if (!count || !input_size) return ZKHIP_OK;
    const Ext ms{{mix_start[0], mix_start[1], mix_start[2], mix_start[3]}}, mx{{mix[0], mix[1], mix[2], mix[3]}};
    void* v_pow;
    ZK_TRY(ctx_reserve(ctx, S_COL_B, input_size * 16, &v_pow));
    uint32_t* powers = (uint32_t*)v_pow;
    if (ext_field == ZKHIP_EXT_X4_PLUS_11) hipLaunchKernelGGL((hal_mix_powers_kernel<MONTY_W_R0>), dim3(1), dim3(64), 0, ctx->stream, powers, ms, mx, (uint64_t)input_size);
    else hipLaunchKernelGGL((hal_mix_powers_kernel<MONTY_W_SP1>), dim3(1), dim3(64), 0, ctx->stream, powers, ms, mx, (uint64_t)input_size);
    LAUNCHED();
    if (input_size <= MIX_PLAN_MAX) {                            // terms grouped by combo: one set of running sums per run, any number of combos
        void* v_plan;
        ZK_TRY(ctx_reserve(ctx, S_COL_A, input_size * 28, &v_plan));
        uint32_t *spow = (uint32_t*)v_plan, *srow = spow + 4 * input_size, *scombo = srow + input_size, *send = scombo + input_size;
        hipLaunchKernelGGL(hal_mix_plan_kernel, dim3(1), dim3(256), 0, ctx->stream, d_combos, powers, (uint32_t)input_size, spow, srow, scombo, send);
        LAUNCHED();
        hipLaunchKernelGGL(hal_mix_sorted_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, d_out, d_in, spow, srow, scombo, send, (uint32_t)input_size, (uint64_t)count);
        LAUNCHED();
        return ZKHIP_OK;
    }
    hipLaunchKernelGGL(hal_mix_poly_coeffs_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, d_out, powers, d_in, d_combos, (uint64_t)input_size, (uint64_t)count);
    LAUNCHED();
    return ZKHIP_OK;
}
int zkhip_batch_evaluate_any(zkhip_ctx* ctx, const uint32_t* d_coeffs, int log_size, const uint32_t* d_which, const uint32_t* d_xs,
                             uint32_t* d_out, size_t eval_count, int ext_field) {
    CHECK_CTX(ctx);
    if (!d_coeffs || !d_which || !d_xs || !d_out || log_size < 0 || log_size > 30 || !ext_field_ok(ext_field) || (reinterpret_cast<uintptr_t>(d_xs) & 15))
        return fail(ZKHIP_ERR_INVALID, "batch_evaluate_any: bad arguments");
    if (!eval_count) return ZKHIP_OK;
    void* v_tab;
    ZK_TRY(ctx_reserve(ctx, S_COL_B, eval_count * EVAL_TAB * 16, &v_tab));
    uint32_t* tab = (uint32_t*)v_tab;
    if (ext_field == ZKHIP_EXT_X4_PLUS_11) {
        hipLaunchKernelGGL((hal_eval_tables_kernel<MONTY_W_R0>), dim3((unsigned)eval_count), dim3(64), 0, ctx->stream, d_xs, tab);
        hipLaunchKernelGGL((hal_batch_evaluate_any_kernel<MONTY_W_R0>), dim3((unsigned)eval_count), dim3(EVAL_T), 0, ctx->stream, d_coeffs, log_size, d_which, tab, d_out);
    } else {
        hipLaunchKernelGGL((hal_eval_tables_kernel<MONTY_W_SP1>), dim3((unsigned)eval_count), dim3(64), 0, ctx->stream, d_xs, tab);
        hipLaunchKernelGGL((hal_batch_evaluate_any_kernel<MONTY_W_SP1>), dim3((unsigned)eval_count), dim3(EVAL_T), 0, ctx->stream, d_coeffs, log_size, d_which, tab, d_out);
    }
    LAUNCHED();
    return ZKHIP_OK;
}
int zkhip_gather_sample(zkhip_ctx* ctx, uint32_t* d_dst, const uint32_t* d_src, size_t idx, size_t size, size_t stride) {
    CHECK_CTX(ctx);
    if (!d_dst || !d_src || idx >= stride) return fail(ZKHIP_ERR_INVALID, "gather_sample: bad arguments (idx < stride)");
    if (!size) return ZKHIP_OK;
    hipLaunchKernelGGL(hal_gather_sample_kernel, dim3((unsigned)((size + 255) / 256)), dim3(256), 0, ctx->stream, d_dst, d_src, (uint64_t)idx, (uint64_t)size, (uint64_t)stride);
    LAUNCHED();
    return ZKHIP_OK;
}
int zkhip_scatter(zkhip_ctx* ctx, uint32_t* d_into, const uint32_t* d_index, const uint32_t* d_offsets, const uint32_t* d_values, size_t rows) {
    CHECK_CTX(ctx);
    if (!d_into || !d_index || !d_offsets || !d_values) return fail(ZKHIP_ERR_INVALID, "scatter: null pointer");
    if (!rows) return ZKHIP_OK;
    hipLaunchKernelGGL(hal_scatter_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, d_into, d_index, d_offsets, d_values, (uint64_t)rows);
    LAUNCHED();
    return ZKHIP_OK;
}
int zkhip_prefix_products_ext(zkhip_ctx* ctx, uint32_t* d_io, size_t n, int ext_field) {
    CHECK_CTX(ctx);
    if (!d_io || !ext_field_ok(ext_field) || (reinterpret_cast<uintptr_t>(d_io) & 15)) return fail(ZKHIP_ERR_INVALID, "prefix_products_ext: bad arguments");
    if (!n) return ZKHIP_OK;
    const uint64_t chunks = (n + SCAN_CH - 1) / SCAN_CH, nblocks = (chunks + 255) / 256;
    void *v_pre, *v_tot;
    ZK_TRY(ctx_reserve(ctx, S_COL_A, nblocks * 256 * 16, &v_pre));
    ZK_TRY(ctx_reserve(ctx, S_COL_B, nblocks * 16, &v_tot));
    uint32_t *pre = (uint32_t*)v_pre, *tot = (uint32_t*)v_tot;
    const dim3 grid((unsigned)nblocks), block(256);
    if (ext_field == ZKHIP_EXT_X4_PLUS_11) {
        hipLaunchKernelGGL((hal_scan_blocks_kernel<MONTY_W_R0>), grid, block, 0, ctx->stream, d_io, (uint64_t)n, pre, tot);
        hipLaunchKernelGGL((hal_scan_totals_kernel<MONTY_W_R0>), dim3(1), block, 0, ctx->stream, tot, nblocks);
        hipLaunchKernelGGL((hal_scan_apply_kernel<MONTY_W_R0>), grid, block, 0, ctx->stream, d_io, (uint64_t)n, pre, tot);
    } else {
        hipLaunchKernelGGL((hal_scan_blocks_kernel<MONTY_W_SP1>), grid, block, 0, ctx->stream, d_io, (uint64_t)n, pre, tot);
        hipLaunchKernelGGL((hal_scan_totals_kernel<MONTY_W_SP1>), dim3(1), block, 0, ctx->stream, tot, nblocks);
        hipLaunchKernelGGL((hal_scan_apply_kernel<MONTY_W_SP1>), grid, block, 0, ctx->stream, d_io, (uint64_t)n, pre, tot);
    }
    LAUNCHED();
    return ZKHIP_OK;
}
int zkhip_hash_rows_sha256(zkhip_ctx* ctx, const uint32_t* d_mat, size_t cols, size_t rows, uint32_t* d_digests) {
    CHECK_CTX(ctx);
    if (!d_mat || !d_digests || cols == 0 || (reinterpret_cast<uintptr_t>(d_digests) & 15)) return fail(ZKHIP_ERR_INVALID, "hash_rows_sha256: bad arguments");
    if (!rows) return ZKHIP_OK;
    hipLaunchKernelGGL(hal_hash_rows_sha256_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, d_mat, (uint64_t)cols, (uint64_t)rows, d_digests);
    LAUNCHED();
    return ZKHIP_OK;
}
int zkhip_hash_fold_sha256(zkhip_ctx* ctx, const uint32_t* d_children, uint32_t* d_parents, size_t count) {
    CHECK_CTX(ctx);
    if (!d_children || !d_parents || ((reinterpret_cast<uintptr_t>(d_children) | reinterpret_cast<uintptr_t>(d_parents)) & 15))
        return fail(ZKHIP_ERR_INVALID, "hash_fold_sha256: bad arguments");
    if (!count) return ZKHIP_OK;
    hipLaunchKernelGGL(hal_hash_fold_sha256_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, d_children, d_parents, (uint64_t)count);
    LAUNCHED();
    return ZKHIP_OK;
}
// full tree, leaves first (layout of zkhip_merkle_commit): leaf hashing + one fold launch per level
int zkhip_merkle_commit_sha256_colmajor(zkhip_ctx* ctx, const uint32_t* d_mat, uint32_t cols, int log_rows, uint32_t* d_tree) {
    CHECK_CTX(ctx);
    if (!d_mat || !d_tree || cols == 0 || log_rows < 0 || log_rows > 28) return fail(ZKHIP_ERR_INVALID, "merkle_commit_sha256_colmajor: bad arguments");
    const uint64_t rows = (uint64_t)1 << log_rows;
    ZK_TRY(zkhip_hash_rows_sha256(ctx, d_mat, cols, rows, d_tree));
    uint32_t* level = d_tree;
    for (uint64_t cnt = rows; cnt > 1; cnt >>= 1) {
        uint32_t* next = level + 8 * cnt;
        ZK_TRY(zkhip_hash_fold_sha256(ctx, level, next, cnt / 2));
        level = next;
    }
    return ZKHIP_OK;
}

}  // extern "C"
