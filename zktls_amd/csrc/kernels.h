// kernels.h -- internal launch interface between the gfx950 kernel translation units
// (ntt.hip, hash.hip, stark.hip, util.hip) and the host-side orchestration (context.cpp,
// prover.cpp).  Nothing here is part of the public C ABI (include/zkhip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "babybear.cuh"

namespace zk {

// ---------------------------------------------------------------- NTT pass (ntt.hip)
// One launch = one pass of a (up to) two-pass NTT over the columns of a row-major
// matrix: every workgroup transforms a tile of M = 2^log_m rows x C columns.
//   in row of tile element n   = tile * in_tile_mul  + n * in_stride
//   out row of tile element k  = tile * out_tile_mul + o(k) * out_stride,
//   o(k) = bitrev_m(k) if bitrev_out else k
//   value written = post[tile*M + k] * NTT_M( pre[n] * in[n] )[k]
struct NttPassArgs {
    const uint32_t* in;
    uint32_t* out;
    uint64_t in_ld, out_ld;          // row pitch, in elements
    uint32_t ncols;
    uint32_t num_tiles;
    uint32_t log_m;                  // 5..10
    uint64_t in_tile_mul, in_stride;
    uint64_t out_tile_mul, out_stride;
    uint32_t bitrev_out;
    uint32_t map_mode;               // 0: column group fastest; 1: XCD-aware; 2..15: XCD-aware + tile index rotated by that many bits; 255: automatic
    uint64_t tile_perm;              // A/B builds only: if non-zero, tile index bit b moves to bit (tile_perm >> 4b) & 15 (after the XCD-aware map)
    uint32_t fast_path;              // A/B builds only: 0 / 2 tile-per-workgroup kernel (default), 1 persistent 1024 x 32 kernel
    uint32_t debug_flags;            // A/B builds (-DZKHIP_AB_HOOKS) only, ignored otherwise: 1 drop loads, 2 drop stores, 4 no transform, 8 no non-temporal policy
    uint32_t cols_per_thread;        // 2: two columns per lane when the shape allows (A/B knob); else 1
    uint32_t bench_tag;              // 1: launched by the roofline hook (zkhip_ntt_pass): same code under its own kernel name (NT = 3 / 4)
    const uint32_t* w1024;           // w_1024^e (forward) or w_1024^-e (inverse), e < 1024
    const uint32_t* pre;             // [M] or nullptr
    const uint32_t* post;            // [num_tiles * M] or nullptr
};
hipError_t launch_ntt_pass(const NttPassArgs& a, bool inverse, hipStream_t s);

// ---- fused inverse-second / forward-first pass of a two-pass LDE with 1024-row tiles (ntt_fused.hip).  One workgroup reads a tile
// once, finishes the inverse transform in registers (the coefficients never reach memory), and for each of TWO cosets t runs
// the first forward pass on them:  out_t[k] = post_t[tile * 1024 + k] * NTT_1024( pre_t[n] * INTT_1024( in )[n] )[k].
//   in row of tile element n     = tile * in_tile_mul + n * in_stride           (intermediate left by the first inverse pass)
//   out_t row of tile element k  = tile * out_tile_mul + bitrev10(k) * out_stride
// 12 B per tile element at two cosets (read 4, write 2 x 4) against 24 B for the three launches it replaces.
constexpr int FUSED_COSETS = 2;           // cosets per launch (a blowup of 4 is two launches)
struct LdeFusedArgs {
    const uint32_t* in;
    uint64_t in_ld, out_ld;
    uint32_t ncols;                  // multiple of 32
    uint32_t num_tiles;              // multiple of 8
    uint32_t map_rot;                // tile index rotated by this many bits after the XCD-aware map (0: none)
    uint32_t grid;                   // workgroups to launch (0: automatic = one per CU, walking the tile list); >= items: one tile each
    uint32_t bench_tag;              // 1: launched by the roofline hook under its own kernel name
    uint64_t in_tile_mul, in_stride;
    uint64_t out_tile_mul, out_stride;
    const uint32_t* w1024_inv;       // w_1024^-(u k1) in thread order (launch_fused_table mode 0)
    const uint32_t* w1024_fwd;       // w_1024^(u k1), thread order
    uint32_t* out[FUSED_COSETS];
    const uint32_t* pre[FUSED_COSETS];    // [1024] each, thread order (mode 1)
    const uint32_t* post[FUSED_COSETS];   // [num_tiles * 1024] each, thread order (mode 2)
};
hipError_t launch_lde_fused(const LdeFusedArgs& a, hipStream_t s);
// the fused launch reads its tables in THREAD order (thread u's 32 entries contiguous, 16-byte loads): per block of 1024 words
//   mode 0: out[32 u + r]   = in[u * rev5(r)]          (tile twiddles w_1024^(+-u k1), one block)
//   mode 1: out[32 u + n1]  = in[u + 32 n1]            (coset powers, one block)
//   mode 2: out[32 u + rho] = in[32 rev5(rho) + u]     (post table, one block per tile)
hipError_t launch_fused_table(uint32_t* out, const uint32_t* in, uint32_t blocks, int mode, hipStream_t s);
bool lde_fused_supported(const LdeFusedArgs& a);      // shape / alignment / 32-bit tile span

// ---- a whole coset LDE of 2^11 .. 2^15 rows in ONE launch (ntt_small.hip): column in registers, 4 + 4 * cosets bytes per trace cell
struct LdeSmallArgs {
    const uint32_t* in;
    uint64_t in_ld, out_ld;
    uint32_t ncols;
    uint32_t cosets;                 // 1 .. 16
    uint32_t groups;                 // column groups of 32 / P columns (filled in by the launcher)
    uint32_t reserved;
    const uint32_t* tw_inv;          // w_N^(-i), i < N / 32
    const uint32_t* tw_fwd;          // w_N^(+i), i < N / 32
    uint32_t* out[16];               // per coset: first row of its block of the LDE
    const uint32_t* pre[16];         // per coset: shift_t^j / N, j < N
};
bool lde_small_supported(const LdeSmallArgs& a, int log_n);
hipError_t launch_lde_small(const LdeSmallArgs& a, int log_n, hipStream_t s);

// ---- native passes over CONTIGUOUS VECTORS (RISC Zero's Hal layout: `count` polynomials of 2^20 coefficients, column-major).
// A polynomial is viewed as a 1024 x 1024 matrix A[r][c] = v[1024 r + c]; a pass transforms the 1024-point columns of 32
// adjacent c at a time (the tile shape of ntt_pass_kernel<4, *, 2, 5>), and where the four-step transpose requires it the tile
// is loaded or stored TRANSPOSED through LDS, so that every global access is still a run of >= 2 KiB: no separate transpose pass.
//   plain side:       element (i, col) at  base + 1024 * i + col                      (lanes along the 32 columns, 128-byte chunks)
//   transposed side:  element (i, col) at  base + 1024 * map(col) + map(i)            (lanes along i: whole 4 KiB lines), map = bitrev10 if *_brev
struct ColPassArgs {
    const uint32_t* in;
    uint32_t* out;
    uint64_t in_batch, out_batch;    // elements between consecutive polynomials on either side
    uint32_t count;
    uint32_t tload, tstore, in_brev, out_brev;
    const uint32_t* w1024;           // w_1024^(+-e)
    const uint32_t* pre;             // [1024] or null: multiplies input element i
    const uint32_t* post2d;          // [1024][1024] or null: multiplies output element (k, col) by post2d[1024 k + col]
};
hipError_t launch_ntt_colpass(const ColPassArgs& a, bool inverse, hipStream_t s);
// out[i * cols + c] = scale * base_row^i ... generic 2-D table: out[1024 k + c] = scale * w^(k c) * shift^c
hipError_t launch_post2d_table(uint32_t* out, uint32_t w, uint32_t shift, uint32_t scale, hipStream_t s);

// ---- transforms of 2^21 / 2^22 rows: N = R N', R = 2^log_r (1 or 2), N' <= 2^20.  The N'-point transforms of the R row classes
// j = n mod R run through the pass kernel above on sub-matrices (row pitch R ld); this streaming pass is the remaining radix-R
// step (one read + one write of every element, 8 B/element):
//   forward:  out[g, k2] = sum_j w_R^(j k2) * tw[j][g] * in[g, j]                      (tw[0] is not read: it is 1)
//   inverse:  out[g, j]  = tw[j][g] * sum_k2 w_R^-(j k2) * in[g, k2]                   (tw[j] carries 1/R)
// element (g, e) of a side lives in row g * group_mul + slot(e) * elem_mul, slot(e) = bitrev_r(e) if bitrev else e.
struct CombineArgs {
    const uint32_t* in;
    uint32_t* out;
    uint64_t in_ld, out_ld;
    uint32_t ncols;
    uint64_t groups;                 // N'
    int log_r;                       // 1 or 2
    uint64_t in_group_mul, in_elem_mul, out_group_mul, out_elem_mul;
    int bitrev_out;                  // forward only: output slot = bitrev_r(k2) (in-place form on a bit-reversed LDE)
    int inverse;
    const uint32_t* tw;              // [R][groups], Montgomery
};
hipError_t launch_ntt_combine(const CombineArgs& a, hipStream_t s);
// tw[j][g] = scale * shift^j * w^(j * idx(g)), idx(g) = bitrev_bits(g) if bits > 0 else g; j < rows
hipError_t launch_combine_table(uint32_t* out, uint32_t rows, uint64_t groups, uint32_t w, uint32_t shift, uint32_t scale, int bits, hipStream_t s);

// out[i] = scale * base^i, i < n   (Montgomery form in and out)
hipError_t launch_pow_table(uint32_t* out, size_t n, uint32_t base, uint32_t scale, hipStream_t s);
// out[i*cols + k] = scale * omega^(i*k) * shift^i
hipError_t launch_post_table(uint32_t* out, uint32_t rows, uint32_t cols, uint32_t omega,
                             uint32_t shift, uint32_t scale, hipStream_t s);

// ---------------------------------------------------------------- hashing (hash.hip)
struct MatDesc {
    const uint32_t* ptr;
    uint64_t ld;       // row pitch in elements
    uint32_t width;
};
constexpr int MAX_LEAF_MATS = 8;
// levels / heights up to this many nodes use the 16-lanes-per-permutation kernels (latency-bound regime)
constexpr uint32_t COOP_MAX_NODES = 16384;
// ... as the calling prover sees it: those kernels trade throughput for latency, which pays while the WHOLE launch is small -- inside a
// lock-step batch (batch.h) a merged launch carries every member's nodes, so the bound is divided by the members (batch.cpp)
uint32_t coop_max_nodes();
// the single-workgroup kernel finishes a tree from this many nodes down to the root
constexpr uint32_t COOP_TOP_NODES = 512;
struct LeafArgs {
    MatDesc mats[MAX_LEAF_MATS];
    int nmats;
    uint64_t height;
    uint32_t* digests;   // [height][8]
};
hipError_t launch_hash_rows(const LeafArgs& a, hipStream_t s);
// parents[i] = compress(children[2i], children[2i+1]), i < count
hipError_t launch_compress_level(const uint32_t* children, uint32_t* parents, uint64_t count,
                                 hipStream_t s);
// nodes[i] = compress(nodes[i], extra[i])  (mixed-height commitments)
hipError_t launch_inject(uint32_t* nodes, const uint32_t* extra, uint64_t count, hipStream_t s);
// one level with an injection in one launch: a.digests[i] = compress(compress(children[2i], children[2i+1]), sponge(row i of a.mats)),
// i < a.height; for levels above the latency-bound regime whose matrices allow 16-byte loads (compress_inject_ok)
bool compress_inject_ok(const LeafArgs& a);
hipError_t launch_compress_inject(const uint32_t* children, const LeafArgs& a, hipStream_t s);
// all remaining levels of a small subtree in one launch: `tree` points at a level with
// `count` (<= COOP_TOP_NODES, power of two) digests followed by room for the levels above it
hipError_t launch_compress_top(uint32_t* tree, uint32_t count, hipStream_t s);
// log2(sub) levels of a medium tree (count <= COOP_MAX_NODES) in one launch: every workgroup reduces `sub` consecutive digests to one
hipError_t launch_compress_sub(uint32_t* tree, uint32_t count, uint32_t sub, hipStream_t s);
// the leaf digests AND log2(sub) levels of a medium tree (height <= COOP_MAX_NODES) in one launch: workgroup b hashes the rows it owns
hipError_t launch_hash_sub(const LeafArgs& a, uint32_t sub, hipStream_t s);
// RISC Zero layout: column-major [cols][rows], Poseidon2 width 24 (rate 16); full tree, leaves first
hipError_t launch_merkle_p24_colmajor(const uint32_t* mat, uint32_t cols, int log_rows, uint32_t* tree, hipStream_t s);
// same hash over a row-major matrix (width % 4 == 0, ld % 4 == 0, 16-byte aligned)
hipError_t launch_merkle_p24_rowmajor(const uint32_t* mat, uint64_t ld, uint32_t width, int log_rows, uint32_t* tree, hipStream_t s);
// upload the Poseidon2 tables in effect (params.cpp) into the __constant__ copy of hash.hip / stark.hip on the current device
struct P2Tables;
hipError_t hash_upload_p2_tables(const P2Tables& t, hipStream_t s);
hipError_t stark_upload_p2_tables(const P2Tables& t, hipStream_t s);
// trace of the Poseidon2 permutation chip for a set of Merkle paths (p2chip.h)
namespace p2chip { struct MerkleTraceArgs; struct LayerPathsArgs; struct P2RArgs; struct MrecChainArgs; }
hipError_t launch_p2chip_merkle(const p2chip::MerkleTraceArgs& a, hipStream_t s);
hipError_t launch_p2chip_layer_paths(const p2chip::LayerPathsArgs& a, hipStream_t s);     // the FRI-layers variant: paths of different depths
hipError_t launch_mrec_chains(const p2chip::MrecChainArgs& a, hipStream_t s);                // machine mode: the queries' Poseidon2 chains walked on the device
hipError_t launch_p2r_rows(const p2chip::P2RArgs& a, hipStream_t s);                        // the shard verifier's chip: chains of sponge + path rows, transcript rows
// raw permutation on `count` states of 16 words (KAT / microbenchmark)
hipError_t launch_permute_states(uint32_t* states, uint64_t count, hipStream_t s);

// ---------------------------------------------------------------- utilities (util.hip)
hipError_t launch_fill_uniform(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows,
                               uint32_t width, hipStream_t s);
hipError_t launch_gen_trace(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows,
                            uint32_t width, hipStream_t s);
hipError_t launch_gen_trace_logup(uint32_t* out, uint64_t ld, uint64_t seed, uint64_t rows, uint32_t width, uint32_t pairs,
                                  uint64_t recv_seed, uint32_t recv_width, hipStream_t s);
// [rows][cols] -> [cols][rows], optional bit reversal of the column index on either side
// naive evaluation for N < 32: out[row(k)] = scale * sum_j in[j] (shift w^k)^j (see util.hip)
hipError_t launch_small_eval(const uint32_t* in, uint64_t in_ld, uint32_t* out, uint64_t out_ld, int log_n, int log_out, uint32_t width,
                             uint32_t w, uint32_t shift, uint32_t scale, int bitrev_out, hipStream_t s);
hipError_t launch_transpose(const uint32_t* in, uint32_t* out, uint64_t rows, uint64_t cols, int rev_bits_in, int rev_bits_out, hipStream_t s);
// element-wise Montgomery <-> canonical conversion
hipError_t launch_convert(const uint32_t* in, uint32_t* out, uint64_t n, bool to_monty_form,
                          hipStream_t s);

// ---------------------------------------------------------------- STARK stages (stark.hip)
// x_p = g w_2N^bitrev(p) (p < 2N), Z_H(x_p)/(x_p - 1), and 1/(2 w_2N^bitrev_n(i)) (i < N)
hipError_t launch_domain_tables(uint32_t* xs, uint32_t* sel_first, uint32_t* sel_last, uint32_t* itw, int log_n, int log_blowup, hipStream_t s);

struct QuotientArgs {
    const uint32_t* lde;        // [2N][ld] trace LDE, bit-reversed rows
    uint64_t ld;
    uint32_t width;
    int log_n;
    int lanes_per_row;          // power of two <= 16 (one DPP row)
    const uint32_t* xs;
    const uint32_t* sel_first;
    uint32_t wn_inv;            // w_N^-1
    uint32_t inv_zh_even, inv_zh_odd;
    const uint32_t* alpha_pow;  // [G][16] words: three ext weights + (g+1, 2g+3, 5g+7, 0) in Montgomery form; then [Q + 3] ext.
                                // weight of constraint k is alpha^(K-1-k), K = 3 G + (Q ? Q + 3 : 0)
    // LogUp part (pairs = Q > 0): permutation-trace LDE [2N][perm_ld], lookup challenges, last-row selector
    uint32_t pairs;
    const uint32_t* perm;
    uint64_t perm_ld;
    Ext gamma, beta;
    Ext cumsum;                 // last-row constraint S = cumsum (zero unless tables look each other up)
    const uint32_t* sel_last;
    uint32_t* addend_out;       // logup_addend_kernel: [2N][4], the LogUp constraints of every point (weights folded in), storage order
    const uint32_t* addend;     // quotient_kernel: the same table, added on lane 0 (NULL: no lookups)
    uint32_t* out;              // [2][N][4]: chunk k, natural row j
    uint32_t* lde_out;          // optional: the value of row p also goes to lde_out[p * lde_ld + 4 * chunk ...] -- on its own coset a chunk's
    uint64_t lde_ld;            // low-degree extension IS the quotient value, so that half of the chunk LDE needs no transform
};
hipError_t launch_quotient(const QuotientArgs& a, hipStream_t s);
hipError_t launch_logup_addend(const QuotientArgs& a, hipStream_t s);      // needs pairs > 0, perm, addend_out

// quotient values of a constraint PROGRAM (air.h): the interpreter form of launch_quotient for an AIR supplied as data.
constexpr uint32_t AIR_SLOT_EXTRA = 4;      // per-point slots after the two rows: is_first, is_last, is_transition, 1 (then the public values)
// LDS layout of the term-parallel kernels (stark.hip): a group of 8 points = 8 CONSECUTIVE rows of the trace domain on one coset of the
// quotient domain, so the next row of point q is the local row of point q + 1 and 9 rows serve the 8 points.  Columns are kept in groups
// of four: column c of row r (r < 9) at AIR_GP (c >> 2) + 4 r + (c & 3) -- a lane holding four consecutive columns of a row (one 16-byte
// global load) stores them with one 16-byte LDS write, and the pitch of 36 words spreads the lanes of a row over all banks.  The selector
// / constant / public-value slots follow in groups of their own (rows 0..7).  A term record carries, per factor, the LDS word of that
// factor for point 0 (air_lds_base); point q is 4 q words further.
constexpr uint32_t AIR_GP = 36;
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t air_lds_base(uint32_t slot, uint32_t W) {
    if (slot < W) return AIR_GP * (slot >> 2) + (slot & 3u);
    if (slot < 2 * W) { const uint32_t c = slot - W; return AIR_GP * (c >> 2) + (c & 3u) + 4u; }
    const uint32_t x = slot - 2 * W;
    return AIR_GP * ((W >> 2) + (x >> 2)) + (x & 3u);
}
struct QuotientAirArgs {
    const uint32_t* lde; uint64_t ld; uint32_t width; int log_n;
    const uint32_t* xs; const uint32_t* sel_first; const uint32_t* sel_last;
    uint32_t wn_inv;
    int log_qd;                 // log2 of the number of quotient chunks (1 or 2): the quotient domain is the first 2^(log_n + log_qd) rows
    uint32_t inv_zh[4];         // 1 / Z_H on the points e = j mod 2^log_qd
    const uint32_t* body;       // device: program body, Montgomery coefficients (air_device_image)
    uint32_t n_constraints;
    const uint32_t* weights;    // device: [K] extension weights alpha^(K-1-k)
    const uint32_t* pub;        // device: public values, Montgomery
    uint32_t* out;              // [2^log_qd][N][4] natural chunk order, as launch_quotient
    uint32_t* lde_out; uint64_t lde_ld;     // log_qd == 1 only (as launch_quotient)
    // term-parallel form (air_term_records): used when recs != nullptr and the slots of 8 points fit the LDS
    const uint32_t* recs; uint32_t n_terms; uint32_t n_public;
    const uint32_t* addend;     // optional [2^(log_n + log_qd)][4]: added to the folded constraints before the division by Z_H (lookups)
    uint32_t no_chain;          // A/B: 1 keeps the one-group-per-workgroup kernel
    uint32_t wide;              // the records are those of the wide form (air_term_records_wide; stark.hip quotient_air_wide_kernel)
    uint32_t cls[6];            // wide form: records of n factors are cls[n - 1] .. cls[n]
};
// The wide form of the term kernel (a lane per point, 64 points + 1 row per tile, column-major in LDS with the selector / constant /
// public-value slots as further columns): for term-heavy programs (SHA-256 chip, 3 366 records: 5.0 against 5.7 ms per 2^21 points; the
// Poseidon2 chip's 1 008 records are a draw, 2.3 against 2.2, and stay with the 8-point form) whose tile fits the LDS and traces of at
// least 64 rows.  The host decides with this predicate when it builds the records.  air_wide_word: LDS word of a slot for point 0.
constexpr uint32_t AIR_WIDE_POINTS = 64, AIR_WIDE_PITCH = 65;
inline bool air_wide_form(uint32_t width, uint32_t n_terms, int log_n, uint32_t n_public) {
    return n_terms >= 2048 && width % 4 == 0 && (size_t)AIR_WIDE_PITCH * (width + AIR_SLOT_EXTRA + n_public) * 4 <= 160 * 1024 && log_n >= 6;
}
ZK_HD uint32_t air_wide_word(uint32_t slot, uint32_t W) {
    if (slot < W) return AIR_WIDE_PITCH * slot;
    if (slot < 2 * W) return AIR_WIDE_PITCH * (slot - W) + 1u;
    return AIR_WIDE_PITCH * (W + (slot - 2 * W));
}
hipError_t launch_quotient_air(const QuotientAirArgs& a, hipStream_t s);

// out[k][p] = 1 / (x_p - z_k), k < npoints (<= 2), p < count; xw (optional): xw[k][p] = x_p / (x_p - z_k), p < xw_count
hipError_t launch_inv_denominators(const uint32_t* xs, uint64_t count, const Ext& z0, const Ext& z1, int npoints,
                                   uint32_t* out, uint32_t* xw, uint64_t xw_count, hipStream_t s);

struct OpenArgs {
    const uint32_t* mat;        // first `rows` rows of a bit-reversed LDE
    uint64_t ld;
    uint32_t width;
    uint64_t rows;              // N
    const uint32_t* xw;         // [npts][xw_stride] ext: x_q / (x_q - z_k)
    uint64_t xw_stride;
    uint32_t* partial;          // [ceil(rows/2048)][npts][width] ext
    int tx;                     // columns per workgroup (power of two <= 64)
};
// out[k][col] = -scale_k * sum_q mat[q][col] xw_k[q]
size_t open_chunks(uint64_t rows, uint32_t width, uint64_t ld, const uint32_t* mat);   // row chunks (= partial sums per column and point) launch_open will use
hipError_t launch_open(const OpenArgs& a, int npts, const Ext& scale0, const Ext& scale1, uint32_t* out, hipStream_t s);

struct ReducedArgs {
    const uint32_t* tlde; uint64_t t_ld; uint32_t width;
    const uint32_t* qlde; uint64_t q_ld; uint32_t q_width;    // quotient matrix: 8 columns (two chunks) or 16 (four)
    const uint32_t* plde; uint64_t p_ld; uint32_t p_width;   // permutation trace LDE (p_width = 0: none)
    uint64_t rows;              // 2N
    const uint32_t* alpha_pow;  // [max(width, p_width, 8)] ext: alpha^j
    const uint32_t* dinv;       // [2][rows] ext
    Ext y_loc, y_next, y_pl, y_pn, y_q, off_next, off_pl, off_pn, off_q;
    Ext off_loc;                // weight of the first term (alpha^0 unless several matrices share one vector)
    int accumulate;             // 1: out[p] += ..., 0: out[p] = ...
    uint32_t* out;              // [rows] ext
};
// scratch_at: [2][rows] ext workspace for the per-row alpha-dots of the trace / permutation matrices
hipError_t launch_reduced_opening(const ReducedArgs& a, uint32_t* scratch_at, hipStream_t s);

struct PermArgs {
    const uint32_t* trace; uint64_t ld;     // main trace (natural rows), Montgomery
    uint64_t rows; uint32_t pairs;
    Ext gamma, beta;
    uint32_t* out; uint64_t out_ld;         // [rows][4 (pairs + 1)]
};
// block_scratch: ceil(rows / 256) ext values
hipError_t launch_perm_trace(const PermArgs& a, uint32_t* block_scratch, hipStream_t s);

// ---- lookups as data (machine mode, prover.cpp): a chip's interaction table on the device = ni records of 12 words
// {sign (0 send, 1 receive), multiplicity column or 0xFFFFFFFF for the constant 1, bus (Montgomery), values nv, nv columns, padding};
// fingerprint of a tuple d = gamma + bus + sum_t beta^(t+1) v_t; one extension column phi_j per pair of interactions (2j, 2j+1).
constexpr int LOOKUP_REC_WORDS = 12;
struct LookupArgs {
    const uint32_t* table; uint32_t ni, cols;      // device records; cols = ceil(ni / 2)
    Ext gamma, bpow[9];                            // beta^0 .. beta^8
    // round 6: the columns the interactions read are few (26 of the Poseidon2 chip's 384, in 4 of its 24 sixteen-column chunks) and a lane that reads them from
    // ITS row makes 64 scattered requests per load.  cmap (device, LOOKUP_MAX_COLS bytes; nullptr: rows are read where they lie): column -> its place in a staged
    // row of n_used words, 0xFF = not read; chunk_mask: bit c = chunk c (columns 16 c .. 16 c + 15) holds a column that is read.  The kernels then load those
    // chunks of a workgroup's 256 rows with 16-byte loads, 64 bytes of a row per four lanes, into an LDS tile of pitch n_used | 1.
    const uint8_t* cmap; uint32_t n_used, chunk_mask;
};
constexpr uint32_t LOOKUP_MAX_COLS = 512, LOOKUP_STAGE_MAX_USED = 56;   // (256 rows x 57 words = 57 KiB of LDS)
struct MachinePermArgs {
    LookupArgs lk;
    const uint32_t* trace; uint64_t ld; uint64_t rows;
    uint32_t* out; uint64_t out_ld;                // [rows][4 (cols + 1)]
    // a keyed machine's rows are [preprocessed | main]: with pre != nullptr columns [0, pre_w) come from the key's trace (pitch pre_ld) and column c >= pre_w is column
    // c - pre_w of `trace` -- the staged form reads both where they lie instead of a copy of the two side by side (pre_w a multiple of 4; staged launches only)
    const uint32_t* pre; uint64_t pre_ld; uint32_t pre_w;
};
bool lookup_perm_two_sources_ok(const MachinePermArgs& a);     // stark.hip: the launch can take (pre, trace) as they lie
hipError_t launch_perm_trace_machine(const MachinePermArgs& a, uint32_t* block_scratch, hipStream_t s);
// the lookup constraints of one chip on its quotient domain (the first 2N rows of the bit-reversed LDEs), folded with their
// weights: addend[p] (extension) = sum_j w_j (phi_j d_a d_b - (m_a d_b + m_b d_a)) + w_F1 is_first (S - sum phi)
// + w_F2 is_transition (S' - S - sum phi') + w_F3 is_last (S - cumsum); the program kernel adds it before dividing by Z_H
struct MachineQuotArgs {
    LookupArgs lk;
    const uint32_t* lde; uint64_t ld; const uint32_t* perm; uint64_t perm_ld; int log_n;
    int log_qd;                                    // the chip's quotient domain has 2^(log_n + log_qd) points (1, or 2 for a program of degree 4 / 5)
    const uint32_t* xs; const uint32_t* sel_first; const uint32_t* sel_last; uint32_t wn_inv;
    const uint32_t* weights;                       // device: [cols + 3] extension weights, in constraint order
    Ext cumsum;
    uint32_t* addend;                              // [2^log_qd N][4]
};
hipError_t launch_lookup_addend(const MachineQuotArgs& a, hipStream_t s);

hipError_t launch_fri_fold(const uint32_t* in, uint32_t* out, const uint32_t* itw, uint64_t half, const Ext& beta, hipStream_t s);
hipError_t launch_ext_add(uint32_t* dst, const uint32_t* src, uint64_t count, hipStream_t s);
// the same fold with the challenge read from device memory: beta = (*beta_ptr)^(2^squarings)
hipError_t launch_fri_fold_dev(const uint32_t* in, uint32_t* out, const uint32_t* itw, uint64_t half, const uint32_t* beta_ptr,
                               int squarings, hipStream_t s);

// Fiat-Shamir duplex challenger state kept in device memory (same fields as the host Challenger, Montgomery words)
struct DevChallenger {
    uint32_t state[16];
    uint32_t in[8];
    uint32_t out[8];
    int32_t n_in, n_out;
};
// observe the 8 words at `root`, then sample one extension element into beta_out[0..4); also copies the root to root_log
// (one 16-lane cooperative permutation per duplexing; a single wave)
hipError_t launch_fri_challenge(DevChallenger* chal, const uint32_t* root, uint32_t* beta_out, uint32_t* root_log, hipStream_t s);


struct GrindArgs {
    uint32_t state[16];         // Montgomery
    int slot;                   // where the candidate witness goes
    uint32_t mask;              // (1 << bits) - 1
};
// *result = min(*result, smallest hit in [base, base + count))
hipError_t launch_grind(const GrindArgs& a, uint32_t base, uint32_t count, uint32_t* result, hipStream_t s);

struct GatherDesc {
    const uint32_t* src;
    uint32_t dst_off;           // words
    uint32_t nwords;
};
hipError_t launch_gather(const GatherDesc* descs, uint32_t ndesc, uint32_t* dst, hipStream_t s);

// plain copy / fill as kernels (util.hip): what the members of a lock-step batch use instead of hipMemcpyAsync / hipMemsetAsync
hipError_t launch_copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t s);
hipError_t launch_fill_bytes(void* dst, int byte, size_t bytes, hipStream_t s);
hipError_t launch_copy2d(uint32_t* dst, uint64_t dst_ld, const uint32_t* src, uint64_t src_ld, uint32_t width, uint64_t rows, hipStream_t s);
}  // namespace zk
