/*
 * oracle/oracle.h -- API of the CPU restatement (the parity oracle).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/bb.h).  PARITY UNPINNED: every function
 * below restates a PUBLISHED algorithm whose reference implementation lives in a
 * crate that is absent from /root/reference; the reference's own call sites are
 * crates/guest-prover-sp1/src/sp1.rs:113 (setup), :116 (prove), :120 (verify) and
 * crates/guest-prover-r0/src/prover.rs:90 (prove_with_opts).  Pinned crates
 * (reference Cargo.lock line): p3-dft :3903, p3-poseidon2 :4030, p3-symmetric :4044,
 * p3-merkle-tree :4013, p3-challenger :3875, p3-fri :3930, p3-uni-stark :4055,
 * sp1-stark :6172 -- all 0.2.1-succinct / 4.1.4.
 *
 * All field values crossing this API are CANONICAL residues (not Montgomery).
 * Matrices are row-major: element (row r, column c) at m[r * width + c].
 */
#ifndef ORACLE_ORACLE_H
#define ORACLE_ORACLE_H

#include "bb.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- threads (OpenMP) used by the heavy loops; returns the count in effect ---- */
int orc_set_threads(int n);

/* ---- field helpers exported for the Python tests ---- */
uint32_t orc_bb_mul(uint32_t a, uint32_t b);
uint32_t orc_bb_inv(uint32_t a);
uint32_t orc_bb_pow(uint32_t a, uint64_t e);
uint32_t orc_two_adic_generator(int bits);
void orc_bb4_mul(const uint32_t a[4], const uint32_t b[4], uint32_t out[4]);
void orc_bb4_inv(const uint32_t a[4], uint32_t out[4]);
void orc_to_monty(const uint32_t* in, uint32_t* out, size_t n);
void orc_from_monty(const uint32_t* in, uint32_t* out, size_t n);

/* ---- NTT / LDE  (p3-dft: Radix2Dit / Radix2DitParallel) ---- */
/* O(n^2) definition: out[k] = sum_j in[j] w^(jk)  (inverse: w^-1 and 1/n). */
void orc_dft_naive(const uint32_t* in, uint32_t* out, int log_n, size_t width, int inverse);
/* radix-2 NTT over every column, in place, natural order in and out. */
void orc_ntt(uint32_t* a, int log_n, size_t width, int inverse);
/* coset_lde_batch(evals, log_blowup, shift).bit_reverse_rows():
 * out row bitrev(i) = f(shift * w_{N*B}^i), f interpolating `in` on the subgroup. */
void orc_coset_lde(const uint32_t* in, uint32_t* out, int log_n, size_t width,
                   int log_blowup, uint32_t shift);

/* ---- Poseidon2 width 16 (p3-poseidon2) + sponge/compress (p3-symmetric) ---- */
void orc_poseidon2_permute(uint32_t state[16]);
/* PaddingFreeSponge<16,8,8>: overwrite-mode absorb of `n` elements. */
void orc_sponge_hash(const uint32_t* in, size_t n, uint32_t out[8]);
/* TruncatedPermutation<2,8,16>: permute(left || right)[0..8]. */
void orc_compress(const uint32_t left[8], const uint32_t right[8], uint32_t out[8]);
/* the same eight at a time (poseidon2_x8.c: AVX-512 lanes behind a start-up self-check against the scalar functions; scalar otherwise) */
int orc_simd_enabled(void);
void orc_sponge_hash_x8(const uint32_t* const in[8], size_t n, uint32_t* const out[8]);
void orc_compress_x8(const uint32_t* const left[8], const uint32_t* const right[8], uint32_t* const out[8]);

/* ---- width 24 (RISC Zero's Poseidon2 shape) ---- */
void orc_poseidon2_24_permute(uint32_t state[24]);
void orc_sponge24_hash(const uint32_t* in, size_t n, size_t stride, uint32_t out[8]);
void orc_compress24(const uint32_t left[8], const uint32_t right[8], uint32_t out[8]);
void orc_merkle_tree_p24_colmajor(const uint32_t* mat, size_t cols, int log_rows, uint32_t* tree);

/* ---- Merkle commitment (p3-merkle-tree FieldMerkleTreeMmcs, equal heights) ----
 * leaf r = sponge(row r of mats[0] || row r of mats[1] || ...).
 * tree: all levels, level 0 = 2^log_h leaf digests, then 2^(log_h-1) ..., root last;
 *       (2^(log_h+1) - 1) * 8 words. */
void orc_hash_rows(const uint32_t* const* mats, const size_t* widths, int nmats,
                   size_t height, uint32_t* digests);
void orc_merkle_tree(const uint32_t* const* mats, const size_t* widths, int nmats,
                     int log_h, uint32_t* tree);
/* Mixed-height commit (matrices sorted by height, tallest first, heights powers of two):
 * shorter matrices are injected at their level as compress(node, sponge(row)). */
void orc_merkle_tree_mixed(const uint32_t* const* mats, const size_t* widths,
                           const int* log_heights, int nmats, uint32_t* tree);
/* verify an opening of leaf `index`: siblings[log_h][8], rows of all matrices. */
int orc_merkle_verify(const uint32_t root[8], int log_h, size_t index,
                      const uint32_t* const* rows, const size_t* widths, int nmats,
                      const uint32_t* siblings);
/* one row-major matrix, hash selected by hash_width (16: as above; 24: Poseidon2 width 24) */
void orc_merkle_tree_hw(const uint32_t* mat, size_t width, int log_h, uint32_t* tree, int hash_width);
int orc_merkle_verify_hw(const uint32_t root[8], int log_h, size_t index, const uint32_t* row, size_t width,
                         const uint32_t* siblings, int hash_width);

/* ---- Fiat-Shamir duplex challenger (p3-challenger DuplexChallenger<16,8>) ---- */
typedef struct {
    uint32_t state[16];
    uint32_t input[8];
    int n_input;
    uint32_t output[8];
    int n_output;
} orc_challenger_t;
void orc_chal_init(orc_challenger_t* c);
void orc_chal_observe(orc_challenger_t* c, uint32_t v);
void orc_chal_observe_slice(orc_challenger_t* c, const uint32_t* v, size_t n);
uint32_t orc_chal_sample(orc_challenger_t* c);
void orc_chal_sample_ext(orc_challenger_t* c, uint32_t out[4]);
uint32_t orc_chal_sample_bits(orc_challenger_t* c, int bits);
/* smallest witness w (canonical order) with check_witness(bits, w); observes it. */
uint32_t orc_chal_grind(orc_challenger_t* c, int bits);
int orc_chal_check_witness(orc_challenger_t* c, int bits, uint32_t witness);

/* ---- synthetic shard: trace generator + AIR ---- */
/* counter-based uniform field element for (seed, linear index) */
uint32_t orc_synth_value(uint64_t seed, uint64_t index);
/* fill a matrix with orc_synth_value(seed, r*width + c)  (NTT / Merkle workloads) */
void orc_fill_uniform(uint64_t seed, int log_n, size_t width, uint32_t* out);
/* AIR-satisfying trace for shard `shard` (width % 4 == 0) */
void orc_gen_trace(uint64_t seed, uint64_t shard, int log_n, size_t width, uint32_t* out);
/* trace whose odd groups (first `pairs` pairs) receive the even groups' (a, b) under the row
 * permutation 5i+3 mod N: satisfies the LogUp-extended AIR (orc_params_t.logup_pairs = pairs) */
void orc_gen_trace_logup(uint64_t seed, uint64_t shard, int log_n, size_t width, int pairs, uint32_t* out);
/* the same with lookups BETWEEN two tables of equal height: receiver groups read the partner table's sender groups */
void orc_gen_trace_logup_cross(uint64_t seed, uint64_t shard, uint64_t partner_shard, int log_n, size_t width, size_t partner_width,
                               int pairs, uint32_t* out);
/* permutation trace [phi_0 .. phi_{Q-1} | S]: N x 4(Q+1) words */
void orc_perm_trace(const uint32_t* trace, int log_n, size_t width, int pairs,
                    const uint32_t gamma[4], const uint32_t beta[4], uint32_t* out);
void orc_quotient_values_logup(const uint32_t* lde, int log_n, size_t width, const uint32_t* perm_lde, int pairs,
                               const uint32_t gamma[4], const uint32_t beta[4], const uint32_t alpha[4], uint32_t* out);
/* the same with the last-row constraint S = cumsum (tables that look each other up) */
void orc_quotient_values_logup_c(const uint32_t* lde, int log_n, size_t width, const uint32_t* perm_lde, int pairs,
                                 const uint32_t gamma[4], const uint32_t beta[4], const uint32_t alpha[4], const uint32_t cumsum[4], uint32_t* out);
/* number of constraint violations of the synthetic AIR on a trace (0 = valid) */
size_t orc_check_trace(const uint32_t* trace, int log_n, size_t width);

/* ---- STARK stages (sp1-stark / p3-uni-stark / p3-fri), exposed for parity ---- */
typedef struct {
    int log_blowup;       /* 1 (SP1 core) .. 3; 2 = RISC Zero's blowup 4 */
    int num_queries;      /* 100 (SP1 core); 50 (RISC Zero)   */
    int pow_bits;         /* 16 (SP1 core); 0 (RISC Zero)     */
    int logup_pairs;      /* 0: no lookup argument; Q > 0: Q LogUp sender/receiver group pairs */
    /* FRI / hash shape; 0 = the SP1 default in each field (fold by 2, constant final polynomial, width 16) */
    int log_fold;         /* committed FRI layers fold by 2^log_fold: 1 (p3-fri) or e.g. 4 (RISC Zero folds by 16) */
    int log_final;        /* stop folding at a polynomial of < 2^log_final coefficients, sent in clear (RISC Zero: 8) */
    int hash_width;       /* Poseidon2 width of every Merkle tree: 16 (rate 8) or 24 (rate 16, RISC Zero) */
    int code_width;       /* 0: one trace commitment.  Wc > 0 (multiple of 4, < width): RISC Zero's group order -- the first Wc columns
                           * ("code") and the rest ("data") are committed as two trees, code root first; with lookups the permutation
                           * trace is the third ("accum") group, then the quotient ("check") */
} orc_params_t;

/* quotient values on the LDE coset, in bit-reversed row order like the LDE:
 * out[p] (4 words) for p in [0, 2^(log_n+1)).  lde: bit-reversed rows, blowup 2. */
void orc_quotient_values(const uint32_t* lde, int log_n, size_t width,
                         const uint32_t alpha[4], uint32_t* out);
/* open every column of a committed LDE (bit-reversed rows, 2^(log_n+log_blowup) rows)
 * at ext point z by barycentric interpolation on its low coset: out[width][4]. */
void orc_open_at(const uint32_t* lde, int log_n, size_t width, const uint32_t z[4],
                 uint32_t* out);
/* one FRI fold (arity 2) of `in` (2^log_h ext elements, bit-reversed) -> 2^(log_h-1). */
void orc_fri_fold(const uint32_t* in, int log_h, const uint32_t beta[4], uint32_t* out);

/* fold of arity 2^log_arity (<= 64), definition by interpolation; out has 2^(log_h-log_arity) entries */
void orc_fri_fold_k(const uint32_t* in, int log_h, int log_arity, const uint32_t beta[4], uint32_t* out);

/* full shard proof; returns bytes written (0 on error); proof layout: DESIGN.md. */
size_t orc_proof_size(int log_n, size_t width, const orc_params_t* prm, size_t n_public);
size_t orc_prove_shard(const uint32_t* trace, int log_n, size_t width,
                       const uint32_t* public_values, size_t n_public,
                       const orc_params_t* prm, uint8_t* proof, size_t cap);
/* 0 = accept; nonzero = reject code (which check failed). */
int orc_verify_shard(const uint8_t* proof, size_t len, int log_n, size_t width,
                     const uint32_t* public_values, size_t n_public,
                     const orc_params_t* prm);

/* ---- a shard of several chips with different heights (oracle/chips.c): tallest first, heights log_ns[c] in [5, 20],
 * at most 8 chips per height (32 in all), SP1 FRI shape (log_fold 1, log_final 0, hash width 16) at any log_blowup ---- */
/* pairs (may be NULL): LogUp pairs per chip; the permutation traces form a third tree.  partners (may be NULL):
 * partners[c] = -1: chip c's lookups stay inside the chip (orc_gen_trace_logup); d >= 0: its receiver groups hold chip d's
 * sender groups (orc_gen_trace_logup_cross; mutual, equal heights and pair counts) -- then every chip with pairs exposes the
 * final value of its running sum and the verifier checks that they add up to zero (sp1-stark's local cumulative sums) */
/* chips with their own constraint programs (progs[c] NULL: the synthetic AIR); degree <= 5 (4 / 5: four quotient chunks for that chip,
 * log_blowup >= 2); no lookups (proof version 9) */
size_t orc_chips_proof_size_air(const int* log_ns, const size_t* widths, const uint32_t* const* progs, const size_t* prog_words, int n_chips,
                                const orc_params_t* prm, size_t n_public);
size_t orc_prove_chips_air(const uint32_t* const* traces, const int* log_ns, const size_t* widths, const uint32_t* const* progs,
                           const size_t* prog_words, int n_chips, const uint32_t* public_values, size_t n_public, const orc_params_t* prm,
                           uint8_t* proof, size_t cap);
int orc_verify_chips_air(const uint8_t* proof, size_t len, const int* log_ns, const size_t* widths, const uint32_t* const* progs,
                         const size_t* prog_words, int n_chips, const uint32_t* public_values, size_t n_public, const orc_params_t* prm);
/* the machine: chips with programs AND interaction tables (lookups as data, multiplicities, buses; format in chips.c); version 10 */
size_t orc_machine_proof_size(const int* log_ns, const size_t* widths, const uint32_t* const* progs, const size_t* prog_words,
                              const uint32_t* const* tables, const size_t* table_words, int n_chips, const orc_params_t* prm, size_t n_public);
size_t orc_prove_machine(const uint32_t* const* traces, const int* log_ns, const size_t* widths, const uint32_t* const* progs,
                         const size_t* prog_words, const uint32_t* const* tables, const size_t* table_words, int n_chips,
                         const uint32_t* public_values, size_t n_public, const orc_params_t* prm, uint8_t* proof, size_t cap);
int orc_verify_machine(const uint8_t* proof, size_t len, const int* log_ns, const size_t* widths, const uint32_t* const* progs,
                       const size_t* prog_words, const uint32_t* const* tables, const size_t* table_words, int n_chips,
                       const uint32_t* public_values, size_t n_public, const orc_params_t* prm);
/* the keyed machine (proof version 11): chip c has pre_widths[c] PREPROCESSED columns (0: none), committed once by setup -- sp1-stark's
 * StarkMachine::setup, which the reference calls at crates/guest-prover-sp1/src/sp1.rs:113; the root is the verifying key's commitment.
 * Programs and interaction tables address the combined row [preprocessed | main].  orc_machine_setup returns 0 and the root. */
int orc_machine_setup(const uint32_t* const* pre_traces, const int* log_ns, const size_t* pre_widths, int n_chips, const orc_params_t* prm, uint32_t root[8]);
size_t orc_machine_proof_size_keyed(const int* log_ns, const size_t* widths, const size_t* pre_widths, const uint32_t* const* progs, const size_t* prog_words,
                                    const uint32_t* const* tables, const size_t* table_words, int n_chips, const orc_params_t* prm, size_t n_public);
size_t orc_prove_machine_keyed(const uint32_t* const* traces, const uint32_t* const* pre_traces, const int* log_ns, const size_t* widths, const size_t* pre_widths,
                               const uint32_t* const* progs, const size_t* prog_words, const uint32_t* const* tables, const size_t* table_words, int n_chips,
                               const uint32_t* public_values, size_t n_public, const orc_params_t* prm, uint8_t* proof, size_t cap);
int orc_verify_machine_keyed(const uint8_t* proof, size_t len, const int* log_ns, const size_t* widths, const size_t* pre_widths, const uint32_t pre_root[8],
                             const uint32_t* const* progs, const size_t* prog_words, const uint32_t* const* tables, const size_t* table_words, int n_chips,
                             const uint32_t* public_values, size_t n_public, const orc_params_t* prm);
size_t orc_chips_proof_size(const int* log_ns, const size_t* widths, const int* pairs, const int* partners, int n_chips, const orc_params_t* prm, size_t n_public);
size_t orc_prove_chips(const uint32_t* const* traces, const int* log_ns, const size_t* widths, const int* pairs, const int* partners, int n_chips,
                       const uint32_t* public_values, size_t n_public, const orc_params_t* prm, uint8_t* proof, size_t cap);
int orc_verify_chips(const uint8_t* proof, size_t len, const int* log_ns, const size_t* widths, const int* pairs, const int* partners, int n_chips,
                     const uint32_t* public_values, size_t n_public, const orc_params_t* prm);

/* intermediates of the last orc_prove_shard call in this thread (for parity tests) */
typedef struct {
    uint32_t trace_root[8];
    uint32_t quotient_root[8];
    uint32_t alpha[4];
    uint32_t zeta[4];
    uint32_t fri_alpha[4];
    uint32_t pow_witness;
} orc_prove_debug_t;
void orc_last_prove_debug(orc_prove_debug_t* out);

/* ---- RISC Zero Hal operators (oracle/hal.c; risc0-zkp 1.2.5 hal::Hal, reference Cargo.lock:5057): column-major
 * [count][size] vectors, extension elements over x^4 = ext_w (11, or p - 11 for RISC Zero's x^4 + 11) ---- */
void orc_hal_ext_mul(const uint32_t a[4], const uint32_t b[4], uint32_t ext_w, uint32_t out[4]);
void orc_hal_eltwise_add(uint32_t* out, const uint32_t* a, const uint32_t* b, size_t n);
void orc_hal_eltwise_sum_ext(uint32_t* out, const uint32_t* in, size_t count, size_t to_add);
void orc_hal_eltwise_zeroize(uint32_t* io, size_t n);
void orc_hal_zk_shift(uint32_t* io, size_t count, int log_size, uint32_t shift);
void orc_hal_mix_poly_coeffs(uint32_t* out, const uint32_t mix_start[4], const uint32_t mix[4], const uint32_t* in,
                             const uint32_t* combos, size_t input_size, size_t count, uint32_t ext_w);
void orc_hal_batch_evaluate_any(const uint32_t* coeffs, int log_size, const uint32_t* which, const uint32_t* xs, uint32_t* out,
                                size_t eval_count, uint32_t ext_w);
void orc_hal_gather_sample(uint32_t* dst, const uint32_t* src, size_t idx, size_t size, size_t stride);
void orc_hal_scatter(uint32_t* into, const uint32_t* index, const uint32_t* offsets, const uint32_t* values, size_t rows);
void orc_hal_prefix_products_ext(uint32_t* io, size_t n, uint32_t ext_w);
void orc_hal_hash_rows_sha256(const uint32_t* mat, size_t cols, size_t rows, uint32_t* digests);
void orc_hal_hash_fold_sha256(const uint32_t* children, uint32_t* parents, size_t count);

/* ---- constraint programs: the AIR as data (oracle/air.c; format in its header comment) ---- */
int orc_air_validate(const uint32_t* prog, size_t words, size_t width, size_t n_public);
void orc_air_digest(const uint32_t* prog, size_t words, uint32_t out[8]);
size_t orc_air_synthetic(size_t width, size_t n_public, uint32_t* out, size_t cap);
int orc_air_log_quotient_degree(const uint32_t* prog);
void orc_quotient_values_air(const uint32_t* prog, const uint32_t* lde, int log_n, size_t width, const uint32_t* pub,
                             const uint32_t alpha[4], int log_quotient_degree, uint32_t* out);
size_t orc_proof_size_air(int log_n, size_t width, const orc_params_t* prm, size_t n_public, int log_quotient_degree);
size_t orc_prove_shard_air(const uint32_t* prog, size_t prog_words, const uint32_t* trace, int log_n, size_t width,
                           const uint32_t* public_values, size_t n_public, const orc_params_t* prm, uint8_t* proof_bytes, size_t cap);
int orc_verify_shard_air(const uint32_t* prog, size_t prog_words, const uint8_t* proof_bytes, size_t len, int log_n, size_t width,
                         const uint32_t* public_values, size_t n_public, const orc_params_t* prm);

#ifdef __cplusplus
}
#endif
#endif
