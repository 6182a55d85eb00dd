/*
 * zkhip_hal.h -- the RISC Zero `Hal` operator set of libzkhip.so: column-major interpolate / expand, the width-24 Poseidon2 and
 * SHA-256 hash suites, and the eltwise / zk_shift / mix_poly_coeffs / batch_evaluate_any / gather / scatter / prefix-product operators, each citing the
 * risc0-zkp 1.2.5 `hal::Hal` method it stands in for (reference Cargo.lock:5057; call site crates/guest-prover-r0/src/prover.rs:90).
 * The context, status codes and conventions are include/zkhip.h's.
 */
#ifndef ZKHIP_HAL_H
#define ZKHIP_HAL_H
#include "zkhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* RISC Zero layout (risc0-zkp Hal::hash_rows + hash_fold, reference
 * Cargo.lock:5057, call site crates/guest-prover-r0/src/prover.rs:90): d_mat is COLUMN-major
 * [cols][2^log_rows]; leaf r = Poseidon2-width-24 sponge (rate 16) over row r; parents =
 * permute(l || r || 0^8)[0..8].  Own constants "zktls-amd/p2-bb24-v1". */
int zkhip_merkle_commit_p24_colmajor(zkhip_ctx* ctx, const uint32_t* d_mat, uint32_t cols, int log_rows,
                                     uint32_t* d_tree);
/* Hal::batch_interpolate_ntt: `count` polynomials, column-major [count][2^log_size]; evaluations
 * in BIT-REVERSED order in, coefficients in natural order out (scaled by 1/size). */
int zkhip_batch_interpolate_colmajor(zkhip_ctx* ctx, const uint32_t* d_evals, uint32_t* d_coeffs,
                                     uint32_t count, int log_size);
/* Hal::zk_shift + batch_expand_into_evaluate_ntt: coefficients (natural) -> evaluations of the same
 * polynomials on shift * <w_(size * 2^log_blowup)>, bit-reversed, [count][size << log_blowup].
 * RISC Zero expands by 4 (log_blowup = 2).  `shift` canonical. */
int zkhip_batch_expand_colmajor(zkhip_ctx* ctx, const uint32_t* d_coeffs, uint32_t* d_evals,
                                uint32_t count, int log_size, int log_blowup, uint32_t shift);

/* ---- RISC Zero `Hal` operator set (risc0-zkp 1.2.5 trait hal::Hal, reference Cargo.lock:5057; kernels risc0-sys 1.2.5,
 * Cargo.lock:5045; call site crates/guest-prover-r0/src/prover.rs:90).  Data as the Hal
 * holds it: polynomials / columns are contiguous device vectors (column-major [count][size]) of Montgomery words, extension
 * elements are 4 consecutive words, 16-byte aligned.  `ext_field` selects the extension the operator multiplies in:
 * ZKHIP_EXT_X4_MINUS_11 = F_p[x]/(x^4 - 11) (Plonky3 / SP1, what the shard prover uses) or ZKHIP_EXT_X4_PLUS_11 =
 * F_p[x]/(x^4 + 11) (RISC Zero).  Extension challenges passed by value (mix, mix_start) are HOST pointers, Montgomery form. ---- */
typedef enum { ZKHIP_EXT_X4_MINUS_11 = 0, ZKHIP_EXT_X4_PLUS_11 = 1 } zkhip_ext_field;
/* Hal::eltwise_add_elem: out[i] = a[i] + b[i] */
int zkhip_eltwise_add(zkhip_ctx* ctx, uint32_t* d_out, const uint32_t* d_a, const uint32_t* d_b, size_t n);
/* Hal::eltwise_copy_elem */
int zkhip_eltwise_copy(zkhip_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t n);
/* Hal::eltwise_zeroize_elem: cells still holding the "unset" marker 0xffffffff become 0 */
int zkhip_eltwise_zeroize(zkhip_ctx* ctx, uint32_t* d_io, size_t n);
/* Hal::eltwise_sum_extelem: out[i] = sum_j in[j * count + i] over extension elements (i < count, j < to_add) */
int zkhip_eltwise_sum_ext(zkhip_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t count, size_t to_add);
/* Hal::zk_shift: coefficient i of each of `count` polynomials of 2^log_size coefficients times shift^i (RISC Zero shifts
 * by 3); `shift` canonical.  Any 4-byte-aligned d_io (a slice inside a larger buffer: 16-byte alignment only selects the faster form), any count */
int zkhip_zk_shift(zkhip_ctx* ctx, uint32_t* d_io, size_t count, int log_size, uint32_t shift);
/* Hal::mix_poly_coeffs: d_out[combos[i] * count + idx] += mix_start * mix^i * d_in[i * count + idx], i < input_size, idx < count;
 * d_out holds extension elements ([n_combos][count]), d_in base elements ([input_size][count]), d_combos device u32 */
int zkhip_mix_poly_coeffs(zkhip_ctx* ctx, uint32_t* d_out, const uint32_t mix_start[4], const uint32_t mix[4], const uint32_t* d_in,
                          const uint32_t* d_combos, size_t input_size, size_t count, int ext_field);
/* Hal::batch_evaluate_any: d_out[e] = polynomial d_which[e] (2^log_size base coefficients, lowest first, polynomial p at
 * d_coeffs + p * 2^log_size) evaluated at the extension point d_xs[e] */
int zkhip_batch_evaluate_any(zkhip_ctx* ctx, const uint32_t* d_coeffs, int log_size, const uint32_t* d_which, const uint32_t* d_xs,
                             uint32_t* d_out, size_t eval_count, int ext_field);
/* Hal::gather_sample: d_dst[g] = d_src[g * stride + idx], g < size (row idx of a column-major matrix: a FRI query row) */
int zkhip_gather_sample(zkhip_ctx* ctx, uint32_t* d_dst, const uint32_t* d_src, size_t idx, size_t size, size_t stride);
/* Hal::scatter: d_into[d_offsets[k]] = d_values[k] for k in [d_index[r], d_index[r + 1]), r < rows */
int zkhip_scatter(zkhip_ctx* ctx, uint32_t* d_into, const uint32_t* d_index, const uint32_t* d_offsets, const uint32_t* d_values, size_t rows);
/* Hal::prefix_products: inclusive prefix products of n extension elements, in place (the accumulator columns) */
int zkhip_prefix_products_ext(zkhip_ctx* ctx, uint32_t* d_io, size_t n, int ext_field);
/* Hal::hash_rows / hash_fold with the SHA-256 hash suite: leaf r = SHA-256 over the CANONICAL words of row r of the
 * column-major [cols][rows] matrix, each word serialised big-endian, FIPS 180-4 padding; a node = SHA-256 of its children's
 * 64 bytes.  Digests are the eight 32-bit state words (plain integers, not field elements).  This byte convention is this
 * library's own (stated in DESIGN.md 4.4); the Poseidon2 variants are zkhip_merkle_commit_p24_colmajor / zkhip_merkle_commit. */
int zkhip_hash_rows_sha256(zkhip_ctx* ctx, const uint32_t* d_mat, size_t cols, size_t rows, uint32_t* d_digests);
int zkhip_hash_fold_sha256(zkhip_ctx* ctx, const uint32_t* d_children, uint32_t* d_parents, size_t count);
int zkhip_merkle_commit_sha256_colmajor(zkhip_ctx* ctx, const uint32_t* d_mat, uint32_t cols, int log_rows, uint32_t* d_tree);

#ifdef __cplusplus
}
#endif
#endif /* ZKHIP_HAL_H */
