/*
 * zkhip_chips.h -- the CHIP LEVEL of libzkhip.so: what include/zkhip.h's provers are made of, for callers that build machines of their own or look inside.
 *   - chip programs (constraint programs as data: the synthetic AIR, the SHA-256 compression chip, the Poseidon2 permutation chip, the FRI-fold chip
 *     and its variants) and their trace generators on the device;
 *   - the descriptions of the recursion machines (programs, interaction tables, preprocessed traces per chip) that a verifier derives keys from, and
 *     the host-side table hook the tests compare the device's witness kernels with;
 *   - the Poseidon2 chip's Merkle-path prover and the FRI-only recursion mode (zkhip_prove_fri_indices[_batch]: the cheap mode whose verifier reads
 *     the inner proof; the whole-verifier machines are zkhip.h's zkhip_prove_shard_verifier / zkhip_prove_machine_verifier).  Round 6 removed the
 *     three earlier FRI-only generations (zkhip_prove_fri_queries / _layers / _transcript with their keys, sizes, trace generators and verifiers):
 *     their chips live on as the FOLD, SAMPLES and Poseidon2 chips of the shard verifier machines, their programs and views below;
 *   - diagnostics and self-tests.
 * Each entry names the upstream structure it stands in for (sp1-core-machine / sp1-recursion chips, reference Cargo.lock:5822, 6047, 6172).
 */
#ifndef ZKHIP_CHIPS_H
#define ZKHIP_CHIPS_H
#include "zkhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A second real chip: the width-16 Poseidon2 permutation with Merkle-path chaining -- what a recursion machine (a STARK verifier proven
 * inside a STARK: the compress / shrink / wrap stages behind SP1ProofMode::Groth16, crates/guest-prover-sp1/src/sp1.rs:116; sp1-recursion's
 * Poseidon2 chips, reference Cargo.lock:6172 ff.) spends its rows on.  One row = one permutation of the parameter set in effect, every
 * intermediate in a column (ZKHIP_P2CHIP_WIDTH = 360 columns, degree <= 3); flag columns chain rows into Merkle paths (a row's
 * digest-carrying input half = the previous row's digest), into LEAF HASHES (the overwrite-mode sponge over an opened row: a row's capacity
 * half = the previous row's) and count the paths that end in the public root.  Public values: root[8], count.  zkhip_p2chip_air writes the constraint program (returns its size in words; the program follows the Poseidon2 tables, so reload
 * it after zkhip_load_poseidon2_params).  zkhip_p2chip_gen_merkle_trace fills a device trace of 2^log_n rows from host arrays: with
 * row_width = 0 path p = `depth` rows and leaves[p][8] is its leaf digest; with row_width = 8 k, leaves[p][row_width] is the OPENED ROW and the
 * path starts with k sponge rows that hash it (a whole opening of a commitment: what a verifier checks per query and matrix);
 * siblings[p][l][8] the sibling at level l, bit l of indices[p] = "the node is a
 * right child at level l" (canonical words); roots[p][8] receives where each path ends.  zkhip_prove_merkle_paths = trace + proof of
 * "I know n_paths Merkle paths that end in root" (refuses paths that do not); zkhip_verify_merkle_paths checks one (the trace height is
 * read from the proof).  The proofs are zkhip_prove_shard_air proofs (version 7). */
#define ZKHIP_P2CHIP_WIDTH 360
size_t zkhip_p2chip_air(uint32_t* program, size_t cap_words);
int zkhip_p2chip_gen_merkle_trace(zkhip_ctx* ctx, const uint32_t* leaves, uint32_t row_width, const uint32_t* siblings, const uint32_t* indices, size_t n_paths,
                                  int depth, int log_n, uint32_t* d_trace, size_t ld, uint32_t* roots);
size_t zkhip_merkle_paths_proof_size(size_t n_paths, int depth, uint32_t row_width, const zkhip_params* prm);
int zkhip_prove_merkle_paths(zkhip_ctx* ctx, const uint32_t* leaves, uint32_t row_width, const uint32_t* siblings, const uint32_t* indices, size_t n_paths, int depth,
                             const uint32_t root[8], const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len);
int zkhip_verify_merkle_paths(const uint8_t* proof, size_t len, const uint32_t root[8], size_t n_paths, const zkhip_params* prm, int* reason);

/* ---- a first step of recursion: the FRI part of a shard proof checked INSIDE a proof (second half).  The reference's
 * hot call is client.prove(.., SP1ProofMode::Groth16) (crates/guest-prover-sp1/src/sp1.rs:116): core -> compress -> shrink -> wrap, and
 * compress verifies shard proofs in-circuit (sp1-recursion, reference Cargo.lock:6172 ff.; RISC Zero lift -> join, prover.rs:90).
 * zkhip_fri_view_shard runs the verifier of a zkhip_prove_shard proof (fold by 2, constant final value: the SP1 shape) and hands out what
 * its FRI check reads: layers = log_n folding challenges (4 words each), the final value, and per query the index (layers + 1 bits), the
 * reduced opening it starts from and one sibling per layer -- canonical words; fails like zkhip_verify_shard if the proof is rejected.
 * The FRI-fold chip (fri_chip.hip; 32 + layers columns rounded up to a multiple of 4, one row per (query, layer), degree 3) folds these
 * chains; its rows send the layer pairs on two lookup buses to a PREPROCESSED table that lists every distinct pair of the view with the
 * number of queries reading it -- fixed multiplicities, so every listed pair is folded exactly as often as the inner proof reads it.
 * zkhip_fri_queries_key commits that table (zkhip_machine_setup): vk is what a verifier recomputes from the inner proof; final_value is
 * what the chains end in.  zkhip_prove_fri_queries generates the chip's trace on the device and proves the two-chip keyed machine
 * (proof version 11; public values: the challenges, then the final value); zkhip_verify_fri_queries checks it on the host.
 * NOT in-circuit yet: the Merkle paths of the pairs (the Poseidon2 chip above proves such paths, it is not on this bus yet), the reduced
 * openings, the transcript.  zkhip_fri_chip_air writes the chip's constraint program (its size in words), zkhip_fri_chip_gen_trace the
 * trace alone (finals: [n_queries][4], the value each chain ends in). */
int zkhip_fri_view_shard(const uint8_t* proof, size_t len, int log_n, uint32_t width, const uint32_t* public_values, size_t n_public,
                         const zkhip_params* prm, uint32_t* betas, uint32_t final_value[4], uint32_t* indices, uint32_t* values, uint32_t* siblings);
uint32_t zkhip_fri_chip_width(int layers);
size_t zkhip_fri_chip_air(int layers, uint32_t* program, size_t cap_words);
/* The same with the Merkle paths of the pairs IN-CIRCUIT (blowup-2 proofs): the FRI-fold chip wired by lookups to the Poseidon2 chip.
 * zkhip_fri_view_shard_paths also hands out the layer roots ([layers][8]) and, per query, the layers' authentication paths one after the
 * other (8 (layers - l) words for layer l; zkhip_fri_view_path_words(layers) words per query).  The machine has four chips: the Poseidon2
 * chip's FRI-layers variant (zkhip_p2chip_air_fri_layers: one path per (query, layer) -- a leaf row hashing the pair, then the compression
 * rows up to the layer's root; leaf rows receive the pairs from the bus, END rows send (layer, root) to the ROOTS table), the fold chip
 * in its wired form (zkhip_fri_layers_chip_air: sends the pairs, and (index, reduced opening) on a query's first row), and two
 * PREPROCESSED tables: QUERIES (index, reduced opening) and ROOTS (layer, root).  The key (zkhip_fri_layers_key) therefore holds no FRI
 * layer value any more: a verifier needs the layer roots of the inner proof and the reduced openings it computes itself.  Statement: "for
 * the layer commitments and the (index, reduced opening) pairs in the key, every query's chain opens the commitments layer by layer and
 * folds, under the public challenges, to the public final value."  Still outside: the trace / quotient openings, the reduced openings,
 * the transcript. */
size_t zkhip_fri_view_path_words(int layers);
int zkhip_fri_view_shard_paths(const uint8_t* proof, size_t len, int log_n, uint32_t width, const uint32_t* public_values, size_t n_public,
                               const zkhip_params* prm, uint32_t* betas, uint32_t final_value[4], uint32_t* indices, uint32_t* values, uint32_t* siblings,
                               uint32_t* roots, uint32_t* paths);
/* the Fiat-Shamir side of the view: the layer roots (8 words each), the challenges they lead to (4 words each), and the duplex
 * challenger as the commit phase finds it -- transcript[0..8) = the capacity half of its state, transcript[8] = pending inputs (0);
 * transcript[9] = the proof-of-work witness the query phase absorbs behind the final value.
 * With these every challenge is one step of a sponge chain over the roots: state <- (root_l | capacity), permute,
 * beta_l = (state[7], state[6], state[5], state[4]), capacity <- state[8..16) -- what a transcript chip has to prove next
 * (docs/RECURSION_NEXT.md; p3-challenger DuplexChallenger, reference Cargo.lock:3875).  Canonical words; host only. */
int zkhip_fri_view_transcript(const uint8_t* proof, size_t len, int log_n, uint32_t width, const uint32_t* public_values, size_t n_public,
                              const zkhip_params* prm, uint32_t* roots, uint32_t* betas, uint32_t transcript[10]);
/* ... and both in ONE pass over the proof (what zkhip_prove_fri_indices_batch runs per shard proof) */
int zkhip_fri_view_all(const uint8_t* proof, size_t len, int log_n, uint32_t width, const uint32_t* public_values, size_t n_public,
                       const zkhip_params* prm, uint32_t* betas, uint32_t final_value[4], uint32_t* indices, uint32_t* values, uint32_t* siblings,
                       uint32_t* roots, uint32_t* paths, uint32_t transcript[10]);
size_t zkhip_fri_layers_chip_air(int layers, uint32_t* program, size_t cap_words);
size_t zkhip_p2chip_air_fri_layers(int layers, uint32_t* program, size_t cap_words);
/* The same machine with the FRI TRANSCRIPT in-circuit: the Poseidon2 chip's trace starts with transcript rows (zkhip_p2chip_air_fri_transcript,
 * 364 columns) -- a sponge chain over the layer roots from the duplex challenger's capacity (zkhip_fri_view_transcript): row l absorbs
 * root_l (sent to the ROOTS table like a path's end), keeps the capacity of row l - 1 (row 0: public) and sends
 * (l, out[7], out[6], out[5], out[4]) on a bus of its own.  The ROOTS table holds the challenges in its MAIN columns (the prover's),
 * receives each once from its transcript row and hands it to the layer's fold rows (zkhip_fri_transcript_chip_air: the fold chip without
 * public challenges).  Public values: the final value and the capacity.  NEITHER THE KEY NOR THE VERIFIER HOLDS A CHALLENGE -- statement:
 * "for the layer commitments and the (index, reduced opening) pairs in the key, every query's chain opens the commitments and folds to the
 * public final value under the challenges the transcript derives from these commitments, starting from this challenger state."  The
 * prover is still handed the view's challenges and refuses when its chain disagrees.  Still outside: how the challenger state came about
 * (the transcript before the commit phase), the query indices, the trace / quotient openings and the reduced openings.  Ref: p3-challenger
 * DuplexChallenger (reference Cargo.lock:3875) behind sp1.rs:116. */
size_t zkhip_fri_transcript_chip_air(int layers, uint32_t* program, size_t cap_words);
size_t zkhip_p2chip_air_fri_transcript(int layers, uint32_t* program, size_t cap_words);
/* The QUERY PHASE of the transcript in-circuit (zkhip_prove_fri_indices): the sponge chain of the transcript machine goes on as the inner
 * proof's verifier does (p3-fri verifier: observe the final polynomial, check the proof-of-work witness, sample the query indices;
 * reference Cargo.lock:3930, 3875) -- one row absorbs the final value and the witness over the front of the rate, further rows only
 * permute; a fifth chip, SAMPLES, takes the 31 bits of every word these rows hand out (canonical decomposition): the first word's low
 * inner_pow_bits bits must be zero, the low layers + 1 bits of the others are the query indices, which reach the QUERIES table's MAIN
 * column by query number and from there the first fold row of the query.  The key holds (query number, reduced opening) and the layer
 * roots -- no index; the verifier is handed the final value and the challenger's capacity: "every query, AT THE INDEX THE TRANSCRIPT
 * DRAWS FOR IT, opens these commitments and folds to this final value, and the transcript's proof of work holds."  inner_pow_bits = the
 * grinding bits of the INNER proof (zkhip_params.pow_bits of the proof the view was taken from); witness = its proof-of-work witness
 * (zkhip_fri_view_transcript: transcript[9]).  zkhip_fri_indices_program: the two programs that differ from the transcript machine's
 * (which = 0: the Poseidon2 chip with query-phase rows, 1: the SAMPLES chip).  Still outside: the transcript before the commit phase,
 * the trace / quotient openings and the reduced openings. */
size_t zkhip_fri_indices_program(int which, int layers, int inner_pow_bits, uint32_t* program, size_t cap_words);
int zkhip_fri_indices_key(zkhip_ctx* ctx, int layers, size_t n_queries, int inner_pow_bits, const uint32_t* values, const uint32_t* roots,
                          const zkhip_params* prm, zkhip_machine_key** key, uint32_t vk[8]);
size_t zkhip_fri_indices_proof_size(int layers, size_t n_queries, int inner_pow_bits, const zkhip_params* prm);
int zkhip_prove_fri_indices(zkhip_ctx* ctx, const zkhip_machine_key* key, int layers, size_t n_queries, int inner_pow_bits, const uint32_t* betas,
                            const uint32_t* indices, const uint32_t* values, const uint32_t* siblings, const uint32_t* roots, const uint32_t* paths,
                            const uint32_t capacity[8], uint32_t witness, const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len);
int zkhip_verify_fri_indices(const uint8_t* proof, size_t len, int layers, size_t n_queries, int inner_pow_bits, const uint32_t final_value[4],
                             const uint32_t capacity[8], const uint32_t vk[8], const zkhip_params* prm, int* reason);
/* Many shard proofs in one call -- the compress-like step of the path (sp1.rs:116: core -> COMPRESS verifies the shard proofs; prover.rs:90:
 * lift): per job the FRI view of the shard proof (host), the key of its query-phase machine and the machine's proof; jobs are dealt over
 * `devices` (NULL / 0: every visible device) like every batch of this library -- lock-step lanes for the launch-bound sizes
 * (zkhip_set_lockstep), otherwise `in_flight_per_device` contexts per device.  All jobs share (log_n, width, inner).  Out per job: the
 * proof, and what zkhip_verify_fri_indices takes beside it (vk, final value, capacity).  verify != 0: every proof is checked on the host
 * right after it was made (sp1.rs:120).  Returns the status of the lowest failing job (every job still gets its own). */
typedef struct zkhip_fri_job {
    const uint8_t* shard_proof; size_t shard_proof_len;     /* in: a shard proof of this library (fold by 2, blowup 2, constant final value) */
    const uint32_t* public_values; size_t n_public;
    uint8_t* proof; size_t proof_cap;                       /* in: >= zkhip_fri_indices_proof_size(log_n, inner->num_queries, inner->pow_bits, outer) */
    size_t proof_len;                                       /* out */
    uint32_t vk[8], final_value[4], capacity[8];            /* out */
    int status;                                             /* out */
} zkhip_fri_job;
int zkhip_prove_fri_indices_batch(const int* devices, int n_devices, zkhip_fri_job* jobs, int n_jobs, int log_n, uint32_t width,
                                  const zkhip_params* inner, const zkhip_params* outer, int in_flight_per_device, int verify);

/* ---- chip programs, trace generators and machine descriptions whose statement-level entries are in zkhip.h (documented there, beside the prover that
 * uses them: the AIR-as-data section, the SHA-256 chip, the keyed SHA-256 machine, the shard verifier machines) ---- */
int zkhip_air_synthetic(uint32_t width, size_t n_public, uint32_t* out, size_t cap, size_t* words);
void zkhip_sha256_padding_publics(uint64_t message_len, uint64_t first_block, uint64_t n_active, uint32_t out[75]);
size_t zkhip_sha256_air(uint32_t* program, size_t cap_words);
size_t zkhip_sha256_pad(const uint8_t* message, size_t len, uint8_t* blocks, size_t cap);
int zkhip_sha256_gen_trace(zkhip_ctx* ctx, const uint8_t* blocks, size_t n_active, size_t n_blocks, uint64_t message_len, uint32_t* d_trace, size_t ld,
                           uint32_t publics[91]);
int zkhip_range_table(zkhip_ctx* ctx, const uint32_t* d_trace, size_t ld, size_t rows, const uint32_t* columns, int n_columns, int log_table,
                      uint32_t* d_table, size_t table_ld, uint32_t value_col, uint32_t mult_col);
size_t zkhip_sha256_air_chained(uint32_t* program, size_t cap_words);
int zkhip_sha256_gen_trace_chained(zkhip_ctx* ctx, const uint32_t chain_in[8], const uint8_t* blocks, size_t n_active, size_t n_blocks, uint64_t message_len,
                                   uint64_t first_block, uint32_t* d_trace, size_t ld, uint32_t publics[91]);
size_t zkhip_sha256_machine_describe(size_t message_len, int which, int kind, uint32_t* out, size_t cap, int* log_n, uint32_t* width, uint32_t* pre_width);
size_t zkhip_shard_verifier_describe(int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, size_t n_proofs, int which, int kind, uint32_t* out,
                                     size_t cap_words, int* log_rows, uint32_t* main_width, uint32_t* pre_width);
size_t zkhip_shard_verifier_describe_air(const uint32_t* program, size_t program_words, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public,
                                         size_t n_proofs, int which, int kind, uint32_t* out, size_t cap_words, int* log_rows, uint32_t* main_width, uint32_t* pre_width);
size_t zkhip_machine_verifier_describe(const zkhip_machine_desc* inner, size_t n_proofs, int which, int kind, uint32_t* out, size_t cap, int* log_rows, uint32_t* main_width,
                                       uint32_t* pre_width);
size_t zkhip_machine_verifier_host_tables(const zkhip_machine_desc* inner, const uint8_t* const* proofs, const size_t* proof_lens, size_t n_proofs, const uint32_t* public_values,
                                          size_t n_public, int which, uint32_t* out, size_t cap);
int zkhip_ntt_pass(zkhip_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t ld, int log_n,
                   uint32_t width, int which);
int zkhip_last_prove_debug(zkhip_ctx* ctx, zkhip_prove_debug* out);
int zkhip_selftest_lockstep(int members, int rounds);
int zkhip_selftest_host_simd(double* ns_x16, double* ns_scalar);
double zkhip_host_permutation_ns(int form);
void zkhip_lockstep_stats(uint64_t out[6]);
uint64_t zkhip_lockstep_stack_high_water(void);
/* Where the recursion machines (zkhip_prove_shard_verifier[_air], zkhip_prove_machine_verifier, zkhip_prove_shard_tree) make their per-query witness tables -- ROWSUM,
 * QUERY, the fold rows, the queries' Poseidon2 rows: 0 (default) = device kernels over the inner proofs' words (round 6), 1 = the host's walk of rounds 4 - 5 (the same tables
 * word for word: the fallback, and what tests/test_gpu_recursion_machine.py compares the kernels with).  Process-wide; returns the previous setting. */
int zkhip_recursion_witnesses_on_host(int enable);

#ifdef __cplusplus
}
#endif
#endif /* ZKHIP_CHIPS_H */
