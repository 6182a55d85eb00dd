/*
 * zkhip.h -- C ABI of libzkhip.so: the MI355X (gfx950) shard-prove hot path for zkTLS.
 *
 * This is the drop-in boundary a Rust `ZkProver` backend binds with `extern "C"`
 * (binding sketch: INTEGRATION.md).  The reference has no FFI of its own; each entry
 * point below names the reference interface or third-party call it stands in for:
 *
 *   zkhip_prove_shard / zkhip_verify_shard
 *       the span the reference times at crates/guest-prover-sp1/src/sp1.rs:115-118
 *       (`client.prove`, :116) and checks at :120 (`client.verify`), reached through
 *       trait ZkProver::prove (core/src/prelude.rs:12-18); RISC Zero twin:
 *       crates/guest-prover-r0/src/prover.rs:88-93.
 *   zkhip_prove_chips / zkhip_verify_chips
 *       the same span for a shard made of several chips of different heights, as sp1-stark's ShardProof is
 *       (reference Cargo.lock:6172): one mixed-height commitment per phase, one FRI proof.
 *   zkhip_prove_shard_air, zkhip_prove_chips_air, zkhip_prove_machine (+ their verifiers)
 *       the same span with the AIR supplied as DATA: a constraint program per table (what p3-air `Air::eval` bodies state,
 *       Cargo.lock:3835), several tables of different heights in one proof, and the tables' lookups as interaction tables
 *       (sp1-stark's permutation argument, Cargo.lock:6172) -- the structure of an SP1 shard proof without its chips.
 *   zkhip_prove_sha256 / zkhip_verify_sha256, zkhip_sha256_gen_trace, zkhip_range_table
 *       one real chip on that path -- SHA-256 compression, the hash of the transcripts the guest checks (upstream: the
 *       ShaExtend / ShaCompress chips of sp1-core-machine, Cargo.lock:5822) -- with its trace generated on the device.
 *   zkhip_machine_setup / zkhip_prove_machine_keyed / zkhip_verify_machine_keyed, zkhip_sha256_setup / zkhip_prove_sha256_machine
 *       `let (pk, vk) = client.setup(guest_program)` (sp1.rs:113) and the prove / verify calls that take pk / vk (:116, :120): the
 *       chips' preprocessed columns committed once (sp1-stark StarkMachine::setup, Cargo.lock:6172), proofs opened against that key.
 *   zkhip_prove_transcripts, zkhip_prove_sha256_sharded / zkhip_verify_sha256_sharded
 *       the reference's batch and large-transcript configurations (BASELINE.json configs[2], configs[3]) with a real statement per proof:
 *       many transcripts, or the shards of one long message, dealt over the GPUs of the node by one call.
 *   zkhip_p2chip_air, zkhip_prove_merkle_paths / zkhip_verify_merkle_paths
 *   (DEPRECATED generations of the recursion step -- kept for their tests; new callers: zkhip_prove_shard_verifier, or zkhip_prove_fri_indices_batch as the cheap FRI-only mode)
 *   zkhip_fri_view_shard, zkhip_fri_chip_air, zkhip_fri_queries_key, zkhip_prove_fri_queries / zkhip_verify_fri_queries,
 *   zkhip_fri_view_shard_paths, zkhip_fri_layers_key, zkhip_prove_fri_layers / zkhip_verify_fri_layers,
 *   zkhip_fri_view_transcript, zkhip_fri_transcript_key, zkhip_prove_fri_transcript / zkhip_verify_fri_transcript,
 *   zkhip_fri_indices_key, zkhip_prove_fri_indices / zkhip_verify_fri_indices
 *       a first recursion step: the FRI folds of a shard proof checked inside a (keyed machine) proof -- what `compress` behind
 *       SP1ProofMode::Groth16 (sp1.rs:116) spends its rows on besides Poseidon2.
 *       a second real chip -- the Poseidon2 permutation with Merkle-path / leaf-hash chaining, what the recursion stages behind
 *       SP1ProofMode::Groth16 (sp1.rs:116) spend their rows on (sp1-recursion's Poseidon2 chips, Cargo.lock:6172 ff.).
 *   zkhip_shard_verifier_setup, zkhip_prove_shard_verifier, zkhip_verify_shard_recursive
 *       the COMPRESS stage behind the same line (sp1.rs:116: core -> compress; prover.rs:90: lift -> join): whole shard proofs verified inside
 *       ONE proof whose verifier needs the shape, the public values and the shape's key -- no byte of a shard proof.
 *   zkhip_shard_verifier_setup_air, zkhip_prove_shard_verifier_air, zkhip_verify_shard_recursive_air, zkhip_prove_transcripts_air,
 *   zkhip_prove_sha256_compressed / zkhip_verify_sha256_compressed
 *       the same stage for proofs that carry a real statement: inner proofs of any constraint program of degree <= 3 (the SHA-256 chip's:
 *       64 transcript proofs -> one; the shards of one long message -> one).
 *   zkhip_machine_verifier_setup / _key_host, zkhip_prove_machine_verifier, zkhip_verify_machine_recursive, zkhip_sha256_machine_describe
 *       the same stage over KEYED-MACHINE proofs (lookups, chips of mixed heights, preprocessed openings in-circuit): the join of joins --
 *       sp1-recursion's compress tree (Cargo.lock:6172 ff.; RISC Zero's join of joins, prover.rs:90) -- and the keyed transcript proofs.
 *   zkhip_prove_shard_verifier_batch, zkhip_prove_shard_tree
 *       that tree's first level as ONE call (the joins of one shape dealt over the GPUs, several in flight), and the whole tree as one call
 *       (shard proofs in, ONE proof out: the reference's compress, sp1.rs:116).
 *   zkhip_proof_to_bincode / zkhip_chips_proof_to_bincode (+ _from_bincode)
 *       what `prover_output.bytes()` carries (sp1.rs:122-123): bincode-shaped forms of the proofs ([RECALLED] field order).
 *   zkhip_prove_segment
 *       the span crates/guest-prover-r0/src/prover.rs:88-93 times, from RISC Zero's column-major Hal layout.
 *   zkhip_prove_shard_host, zkhip_prove_shards, zkhip_prove_shards_multi, zkhip_commit
 *       host-pointer variant of the prove entry; all shards of one execution in one call (sp1.rs:116 / prover.rs:90 prove
 *       every shard / segment inside one call), on one GPU or dealt round-robin over the GPUs of the node;
 *       TwoAdicFriPcs::commit (Cargo.lock:3930) alone.
 *   zkhip_perm_trace, zkhip_gen_trace_logup
 *       sp1-stark generate_permutation_trace (Cargo.lock:6172): per-row extension inverses + running sum (LogUp).
 *   zkhip_coset_lde, zkhip_dft, zkhip_ntt_pass
 *       p3-dft Radix2DitParallel::{coset_lde_batch, dft_batch} (reference
 *       Cargo.lock:3903) == risc0-zkp Hal::{batch_interpolate_ntt,
 *       batch_expand_into_evaluate_ntt} (Cargo.lock:5057).
 *   zkhip_hash_rows, zkhip_merkle_commit
 *       p3-merkle-tree FieldMerkleTreeMmcs::commit (Cargo.lock:4013) == Hal::hash_rows
 *       + Hal::hash_fold.
 *   zkhip_quotient_values, zkhip_open_at, zkhip_fri_fold
 *       sp1-stark quotient_values (Cargo.lock:6172), p3-fri TwoAdicFriPcs::open and
 *       fold (Cargo.lock:3930) == Hal::{eval_check, batch_evaluate_any, fri_fold}.
 *
 * Conventions:
 *   - every function returns 0 on success or a negative zkhip_status; it never throws,
 *     aborts or unwinds across the ABI (the Rust glue wraps calls in catch_unwind,
 *     sp1.rs:85); zkhip_last_error() gives the thread-local message;
 *   - a context is bound to one device ordinal and one HIP stream; calls on one context
 *     are serialised by the caller, separate contexts are independent; no global state (one opt-in cache: the internal
 *     contexts of zkhip_prove_shards*, see zkhip_release_cached_contexts) and NO environment variable is read: the mode
 *     flags the reference passes through the process environment (sp1.rs:20-29) stay the caller's business;
 *   - matrices are row-major uint32 words in MONTGOMERY form (R = 2^32), canonical range
 *     [0, p), p = 2^31 - 2^27 + 1 -- the in-memory form of p3 MontyField31; `ld` is the
 *     row pitch in words; extension elements are 4 consecutive words (x^4 = 11);
 *   - pointers named d_* are DEVICE pointers (hipMalloc / torch data_ptr); proof bytes
 *     and public values are HOST pointers; digests on the device are 8 Montgomery words;
 *   - proof bytes hold CANONICAL little-endian words (layout: DESIGN.md section 6);
 *     a proof of <= 4 bytes means "no proof" (sp1.rs:128-130), which this library
 *     never returns on success.
 *   - there is NO CPU fallback: without a usable gfx950 device every compute entry
 *     point fails with ZKHIP_ERR_NO_DEVICE.
 *
 * Three headers (round 6): THIS one holds what a `ZkProver` backend binds -- contexts, the operators of the shard prover, the shard / multi-chip /
 * machine provers and verifiers, the statement-level SHA-256 entries, the batch entries, the recursion (compress) stage, parameters and
 * serialisation; zkhip_hal.h the RISC Zero `Hal` operator set; zkhip_chips.h the chip level -- chip programs, trace
 * generators, the descriptions of the recursion machines, the FRI-only machines, diagnostics and self-tests.  INTEGRATION.md section 1 lists the
 * entries a `ZkProver::prove` needs.
 */
#ifndef ZKHIP_H
#define ZKHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKHIP_VERSION 100 /* 0.1.0 */

typedef enum {
    ZKHIP_OK = 0,
    ZKHIP_ERR_INVALID = -1,     /* bad argument                        */
    ZKHIP_ERR_NO_DEVICE = -2,   /* no HIP device / wrong architecture  */
    ZKHIP_ERR_HIP = -3,         /* HIP runtime error (see last_error)  */
    ZKHIP_ERR_NOMEM = -4,
    ZKHIP_ERR_BUFFER = -5,      /* caller buffer too small             */
    ZKHIP_ERR_VERIFY = -6,      /* proof rejected                      */
    ZKHIP_ERR_INTERNAL = -7
} zkhip_status;

typedef struct zkhip_ctx zkhip_ctx;

/* Proof-system shape.  SP1-core-like (sp1-stark 4.1.4 BabyBearPoseidon2): log_blowup 1, 100 queries,
 * 16 proof-of-work bits, the last three fields 0.  RISC-Zero-like (risc0-zkp 1.2.5, reference
 * Cargo.lock:5057): log_blowup 2, 50 queries, 0 PoW bits, log_fold 4,
 * log_final 8, hash_width 24. */
typedef struct {
    int32_t log_blowup;         /* 1 .. 3 */
    int32_t num_queries;
    int32_t pow_bits;
    /* 0: no lookup argument.  Q > 0: the first Q pairs of column groups are tied by a LogUp
     * lookup argument (sp1-stark permutation trace): group 2q+1 must hold a
     * row permutation of group 2q's (a, b) columns, see zkhip_gen_trace_logup. */
    int32_t logup_pairs;
    /* FRI / hash shape; 0 selects the SP1 default of each field */
    int32_t log_fold;           /* every committed FRI layer folds by 2^log_fold (default 1; RISC Zero: 4);
                                 * (log_n - log_final) must be a multiple of it */
    int32_t log_final;          /* folding stops at a polynomial of < 2^log_final coefficients, sent in clear
                                 * (default 0: a constant; RISC Zero: 8) */
    int32_t hash_width;         /* Poseidon2 width of every Merkle tree: 16 (rate 8, default) or 24 (rate 16) */
    /* RISC Zero's group order (risc0-zkp prove::Prover commits a code, a data and an accum group: reference Cargo.lock:5057,
     * behind crates/guest-prover-r0/src/prover.rs:90).  0: one trace commitment.  Wc > 0 (a multiple of 4, < width): the first Wc
     * columns ("code") and the rest ("data") get a Merkle tree each, the code root committed and observed first; with
     * logup_pairs > 0 the permutation trace is the third ("accum") group, then the quotient ("check") -- proof version 8.
     * Only zkhip_prove_shard / zkhip_prove_segment / zkhip_verify_shard read it; leave it 0 elsewhere. */
    int32_t code_width;
} zkhip_params;
/* FRI configurations of the two SDKs the reference drives, as initialisers.  [RECALLED]: sp1-stark 4.1.4 /
 * risc0-zkp 1.2.5 are not in /root/reference (Cargo.lock:6172, 5057) -- to be checked against them before any wire-compatibility claim.
 *   SP1 core shards (sp1.rs:116, first stage of `prove`):   blowup 2,  100 queries, 16 proof-of-work bits, fold by 2, constant final polynomial
 *   SP1 compress / recursion stage:                         blowup 4,  50 queries, 16 bits            (same FRI shape)
 *   SP1 shrink / wrap stage: blowup 16, 25 queries -- log_blowup 4 is beyond this library's [1, 3]
 *   RISC Zero segments (prover.rs:90):                      blowup 4,  50 queries, no proof of work, fold by 16, 256 final coefficients, Poseidon2 width 24 */
#define ZKHIP_PARAMS_SP1_CORE     {1, 100, 16, 0, 0, 0, 0, 0}
#define ZKHIP_PARAMS_SP1_COMPRESS {2, 50, 16, 0, 0, 0, 0, 0}
#define ZKHIP_PARAMS_RISC0        {2, 50, 0, 0, 4, 8, 24, 0}

/* ---- library / context ---- */
int zkhip_version(void);
const char* zkhip_last_error(void);
/* number of usable devices (0 when there is no GPU; never fails) */
int zkhip_device_count(void);
/* stream: a hipStream_t owned by the caller, or NULL to let the context create one */
int zkhip_ctx_create(int device, void* stream, zkhip_ctx** out);
void zkhip_ctx_destroy(zkhip_ctx* ctx);
int zkhip_ctx_sync(zkhip_ctx* ctx);
/* How this PROCESS's host threads wait for the GPU: 0 = the runtime's default, which polls -- lowest latency, one busy core per waiting
 * thread (a rank with four shards in flight keeps ~4.7 cores busy); 1 = hipDeviceScheduleBlockingSync on every visible device: a waiting
 * thread sleeps until the interrupt (tens of microseconds more per wait).  For hosts that give the process fewer cores than it has waiting
 * threads -- a container CPU quota, eight ranks on one node.  `device`: the ordinal the process will use, or -1 for every visible device
 * (a rank of eight should name its own).  Must be called before that device is used by the process (before the first context, and
 * before another library initialises the device); ZKHIP_ERR_INVALID / ZKHIP_ERR_HIP when it is too late. */
int zkhip_set_wait_mode(int blocking, int device);
void* zkhip_ctx_stream(zkhip_ctx* ctx);

/* ---- device memory helpers (for callers without their own allocator) ---- */
int zkhip_malloc(zkhip_ctx* ctx, size_t bytes, void** d_ptr);
int zkhip_free(zkhip_ctx* ctx, void* d_ptr);
int zkhip_memcpy_h2d(zkhip_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int zkhip_memcpy_d2h(zkhip_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
/* element-wise canonical <-> Montgomery on the device (n words, in place allowed) */
int zkhip_to_monty(zkhip_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t n);
int zkhip_from_monty(zkhip_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, size_t n);

/* ---- synthetic shards: element (r, c) = splitmix64(seed)[r*width+c] mod p ---- */
int zkhip_fill_uniform(zkhip_ctx* ctx, uint64_t seed, int log_n, uint32_t width,
                       uint32_t* d_out, size_t ld);
/* AIR-satisfying trace of shard `shard` (width % 4 == 0); stream seed = seed + shard */
int zkhip_gen_trace(zkhip_ctx* ctx, uint64_t seed, uint64_t shard, int log_n, uint32_t width,
                    uint32_t* d_out, size_t ld);

/* same AIR with lookups: the odd group of each of the first `pairs` group pairs receives the even
 * group's (a, b) columns under the row permutation 5i+3 mod N (prove with logup_pairs = pairs) */
int zkhip_gen_trace_logup(zkhip_ctx* ctx, uint64_t seed, uint64_t shard, int log_n, uint32_t width,
                          int pairs, uint32_t* d_out, size_t ld);

/* lookups BETWEEN two tables of equal height (zkhip_chip.partner): this table's receiver groups hold the sender groups of
 * the partner table (stream seed + partner_shard, row pitch partner_width) under the same row permutation */
int zkhip_gen_trace_logup_cross(zkhip_ctx* ctx, uint64_t seed, uint64_t shard, uint64_t partner_shard, int log_n, uint32_t width,
                                uint32_t partner_width, int pairs, uint32_t* d_out, size_t ld);

/* ---- NTT / LDE over the columns of a row-major matrix, 2^log_n rows, 0 <= log_n <= 22 (fewer than 32 rows: by
 * definition, out of place).  Up to 2^20 rows a transform is two launches of the pass kernel; 2^21 and 2^22 rows (SP1 core
 * shards reach those heights: reference benchmark.md:9) add one streaming radix-2 / radix-4 pass and take a row pitch of at
 * most 512 / 256 words -- except zkhip_coset_lde at log_blowup 1, which runs its tile passes on dense 2^20-row classes and takes any
 * pitch (up to 1024 columns). ---- */
/* forward DFT: natural rows in; rows out natural (bitrev_out = 0) or bit-reversed (1);
 * inverse DFT (inverse = 1): natural in, natural out, scaled by 1/N. */
int zkhip_dft(zkhip_ctx* ctx, const uint32_t* d_in, size_t in_ld, uint32_t* d_out, size_t out_ld,
              int log_n, uint32_t width, int inverse, int bitrev_out);
/* coset_lde_batch(in, log_blowup, shift).bit_reverse_rows(): out has 2^(log_n+log_blowup)
 * rows; row bitrev(i) = f(shift * w^i).  `shift` is a CANONICAL field element. */
int zkhip_coset_lde(zkhip_ctx* ctx, const uint32_t* d_in, size_t in_ld, uint32_t* d_out,
                    size_t out_ld, int log_n, uint32_t width, int log_blowup, uint32_t shift);
/* benchmark hook: ONE launch of the NTT pass kernel (the roofline kernel) on a 2^log_n x width matrix.
 * which = 0 strided (first) pass, 1 contiguous (second) pass of a forward transform from d_in to d_out (d_out may equal d_in).
 * which = 2..5: one of the four launches of the trace LDE exactly as zkhip_prove_shard enqueues them, on THIS context's own
 * workspaces (the in-proof buffer placement): 2 = first inverse pass (strided, d_in = the trace -> coefficient workspace),
 * 3 = second inverse pass (contiguous, in place), 4 = first forward pass of the coset (coefficients -> LDE workspace,
 * strided bit-reversed stores), 5 = second forward pass (contiguous, in place); d_out is ignored, d_in only read by 2. */
/* The launches of the trace LDE as the prover enqueues them TODAY for 2^20 rows x a multiple of 32 columns (same workspaces as
 * which = 2..5 above): which = 6 first inverse pass in its block form (strided in -> one contiguous block per tile), 7 the FUSED
 * middle launch (second inverse pass + first forward pass of both cosets of a blowup-2 LDE: reads the blocks once, writes both
 * cosets; 12 B per trace cell), 5 the second forward pass of a coset.  One LDE = 6, 7, 5, 5 (36 B per trace cell; the unfused
 * sequence 2, 3, 4, 5, 4, 5 moves 48 B and remains in use for every other shape). */
/* LDE fusion switch of a context (default 1 = on where the shape allows): 0 forces the unfused four-pass sequence everywhere.
 * Returns the previous setting, or a negative status.  Both settings produce bit-identical matrices (tests/test_gpu_parity.py);
 * the switch exists for that test and for A/B timing. */
int zkhip_ctx_set_lde_fusion(zkhip_ctx* ctx, int on);

/* ---- commit: the PCS `commit` of one trace matrix in a single call (p3-fri TwoAdicFriPcs::commit, reference
 * Cargo.lock:3930; RISC Zero `commit_group`): coset LDE on shift * <w_{N 2^b}> (bit-reversed rows) into d_lde
 * (2^(log_n+b) x width), Merkle tree into d_tree ((2^(log_n+b+1) - 1) * 8 words), root copied to the host
 * (8 canonical words).  hash_width 16 or 24. ---- */
int zkhip_commit(zkhip_ctx* ctx, const uint32_t* d_trace, size_t ld, int log_n, uint32_t width, int log_blowup,
                 int hash_width, uint32_t* d_lde, uint32_t* d_tree, uint32_t root[8]);

/* ---- Poseidon2 Merkle commitment ---- */
/* states: count x 16 words, permuted in place (known-answer tests) */
int zkhip_poseidon2_permute(zkhip_ctx* ctx, uint32_t* d_states, size_t count);
/* leaf digests of the row-wise concatenation of nmats (<= 4) equal-height matrices */
int zkhip_hash_rows(zkhip_ctx* ctx, const uint32_t* const* d_mats, const size_t* lds,
                    const uint32_t* widths, int nmats, size_t height, uint32_t* d_digests);
/* full tree over 2^log_h leaves: d_tree holds (2^(log_h+1) - 1) * 8 words, level 0
 * (leaves) first, root last. */
int zkhip_merkle_commit(zkhip_ctx* ctx, const uint32_t* const* d_mats, const size_t* lds,
                        const uint32_t* widths, int nmats, int log_h, uint32_t* d_tree);

/* matrices of different power-of-two heights (an SP1 shard commits one matrix per chip):
 * the tallest form the leaves, a shorter one is injected at the level that has as many nodes as
 * it has rows, node = compress(node, sponge(row)).  Tree layout as above for the tallest height;
 * at most 8 matrices per distinct height, 32 in all. */
int zkhip_merkle_commit_mixed(zkhip_ctx* ctx, const uint32_t* const* d_mats, const size_t* lds,
                              const uint32_t* widths, const int* log_heights, int nmats, uint32_t* d_tree);

/* (the RISC Zero `Hal` operator set -- column-major transforms, width-24 Poseidon2 and SHA-256 hashing, the eltwise / mix / evaluate operators: include/zkhip_hal.h) */

/* ---- STARK stages (synthetic AIR, log_blowup = 1) ---- */
/* quotient values on the LDE coset, bit-reversed rows: d_out[2^(log_n+1)][4] */
int zkhip_quotient_values(zkhip_ctx* ctx, const uint32_t* d_lde, size_t ld, int log_n,
                          uint32_t width, const uint32_t alpha[4] /* host, Montgomery */,
                          uint32_t* d_out);
/* LogUp permutation trace of a main trace (natural rows): d_out[2^log_n][4 (pairs + 1)] =
 * [phi_0 .. phi_{Q-1} | running sum]; gamma, beta: host, Montgomery extension elements */
int zkhip_perm_trace(zkhip_ctx* ctx, const uint32_t* d_trace, size_t ld, int log_n, uint32_t width, int pairs,
                     const uint32_t gamma[4], const uint32_t beta[4], uint32_t* d_out);
/* barycentric opening of every column of a bit-reversed LDE at `npoints` extension
 * points (host, Montgomery): h_out[npoints][width][4] (host, Montgomery) */
int zkhip_open_at(zkhip_ctx* ctx, const uint32_t* d_lde, size_t ld, int log_n, int log_blowup,
                  uint32_t width, const uint32_t* z, int npoints, uint32_t* h_out);
/* one fold-by-2 FRI step on 2^log_h extension elements (bit-reversed order) */
int zkhip_fri_fold(zkhip_ctx* ctx, const uint32_t* d_in, int log_h,
                   const uint32_t beta[4] /* host, Montgomery */, uint32_t* d_out);

/* fold of arity 2^log_arity (RISC Zero folds by 16, log_arity = 4): log_arity chained fold-by-2
 * launches with beta, beta^2, beta^4, ...; d_out gets 2^(log_h - log_arity) elements */
int zkhip_fri_fold_k(zkhip_ctx* ctx, const uint32_t* d_in, int log_h, int log_arity,
                     const uint32_t beta[4] /* host, Montgomery */, uint32_t* d_out);

/* ---- whole shard ---- */
size_t zkhip_proof_size(int log_n, uint32_t width, const zkhip_params* prm, size_t n_public);
/* d_trace: 2^log_n x width AIR trace (Montgomery).  public_values: host, canonical.
 * proof: host buffer of `cap` bytes; *len receives the bytes written. */
int zkhip_prove_shard(zkhip_ctx* ctx, const uint32_t* d_trace, size_t ld, int log_n,
                      uint32_t width, const uint32_t* public_values, size_t n_public,
                      const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len);
/* Host-pointer variant: h_trace is a HOST buffer (pageable or pinned) of 2^log_n x width CANONICAL words, row-major.
 * The library stages it in HBM (one H2D copy + conversion to Montgomery form), then runs zkhip_prove_shard.  This is
 * the entry for callers whose executor leaves the trace in host memory (the reference's provers all do); the copy is
 * PCIe-bound, see DESIGN.md section 7 for the measured rate. */
int zkhip_prove_shard_host(zkhip_ctx* ctx, const uint32_t* h_trace, int log_n, uint32_t width,
                           const uint32_t* public_values, size_t n_public, const zkhip_params* prm,
                           uint8_t* proof, size_t cap, size_t* len);
/* A whole execution at once: the shards of one `prove` call (sp1.rs:116 proves every shard of the execution; prover.rs:90 every
 * segment) are independent, so up to `in_flight` of them are proven at the same time, each on an internal context (= HIP stream)
 * and host thread -- the latency-bound stretches of one proof hide under the kernels of the others (14.9 ms per
 * 2^20 x 256 shard with four in flight against 19 ms one by one).  Traces are device pointers that stay valid during the call, or HOST
 * pointers to canonical words when `host_traces` is non-zero (as zkhip_prove_shard_host).  Every job reports its own status and proof
 * length; the return value is ZKHIP_OK when all succeeded, else the code of the first failing job (zkhip_last_error: its message). */
typedef struct zkhip_shard_job {
    const uint32_t* trace;          /* 2^log_n x width, row-major; device (Montgomery, pitch ld) or host (canonical, dense) */
    size_t ld;
    int32_t log_n;
    uint32_t width;
    const uint32_t* public_values;  /* host, canonical */
    size_t n_public;
    uint8_t* proof;                 /* host buffer */
    size_t proof_cap;
    size_t proof_len;               /* out */
    int32_t status;                 /* out */
} zkhip_shard_job;
int zkhip_prove_shards(int device, zkhip_shard_job* jobs, int n_jobs, const zkhip_params* prm, int in_flight, int host_traces);
/* The same batch over SEVERAL GPUs of one node from one process and one call -- what a Rust `ZkProver::prove` needs, since the
 * reference proves all shards of an execution inside one call (crates/guest-prover-sp1/src/sp1.rs:116; segments:
 * crates/guest-prover-r0/src/prover.rs:90).  Shards are independent: shard s is proven on
 * devices[s mod n_devices] (zkhip_shard_device), `in_flight_per_device` at a time on each device; there is no exchange between
 * devices, proofs land in the jobs' host buffers.  devices == NULL with n_devices == 0 means every visible device.  With device
 * traces (host_traces == 0) job s's trace pointer must live on the device zkhip_shard_device(s, ...) names; host traces are
 * staged by the library on that device.  Status / error reporting as zkhip_prove_shards. */
int zkhip_prove_shards_multi(const int* devices, int n_devices, zkhip_shard_job* jobs, int n_jobs, const zkhip_params* prm,
                             int in_flight_per_device, int host_traces);
/* The same batch when every job is a trace of ONE constraint program (zkhip_prove_shard_air's format): e.g. the SHA-256 chip traces of
 * sixty-four transcripts dealt over the GPUs of a node.  Device traces only; each job brings its own public values. */
int zkhip_prove_shards_air_multi(const int* devices, int n_devices, zkhip_shard_job* jobs, int n_jobs, const uint32_t* program, size_t program_words,
                                 const zkhip_params* prm, int in_flight_per_device);
/* the device ordinal shard `shard_index` of a batch is proven on (devices == NULL: ordinal shard_index mod n_devices); -1 on bad arguments */
int zkhip_shard_device(int shard_index, const int* devices, int n_devices);
/* the internal contexts of zkhip_prove_shards stay cached between calls (creating and freeing multi-GiB workspaces costs more than
 * a proof); this frees them -- and with them the HOST tables the recursion provers keep between calls (zkhip_prove_shard_verifier[_air / _batch],
 * zkhip_prove_machine_verifier, zkhip_prove_shard_tree: hundreds of megabytes of zeroed words per call, zeroed again on a thread of their own when
 * a call is done and handed to the next call of the same shape; at most 6 GiB are kept) */
void zkhip_release_cached_contexts(void);
/* Segment proof from RISC Zero's data layout (risc0-zkp Hal, reference Cargo.lock:5057; call site
 * crates/guest-prover-r0/src/prover.rs:90): d_cols holds `width` contiguous columns of 2^log_n words
 * (column-major [width][2^log_n], Montgomery).  Same proof as zkhip_prove_shard on the transposed matrix;
 * use the RISC-Zero-like zkhip_params for its shape.  Verified by zkhip_verify_shard. */
int zkhip_prove_segment(zkhip_ctx* ctx, const uint32_t* d_cols, int log_n, uint32_t width,
                        const uint32_t* public_values, size_t n_public, const zkhip_params* prm,
                        uint8_t* proof, size_t cap, size_t* len);
/* host-side verifier (CPU; the reference verifies on the CPU too, sp1.rs:120).
 * Returns ZKHIP_OK or ZKHIP_ERR_VERIFY; *reason (optional) gets the failing check. */
int zkhip_verify_shard(const uint8_t* proof, size_t len, int log_n, uint32_t width,
                       const uint32_t* public_values, size_t n_public,
                       const zkhip_params* prm, int* reason);

/* ---- the AIR as DATA: constraint programs.  Upstream a constraint is a Rust `Air::eval`
 * body driven through p3-uni-stark's constraint folders (p3-air, p3-uni-stark 0.2.1-succinct: reference Cargo.lock:3835, 4055;
 * sp1-stark :6172; behind sp1.rs:116): a polynomial in the local / next row, the public values and the selectors
 * is_first_row / is_last_row / is_transition, folded as acc = acc * alpha + constraint.  A program is that polynomial in
 * sum-of-products form, interpreted by the quotient kernel on the device and by the verifier at zeta; any AIR of degree <= 5
 * is proven without touching a kernel: degree <= 3 (the bound of SP1's core machine) gives two quotient chunks
 * (log_quotient_degree 1), degree 4 or 5 four chunks (log_quotient_degree 2; needs log_blowup >= 2).  HOST words, canonical residues:
 *   [0] 0x50524941 "AIRP"  [1] 1  [2] width  [3] constraints K  [4] n_public  [5] total words
 *   K x { selector (0 every row, 1 first row, 2 last row, 3 transition), n_terms, n_terms x { coefficient, degree d <= 5, d variables } }
 *   variable = kind << 30 | index;  kind 0: column of the local row, 1: column of the next row, 2: public value.
 * Value of a constraint = selector * sum_t coeff_t * prod_j var_tj (a selector counts one degree).  Proofs carry version 7: the
 * extended header, then the 8-word digest of the program (zkhip_air_digest), which the transcript observes -- a proof is bound
 * to its AIR.  zkhip_params.logup_pairs must be 0 (the built-in lookup argument belongs to the built-in AIR). ---- */
int zkhip_air_validate(const uint32_t* program, size_t words, uint32_t width, size_t n_public);
int zkhip_air_digest(const uint32_t* program, size_t words, uint32_t out[8]);
/* the built-in synthetic AIR written as a program; out == NULL: *words receives the size */
size_t zkhip_proof_size_air(const uint32_t* program, size_t program_words, int log_n, uint32_t width, const zkhip_params* prm, size_t n_public);
int zkhip_prove_shard_air(zkhip_ctx* ctx, const uint32_t* program, size_t program_words, const uint32_t* d_trace, size_t ld, int log_n,
                          uint32_t width, const uint32_t* public_values, size_t n_public, const zkhip_params* prm,
                          uint8_t* proof, size_t cap, size_t* len);
int zkhip_verify_shard_air(const uint32_t* program, size_t program_words, const uint8_t* proof, size_t len, int log_n, uint32_t width,
                           const uint32_t* public_values, size_t n_public, const zkhip_params* prm, int* reason);
/* stage level: quotient values of a program on its quotient domain = the first 2^(log_n + lqd) rows of the bit-reversed LDE (d_lde
 * must hold at least that many rows; lqd = 1 for degree <= 3, 2 for degree 4 / 5); d_out: that many extension elements, in the
 * LDE's bit-reversed row order (layout of zkhip_quotient_values); alpha: host, Montgomery */
int zkhip_quotient_values_air(zkhip_ctx* ctx, const uint32_t* program, size_t program_words, const uint32_t* d_lde, size_t ld, int log_n,
                              uint32_t width, const uint32_t* public_values, size_t n_public, const uint32_t alpha[4], uint32_t* d_out);

/* ---- a real chip on the constraint-program path: SHA-256 compression (the hash of the TLS transcripts the reference's guest checks;
 * upstream SP1 proves it through the ShaExtend / ShaCompress chips of sp1-core-machine 4.1.4, reference Cargo.lock:5822, behind
 * crates/guest-prover-sp1/src/sp1.rs:116).  One row per round, 64 rows per 64-byte block, 640 columns (612 in use, padded to whole 32-column tiles), degree 3, 91 public values =
 * the digest as 16-bit limbs (low limb of word 0 first), then 75 values the VERIFIER derives from the message's length
 * (zkhip_sha256_padding_publics: the block count, where the 0x80 byte sits, which words must be zero, the length field).
 * THE EXACT RELATION a proof attests (round 5): "I know a message of exactly L bytes whose SHA-256 digest is this" -- L public.  The
 * FIPS 180-4 padding is constrained in-circuit: the number of active blocks is (L + 8) / 64 + 1 (a block counter that must reach zero
 * exactly where ACT drops), the boundary word holds the message's last bytes, then 0x80, then zeros (checked bit by bit on the row where
 * that word is the schedule's W_t), every word between it and the length field is zero, and W_14, W_15 of the last block are 8 L.  (Until
 * round 4 the relation was "some block sequence's compression chain ends in this digest": ACT could drop after any block, padding and
 * length were the host's.)  A chained shard (zkhip_sha256_air_chained: a slice of a longer message) takes the same constraints with the
 * public values of ITS slice.  In the unkeyed program the OUT limbs of the working variables d and h are not range-checked: their integer
 * value mod 2^32 is what the next round consumes and the three-bit carries bound their growth; the keyed machine (zkhip_sha256_setup) looks
 * every 16-bit limb up in a range table.  Column layout and constraints: csrc/sha256_chip.hip. ---- */
#define ZKHIP_SHA256_WIDTH 640
#define ZKHIP_SHA256_PUBLIC 91           /* 16 digest limbs + ZKHIP_SHA256_PADDING_PUBLIC */
#define ZKHIP_SHA256_PADDING_PUBLIC 75
/* the 75 padding values of a trace that holds blocks [first_block, first_block + n_active) of the padded message of message_len bytes
 * (a whole message: first_block 0, n_active (message_len + 8) / 64 + 1).  Host only; what a verifier computes instead of trusting. */
/* the constraint program (a zkhip_prove_shard_air program): returns its length in words; written when cap_words suffices */
/* the digest itself, on the host (what a verifier compares the proof's public values with) */
void zkhip_sha256_digest(const uint8_t* message, size_t len, uint8_t digest[32]);
/* FIPS 180-4 padding: returns the padded length (a multiple of 64); written when cap suffices.  Host only. */
/* trace generation on the device: blocks = the n_active = (message_len + 8) / 64 + 1 padded 64-byte blocks (host memory), n_blocks = a
 * power of two >= n_active; d_trace [64 n_blocks][ld >= 640] Montgomery; publics (host) = the 91 public values */
/* message in, digest (32 bytes, as SHA-256 prints it) and proof out: pad, generate the trace on the device, zkhip_prove_shard_air.
 * zkhip_params: any shape zkhip_prove_shard_air takes (logup_pairs = code_width = 0). */
size_t zkhip_sha256_proof_size(size_t message_len, const zkhip_params* prm);
int zkhip_prove_sha256(zkhip_ctx* ctx, const uint8_t* message, size_t message_len, const zkhip_params* prm, uint8_t digest[32],
                       uint8_t* proof, size_t cap, size_t* len);
/* host-side verifier: the block bound 2^(log_n - 6) is read from the proof header (and bound by the proof's transcript) */
int zkhip_verify_sha256(const uint8_t* proof, size_t len, const uint8_t digest[32], uint64_t message_len, const zkhip_params* prm, int* reason);

/* ---- a shard made of several chips (AIR tables) of different heights, as an SP1 shard is (sp1-stark 4.1.4 ShardProof,
 * reference Cargo.lock:6172, behind crates/guest-prover-sp1/src/sp1.rs:116): one Merkle commitment per phase over all
 * chips (shorter matrices injected at their level), one opening point, one reduced-opening vector per height joining the
 * FRI vector when folding reaches it, one FRI proof.  Chips tallest first, log_n in [5, 22] (above 2^20 rows and at a blowup other than 2 a chip's row pitch is limited as in zkhip_coset_lde: 512 words at 2^21, 256 at 2^22), at most 8 per height and 32
 * in all; zkhip_params: any log_blowup, the SP1 FRI shape (log_fold / log_final / hash_width / logup_pairs 0).
 * A chip may carry in-table LogUp pairs (logup_pairs > 0, trace from zkhip_gen_trace_logup): the permutation traces of
 * those chips are committed together in a third mixed-height tree (sp1-stark's permutation commitment).  Two chips of
 * equal height and pair count may look EACH OTHER up (partner = the other chip's index, mutual; traces from
 * zkhip_gen_trace_logup_cross): every chip with pairs then exposes the final value of its running sum in the proof and the
 * verifier checks that these add up to zero (sp1-stark's local cumulative sums).  `pairs` / `partners` arrays below may be
 * NULL (no lookups / none between chips). ---- */
typedef struct {
    const uint32_t* d_trace;    /* device, row-major 2^log_n x ld words, Montgomery */
    size_t ld;
    int32_t log_n;
    uint32_t width;
    int32_t logup_pairs;        /* 0: none */
    int32_t partner;            /* -1: lookups stay inside the chip; otherwise the index of the chip it exchanges lookups with */
} zkhip_chip;
size_t zkhip_chips_proof_size(const int32_t* log_ns, const uint32_t* widths, const int32_t* pairs, const int32_t* partners, int n_chips,
                              const zkhip_params* prm, size_t n_public);
int zkhip_prove_chips(zkhip_ctx* ctx, const zkhip_chip* chips, int n_chips, const uint32_t* public_values, size_t n_public,
                      const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len);
int zkhip_verify_chips(const uint8_t* proof, size_t len, const int32_t* log_ns, const uint32_t* widths, const int32_t* pairs,
                       const int32_t* partners, int n_chips,
                       const uint32_t* public_values, size_t n_public, const zkhip_params* prm, int* reason);

/* chips with their own AIR: programs[c] is a constraint program (zkhip_prove_shard_air's format; degree <= 5: a chip whose program has
 * degree 4 or 5 gets four quotient chunks -- its own quotient domain of four cosets, 16 quotient columns -- and needs log_blowup >= 2; the
 * header's has-program word carries the program's log_quotient_degree, 1 or 2; its n_public = the
 * shard's) or NULL for the built-in synthetic AIR -- a machine of different tables in one proof (one commitment per phase, one FRI
 * proof), as an SP1 shard is.  Proof version 9: each chip's header entry gains a has-program flag and the programs' digests follow
 * the entries, all observed.  No lookups in this version (logup_pairs = 0, partner = -1 in every zkhip_chip). */
size_t zkhip_chips_proof_size_air(const int32_t* log_ns, const uint32_t* widths, const uint32_t* const* programs, const size_t* program_words,
                                  int n_chips, const zkhip_params* prm, size_t n_public);
int zkhip_prove_chips_air(zkhip_ctx* ctx, const zkhip_chip* chips, const uint32_t* const* programs, const size_t* program_words, int n_chips,
                          const uint32_t* public_values, size_t n_public, const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len);
int zkhip_verify_chips_air(const uint8_t* proof, size_t len, const int32_t* log_ns, const uint32_t* widths, const uint32_t* const* programs,
                           const size_t* program_words, int n_chips, const uint32_t* public_values, size_t n_public, const zkhip_params* prm, int* reason);

/* The machine: chips with their own programs AND lookups as DATA (sp1-stark's permutation argument -- generate_permutation_trace /
 * eval_permutation_constraints, batch size 2, reference Cargo.lock:6172 -- with the interactions written out).  tables[c] is an
 * interaction table or NULL:
 *   [0] 0x50554B4C "LKUP"  [1] interactions I (1..64)  [2] total words
 *   I x { sign (0 send, 1 receive), multiplicity (0xFFFFFFFF: the constant 1, else a column), bus (a field element), values V (1..8), V columns }
 * A tuple's fingerprint is d = gamma + bus + sum_t beta^(t+1) v_t.  The chip's permutation trace has one extension column per pair
 * of interactions, phi_j = s_a m_a / d_a + s_b m_b / d_b (s = +1 send, -1 receive; 1/0 = 0), and the running sum of the row sums;
 * the constraints phi_j d_a d_b = s_a m_a d_b + s_b m_b d_a, S_0 = sum phi, S' = S + sum phi', S_last = C fold after the program's;
 * every chip with a table exposes its cumulative sum C and the verifier checks that the sums of the machine add up to zero:
 * every tuple sent on a bus is received with the same total multiplicity (a range table, a memory bus, a permutation ...).
 * programs[c] NULL: the built-in synthetic AIR.  Proof version 10 = version 6's layout with header entries
 * (log_n, width, has_program, interactions), then the programs' and the tables' digests.  Shape limits as zkhip_prove_chips. */
/* a range table's two columns on the device: d_table[v][value_col] = v and d_table[v][mult_col] = how often v appears in the listed
 * `columns` of d_trace (the multiplicities a `receive` interaction of the table needs), v < 2^log_table; the other columns of d_table are
 * left as they are.  Fails if a looked-up value lies outside the table. */
size_t zkhip_machine_proof_size(const int32_t* log_ns, const uint32_t* widths, const uint32_t* const* programs, const size_t* program_words,
                                const uint32_t* const* tables, const size_t* table_words, int n_chips, const zkhip_params* prm, size_t n_public);
int zkhip_prove_machine(zkhip_ctx* ctx, const zkhip_chip* chips, const uint32_t* const* programs, const size_t* program_words,
                        const uint32_t* const* tables, const size_t* table_words, int n_chips, const uint32_t* public_values, size_t n_public,
                        const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len);
int zkhip_verify_machine(const uint8_t* proof, size_t len, const int32_t* log_ns, const uint32_t* widths, const uint32_t* const* programs,
                         const size_t* program_words, const uint32_t* const* tables, const size_t* table_words, int n_chips,
                         const uint32_t* public_values, size_t n_public, const zkhip_params* prm, int* reason);

/* The keyed machine: setup, then prove.  The reference calls `client.setup(guest_program) -> (pk, vk)` before every prove
 * (crates/guest-prover-sp1/src/sp1.rs:113) and hands vk to verify (:120); in sp1-stark (StarkMachine::setup, reference Cargo.lock:6172)
 * setup generates the chips' PREPROCESSED traces -- program ROM, byte-operation tables: columns fixed by the program, not by the
 * execution -- and commits them once: the commitment is the verifying key's, the traces and their LDEs the proving key's.
 * zkhip_machine_setup does that on the device: pre[c] describes chip c's preprocessed trace (d_trace, ld, log_n, width = preprocessed
 * width, a multiple of 4; width 0: the chip has none; logup_pairs / partner unused), chips tallest first as in the machine.  The key
 * keeps device copies of the traces, their LDEs and the mixed-height tree until zkhip_machine_key_destroy; root[8] (canonical words)
 * is what a verifier needs.  A key belongs to the context it was made with.
 * In a keyed machine a chip's program and interaction table address the COMBINED row [preprocessed | main] (program width =
 * preprocessed + main width); a chip with preprocessed columns brings its own program.  Proof version 11 = version 10 with header
 * entries (log_n, width, has_program, interactions, pre_width), the key's root after the digests (observed before the trace
 * commitment), every chip's openings preceded by its preprocessed columns at zeta and zeta g, every query preceded by the
 * preprocessed rows and their path in the key's tree.  Verifier reject 33: a preprocessed row does not open the key's root. */
typedef struct zkhip_machine_key zkhip_machine_key;
int zkhip_machine_setup(zkhip_ctx* ctx, const zkhip_chip* pre, int n_chips, const zkhip_params* prm, zkhip_machine_key** key, uint32_t root[8]);
void zkhip_machine_key_destroy(zkhip_machine_key* key);
/* zkhip_machine_setup's root[8] WITHOUT a device (host only, no context, no HIP call; csrc/host_key.cpp): h_traces[c] = the chip's preprocessed
 * trace as HOST words, [2^log_ns[c]][pre_widths[c]] in Montgomery form (NULL with width 0: none), chips tallest first.  What a verifier that
 * owns no GPU calls to derive the key it checks proofs against (the reference verifies on the CPU: sp1.rs:120). */
int zkhip_machine_key_host(const uint32_t* const* h_traces, const int32_t* log_ns, const uint32_t* pre_widths, int n_chips, const zkhip_params* prm, uint32_t root[8]);
size_t zkhip_machine_proof_size_keyed(const int32_t* log_ns, const uint32_t* widths, const uint32_t* pre_widths, const uint32_t* const* programs,
                                      const size_t* program_words, const uint32_t* const* tables, const size_t* table_words, int n_chips,
                                      const zkhip_params* prm, size_t n_public);
int zkhip_prove_machine_keyed(zkhip_ctx* ctx, const zkhip_machine_key* key, const zkhip_chip* chips, const uint32_t* const* programs,
                              const size_t* program_words, const uint32_t* const* tables, const size_t* table_words, int n_chips,
                              const uint32_t* public_values, size_t n_public, const zkhip_params* prm, uint8_t* proof, size_t cap, size_t* len);
/* the same with the key's entries assigned to the machine's chips explicitly: chip c uses entry key_entries[c] of the key (-1: the chip has
 * no preprocessed columns).  The key may then hold fewer entries than the machine has chips -- a key of tables only, used by machines
 * whose other chips change height from proof to proof (sp1-stark matches preprocessed traces to a shard's chips by name).  Every entry
 * with columns is used exactly once and in the key's order. */
int zkhip_prove_machine_keyed_at(zkhip_ctx* ctx, const zkhip_machine_key* key, const int32_t* key_entries, const zkhip_chip* chips,
                                 const uint32_t* const* programs, const size_t* program_words, const uint32_t* const* tables, const size_t* table_words,
                                 int n_chips, const uint32_t* public_values, size_t n_public, const zkhip_params* prm, uint8_t* proof, size_t cap,
                                 size_t* len);
int zkhip_verify_machine_keyed(const uint8_t* proof, size_t len, const int32_t* log_ns, const uint32_t* widths, const uint32_t* pre_widths,
                               const uint32_t root[8], const uint32_t* const* programs, const size_t* program_words, const uint32_t* const* tables,
                               const size_t* table_words, int n_chips, const uint32_t* public_values, size_t n_public, const zkhip_params* prm,
                               int* reason);

/* A message of ANY length as a CHAIN of shard proofs -- the reference's large-transcript configuration (BASELINE.json configs[3]: a
 * megabyte-scale response body, 1 -> 8 GPUs) with a real statement per shard.  The chained program (zkhip_sha256_air_chained) is the chip
 * with its initial chaining value PUBLIC as well: 107 public values = the final chaining value's 16 limbs, the initial one's, then the slice's 75 padding values.  Shard s
 * covers blocks [s 2^k, (s + 1) 2^k) of the padded message and proves chain[s] -> chain[s + 1]; chain[0] is the standard initial value,
 * chain[n_shards] the digest.  The chaining values come from one pass of plain compression on the host, after which the shards are
 * independent: zkhip_prove_sha256_sharded deals them over the devices (shard s on devices[s mod n], in_flight_per_device at a time) like
 * zkhip_prove_shards_multi.  chain: (n_shards + 1) x 8 words out; proofs: n_shards x proof_stride bytes (proof_stride >=
 * zkhip_sha256_shard_proof_size(k, prm)), proof_lens[s] out.  zkhip_verify_sha256_sharded checks the whole chain on the host (bad_shard /
 * reason name the first failing shard).  zkhip_sha256_gen_trace_chained = zkhip_sha256_gen_trace from a given chaining value. */
size_t zkhip_sha256_sharded_count(size_t message_len, int log_blocks_per_shard);
size_t zkhip_sha256_shard_proof_size(int log_blocks, const zkhip_params* prm);
int zkhip_prove_sha256_sharded(const int* devices, int n_devices, const uint8_t* message, size_t message_len, int log_blocks_per_shard, const zkhip_params* prm,
                               int in_flight_per_device, uint8_t digest[32], uint32_t* chain, uint8_t* proofs, size_t proof_stride, size_t* proof_lens);
int zkhip_verify_sha256_sharded(const uint8_t* proofs, size_t proof_stride, const size_t* proof_lens, size_t n_shards, const uint32_t* chain,
                                int log_blocks_per_shard, const uint8_t digest[32], uint64_t message_len, const zkhip_params* prm, size_t* bad_shard, int* reason);
/* The chain as ONE proof -- the reference's core -> compress (sp1.rs:116) on a statement about real data: "digest = SHA-256 of a message
 * of message_len bytes", whatever the length.  The shards (all at the full height 2^(6 + log_blocks_per_shard), the last one's unused
 * blocks inactive) are dealt over `devices` as above and then verified in-circuit on ctx's device by the shard verifier machine in air
 * mode (zkhip_prove_shard_verifier_air on the chained program).  The key is a function of (the number of shards, log_blocks_per_shard,
 * inner's queries and proof-of-work bits, outer): zkhip_sha256_compress_setup on a device, zkhip_sha256_compress_key_host without one.
 * The verifier takes the proof, the digest, the length, the chain (8 (n_shards + 1) words: chain[0] must be the standard initial value,
 * chain[n_shards] the digest) and the key -- no shard proof.  At most as many shards as zkhip_shard_verifier_max_proofs_air allows
 * (66 shards of 2^14 blocks = 66 MiB at 100 queries). */
int zkhip_sha256_compress_setup(zkhip_ctx* ctx, size_t message_len, int log_blocks_per_shard, const zkhip_params* inner, const zkhip_params* outer,
                                zkhip_machine_key** key, uint32_t vk[8]);
int zkhip_sha256_compress_key_host(size_t message_len, int log_blocks_per_shard, const zkhip_params* inner, const zkhip_params* outer, uint32_t vk[8]);
size_t zkhip_sha256_compressed_proof_size(size_t message_len, int log_blocks_per_shard, const zkhip_params* inner, const zkhip_params* outer);
int zkhip_prove_sha256_compressed(zkhip_ctx* ctx, const zkhip_machine_key* key, const int* devices, int n_devices, const uint8_t* message, size_t message_len,
                                  int log_blocks_per_shard, const zkhip_params* inner, const zkhip_params* outer, int in_flight_per_device, uint8_t digest[32],
                                  uint32_t* chain, uint8_t* proof, size_t cap, size_t* len);
int zkhip_verify_sha256_compressed(const uint8_t* proof, size_t len, const uint8_t digest[32], uint64_t message_len, const uint32_t* chain, int log_blocks_per_shard,
                                   const uint32_t vk[8], const zkhip_params* inner, const zkhip_params* outer, int* reason);

/* The SHA-256 guest as a keyed machine: setup once, then one proof per message -- the reference's setup -> prove -> verify
 * (sp1.rs:113, :116, :120) on this repo's stand-in guest.  Two chips: the SHA-256 compression chip (zkhip_sha256_air, 640 columns) and a
 * 2^16-row range table that receives the four 16-bit limbs per row the chip's own constraints do not range-check (the OUT limbs of d and
 * h); the table's values are a PREPROCESSED column committed by zkhip_sha256_setup (vk = that commitment, 8 canonical words; the key holds
 * the device data), its multiplicities are counted on the device per proof.  Messages up to 2^14 blocks (1 MiB).  A proof is a version-11
 * machine proof with public values = the digest's 16 limbs and the 75 padding values of the message's length (the statement: digest = SHA-256 of a
 * message of message_len bytes); zkhip_verify_sha256_machine reads the chip's height from the proof, rebuilds
 * the machine and runs zkhip_verify_machine_keyed. */
int zkhip_sha256_setup(zkhip_ctx* ctx, const zkhip_params* prm, zkhip_machine_key** key, uint32_t vk[8]);
size_t zkhip_sha256_machine_proof_size(size_t message_len, const zkhip_params* prm);
int zkhip_prove_sha256_machine(zkhip_ctx* ctx, const zkhip_machine_key* key, const uint8_t* message, size_t message_len, const zkhip_params* prm,
                               uint8_t digest[32], uint8_t* proof, size_t cap, size_t* len);
int zkhip_verify_sha256_machine(const uint8_t* proof, size_t len, const uint8_t digest[32], uint64_t message_len, const uint32_t vk[8], const zkhip_params* prm, int* reason);
/* the keyed machine of a message of this length as data -- chip `which` (0, 1: tallest first), kind 0 its program, 1 its interaction table: what a
 * zkhip_machine_desc takes, so that proofs of zkhip_prove_transcripts go to zkhip_prove_machine_verifier (64 transcript proofs -> ONE proof). */
/* A batch of transcripts in one call -- the reference's batch configuration (BASELINE.json configs[2]: 64 independent transcripts), each proven
 * as the keyed SHA-256 machine: job i runs on devices[i mod n_devices] (NULL / 0: every visible device), `in_flight_per_device` at a time per
 * device, on pooled contexts that keep their proving key between calls (setup once per context).  Messages are host bytes; every job
 * reports its digest, proof length and status; vk receives the verifying key all proofs check against (zkhip_verify_sha256_machine).
 * verify != 0: every proof is also checked against the key on its worker's host thread before the job reports success, as the reference
 * verifies right after proving (sp1.rs:120) -- the check overlaps with the other workers' GPU work.
 * proof_cap >= zkhip_sha256_machine_proof_size(message_len, prm).  Returns ZKHIP_OK or the status of the lowest failing job. */
typedef struct zkhip_transcript_job {
    const uint8_t* message;         /* host */
    size_t message_len;
    uint8_t digest[32];             /* out: SHA-256 of the message */
    uint8_t* proof;                 /* host buffer */
    size_t proof_cap;
    size_t proof_len;               /* out */
    int32_t status;                 /* out */
} zkhip_transcript_job;
int zkhip_prove_transcripts(const int* devices, int n_devices, zkhip_transcript_job* jobs, int n_jobs, const zkhip_params* prm,
                            int in_flight_per_device, int verify, uint32_t vk[8]);
/* The same batch with every job proven as zkhip_prove_sha256 makes it (the chip alone, a version-7 proof of zkhip_sha256_air's program; no
 * key; proof_cap >= zkhip_sha256_proof_size; verify != 0 checks with zkhip_verify_sha256): the form zkhip_prove_shard_verifier_air
 * compresses -- n proofs of one trace height become ONE proof (the reference's core -> compress on a batch of transcripts, sp1.rs:116). */
int zkhip_prove_transcripts_air(const int* devices, int n_devices, zkhip_transcript_job* jobs, int n_jobs, const zkhip_params* prm,
                                int in_flight_per_device, int verify);
/* Small proofs are launch-bound: a proof of a 13 KB message is ~200 kernels, most of a few microseconds, and the GPU retires such kernels
 * at a fixed rate however many streams feed it.  The batch entries therefore prove small jobs OF ONE SHAPE in lock-step (csrc/batch.h):
 * up to `max_batch` provers run as fibers of one host thread on pooled contexts that share one stream, and their launches of the same
 * kernel merge into one launch (gridDim.z = members; their small copies, memsets and waits merge too); `lanes` such batches are in
 * flight per device.  Used by zkhip_prove_transcripts for traces of up to 2^14 rows and by zkhip_prove_shards / _multi / _air_multi for
 * shards of up to 2^24 cells (the sizes measured as launch-bound), when a device gets at least two such jobs.  The proofs are byte for
 * byte those of the unbatched path.
 * max_batch 0 or 1 switches lock-step off (defaults: 16 members, 6 lanes; lanes <= 0 keeps the current value).  Process-wide.
 * Memory: lanes x max_batch contexts per device are in use at once, NOT `in_flight` (which bounds the unbatched path only); the dealer
 * sizes both against the device's free memory (48 bytes of workspace per trace cell and member) and shrinks the batches, then the lanes,
 * when they would not fit; a member that still runs out of device memory is proven once more on its own instead of failing, and after
 * the call the process-wide context pool keeps at most that many contexts and a quarter of the device's memory.
 * zkhip_lockstep_stats: merged launches issued, member launch requests served, rounds whose members asked for different launches, then
 * nanoseconds (summed over the lanes) spent issuing launches, waiting for the stream, and in the members' own host code -- totals since
 * the library was loaded.
 * (The reference proves its batch one transcript after the other, each a full `client.prove` call: sp1.rs:116, BASELINE configs[2].) */
void zkhip_set_lockstep(int max_batch, int lanes);
/* the lanes' fiber scheduler on its own (needs no device): `members` fibers wait, vote and leave for `rounds` rounds and check that the
 * per-thread error string stays theirs across switches.  0, or the number of the first check that failed. */
/* the most bytes of a member's (fiber's) 2 MiB stack touched so far in this process, page granularity (guard pages at both ends of
 * every stack turn an overflow into a fault; this says how far from it the provers run).  0 before the first lock-step batch. */
/* The FRI commit phase of a proof outside lock-step batches is a fixed sequence of ~250 small launches, captured once per shape into a HIP
 * graph and replayed with one hipGraphLaunch per proof (csrc/prover.cpp).  on = 0 keeps the plain launches, process-wide (default on; same
 * proof bytes): a debugging switch that takes graphs out of the picture.  (It was added while chasing the SIGSEGV of runs under `rocprofv3
 * --kernel-trace`: graph launches crash there every time with no code of this library involved, tools/segv/repro_nolib g -- but plain
 * launches and copies from many threads crash inside the same interception too, so the switch does NOT make profiled runs safe;
 * profiles/r04_segv.md.) */
void zkhip_set_fri_graph(int on);

/* (the Poseidon2 permutation chip with its Merkle-path prover, the FRI-only recursion machines and zkhip_prove_fri_indices_batch: include/zkhip_chips.h) */

/* ---- THE SHARD VERIFIER AS A MACHINE: a whole shard proof checked in-circuit (csrc/shard_verifier.inl).
 * What the reference asks for behind `client.prove(&pk, &stdin, SP1ProofMode::Groth16)` (crates/guest-prover-sp1/src/sp1.rs:116: core ->
 * COMPRESS verifies the shard proofs; RISC Zero: lift -> join behind crates/guest-prover-r0/src/prover.rs:90).  Inner proofs: version 1 of
 * this library (zkhip_prove_shard with the SP1 shape -- blowup 2, fold by 2, constant final value, Poseidon2 width 16, no lookups), 2^5 ..
 * 2^22 rows, a width that is a multiple of 8.  The machine has eight chips -- the Poseidon2 chip (transcript sponge rows, every Merkle path
 * of every query), ROWSUM (the opened rows and their batched sums), the fold chip, the transcript table, QUERY (reduced openings), OPENED
 * (opened values, the AIR's constraints at zeta), SAMPLES (proof of work, query indices), SCALARS (zeta^N, selectors, the quotient
 * identity) -- and EVERY structural fact is a preprocessed column: the key (zkhip_shard_verifier_setup) is a function of the inner
 * proof's SHAPE alone.  Statement of an outer proof: "a shard proof of this shape exists that the verifier accepts for these public
 * values".  zkhip_verify_shard_recursive takes the shape, the inner proof's public values and the key -- no byte of the inner proof.
 * zkhip_shard_verifier_describe hands out the machine as data (position `which`, tallest chip first; kind 0 program, 1 interaction table,
 * 2 preprocessed trace as canonical words): tests/recursion_air.py writes the same independently, the words are compared.
 * THE JOIN: n_proofs > 1 makes ONE outer proof verify that many inner proofs of the shape (every chip holds proof 0's rows, then proof 1's, ...;
 * tags, tree numbers and query numbers carry the proof's number; the SCALARS chip has one row per proof).  public_values = those of proof
 * 0, then those of proof 1, ... (n_proofs x n_public words), which are the outer proof's public values.  Sixteen headline shard proofs
 * (15 MB) become one proof of about a megabyte.  (A TREE of joins would need a verifier of THIS machine's proofs -- version 11, lookups and
 * all -- in-circuit; that is not built: the join is flat, bounded by the 2^22-row limit of the Poseidon2 chip -- 136 headline proofs under an outer proof at blowup 2, 68 otherwise,
 * zkhip_shard_verifier_max_proofs -- and by 1024 proofs.) */
int zkhip_shard_verifier_setup(zkhip_ctx* ctx, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, size_t n_proofs, const zkhip_params* outer,
                               zkhip_machine_key** key, uint32_t vk[8]);
size_t zkhip_shard_verifier_proof_size(int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, size_t n_proofs, const zkhip_params* outer);
/* the same key WITHOUT a device: no context, no HIP call -- the preprocessed traces are low-degree-extended and committed on the host's cores
 * (csrc/host_key.cpp), so that a party that owns no GPU derives the key of the shape it means and checks a compressed proof with
 * zkhip_verify_shard_recursive alone, as the reference verifies on the CPU (crates/guest-prover-sp1/src/sp1.rs:120).  vk equals
 * zkhip_shard_verifier_setup's at every shape (headline shape, 16 proofs joined: a 2^19 x 24 table, about a second on 16 cores). */
int zkhip_shard_verifier_key_host(int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, size_t n_proofs, const zkhip_params* outer, uint32_t vk[8]);
/* the largest n_proofs ONE join takes for this inner shape: every permutation of every proof is a row of the Poseidon2 chip, which holds 2^22
 * rows under an outer proof at blowup 2 (136 proofs of the headline shape) and 2^21 under any other (68; `outer` NULL: blowup 2), and the
 * transcript table spends one preprocessed column per proof and sponge row that carries public values (497 proofs with 9 public values);
 * never more than 1024.  0: bad shape.  Host only. */
size_t zkhip_shard_verifier_max_proofs(int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, const zkhip_params* outer);
int zkhip_prove_shard_verifier(zkhip_ctx* ctx, const zkhip_machine_key* key, const uint8_t* const* shard_proofs, const size_t* shard_proof_lens, size_t n_proofs, int log_n,
                               uint32_t width, const uint32_t* public_values, size_t n_public, const zkhip_params* inner, const zkhip_params* outer, uint8_t* proof, size_t cap,
                               size_t* len);
/* The compress stage over several joins and several GPUs (sp1.rs:116: the recursion tree's first level; prover.rs:90: lift -> join): n_proofs /
 * proofs_per_join joins of ONE shape, hence one key.  Join j verifies the shard proofs [j J, (j + 1) J) with their public values and is proven on
 * devices[j mod n_devices] (devices NULL / n_devices 0: every visible device), in_flight_per_device at a time on each (0: four) -- the joins are
 * independent units like the shards below them (no exchange step), and one join's host stretches overlap another's kernels (four joins of 16
 * headline shard proofs on one MI355X: 129 ms one after the other, 96 with two in flight, 76 with four).  Workers run on
 * pooled contexts (zkhip_release_cached_contexts) that keep the shape's proving key: setup once per context and shape, every context arrives
 * at the same vk (returned).  Join j's proof: joined + j joined_stride (stride >= zkhip_shard_verifier_proof_size), joined_lens[j] bytes.
 * verify != 0: every join is checked by zkhip_verify_shard_recursive on its worker's thread (sp1.rs:120).  n_proofs must be a multiple of
 * proofs_per_join (a caller with a remainder repeats its last shard proof, as zktls::compress_join_size describes).  Returns the status of the
 * lowest failing join; the proofs are the bytes zkhip_prove_shard_verifier makes. */
int zkhip_prove_shard_verifier_batch(const int* devices, int n_devices, const uint8_t* const* shard_proofs, const size_t* shard_proof_lens, size_t n_proofs,
                                     size_t proofs_per_join, int log_n, uint32_t width, const uint32_t* public_values, size_t n_public, const zkhip_params* inner,
                                     const zkhip_params* outer, int in_flight_per_device, int verify, uint8_t* joined, size_t joined_stride, size_t* joined_lens,
                                     uint32_t vk[8]);
int zkhip_verify_shard_recursive(const uint8_t* proof, size_t len, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, const uint32_t* public_values,
                                 size_t n_public, size_t n_proofs, const uint32_t vk[8], const zkhip_params* outer, int* reason);
/* ---- THE SAME MACHINE FOR INNER PROOFS OF A CONSTRAINT PROGRAM (round 5; version-7 proofs: zkhip_prove_shard_air, zkhip_prove_sha256, the chained
 * shards of zkhip_prove_sha256_sharded, zkhip_prove_shards_air_multi) -- the proofs that carry a REAL statement become compressible: what
 * `client.prove(.., SP1ProofMode::Groth16)` (crates/guest-prover-sp1/src/sp1.rs:116: core -> COMPRESS) does to the shard proofs of a real guest.
 * Conditions on the inner proofs: SP1 shape (blowup 2, fold by 2, constant final value, Poseidon2 width 16, no lookups), a width that is a
 * multiple of 8, a program of log_quotient_degree 1 whose terms have at most three factors (a selector counts), at most 128 public values.
 * A NINTH chip, EVAL, holds one row per TERM of the program: coefficient and three factor keys preprocessed, the factor values -- opened
 * values at zeta / zeta g from OPENED, public values from the transcript table, selectors from SCALARS -- received over one bus, the fold of
 * the constraints with alpha as a running sum down the rows.  The transcript starts from 18 constant words (the six shape words, logup_pairs,
 * fold, final, hash width, the program's digest).  THE KEY IS A FUNCTION OF THE SHAPE AND THE PROGRAM; the verifier of the outer proof takes
 * the program, the shape, the inner proofs' public values and the key -- no byte of an inner proof.  Joins as above (n_proofs > 1). */
int zkhip_shard_verifier_setup_air(zkhip_ctx* ctx, const uint32_t* program, size_t program_words, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public,
                                   size_t n_proofs, const zkhip_params* outer, zkhip_machine_key** key, uint32_t vk[8]);
int zkhip_shard_verifier_key_host_air(const uint32_t* program, size_t program_words, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public, size_t n_proofs,
                                      const zkhip_params* outer, uint32_t vk[8]);
size_t zkhip_shard_verifier_max_proofs_air(const uint32_t* program, size_t program_words, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public,
                                           const zkhip_params* outer);
size_t zkhip_shard_verifier_proof_size_air(const uint32_t* program, size_t program_words, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits, size_t n_public,
                                           size_t n_proofs, const zkhip_params* outer);
int zkhip_prove_shard_verifier_air(zkhip_ctx* ctx, const zkhip_machine_key* key, const uint32_t* program, size_t program_words, const uint8_t* const* shard_proofs,
                                   const size_t* shard_proof_lens, size_t n_proofs, int log_n, uint32_t width, const uint32_t* public_values, size_t n_public,
                                   const zkhip_params* inner, const zkhip_params* outer, uint8_t* proof, size_t cap, size_t* len);
int zkhip_verify_shard_recursive_air(const uint32_t* program, size_t program_words, const uint8_t* proof, size_t len, int log_n, uint32_t width, size_t n_queries, int inner_pow_bits,
                                     const uint32_t* public_values, size_t n_public, size_t n_proofs, const uint32_t vk[8], const zkhip_params* outer, int* reason);

/* ---- MACHINE MODE (round 5): the inner proofs are version-11 proofs of a KEYED MACHINE -- chips of mixed heights, each with its constraint
 * program (log_quotient_degree 1) and its interaction table, preprocessed columns committed by the inner key -- described by
 * zkhip_machine_desc.  The shard verifier machine's own outer proofs are such proofs (zkhip_shard_verifier_describe[_air] gives their
 * description, zkhip_shard_verifier_setup[_air] their key): with these entries the JOIN'S OUTPUT IS JOINABLE -- a tree of joins
 * (sp1.rs:116 core -> compress; prover.rs:90 lift -> join).  Ten chips (csrc/machine_verifier.inl; tests/recursion_machine.py restates
 * them): the transcript with gamma / beta, the permutation root and the cumulative sums; per chip its program and its LogUp constraints
 * at zeta with its own selectors and quotient; per query the four mixed-height commitments (the inner key's tree, main, permutation,
 * quotient: concatenated leaves, injection of the shorter matrices' rows), one reduced opening per height, FRI with the heights joining
 * on the way down; proof of work.  The key is a function of (the inner machine's description, n_proofs); the verifier takes the
 * description, the inner proofs' public values and the key -- no byte of an inner proof.  Inner proofs: blowup 2 (log_blowup 1), fold by
 * 2, constant final value; at most 16 chips, widths in multiples of 4, every chip with a program and a table, at most 64 proofs. */
typedef struct zkhip_machine_desc {
    int32_t n_chips;                      /* tallest first */
    const int32_t* log_ns;
    const uint32_t* widths;               /* main widths */
    const uint32_t* pre_widths;           /* 0: the chip has no preprocessed columns */
    const uint32_t* const* programs; const size_t* program_words;      /* over the combined row [preprocessed | main] */
    const uint32_t* const* tables; const size_t* table_words;
    uint32_t key_root[8];                 /* the inner machine's key (canonical words) */
    int32_t num_queries, pow_bits;        /* how its proofs are made */
    uint32_t n_public;
} zkhip_machine_desc;
int zkhip_machine_verifier_setup(zkhip_ctx* ctx, const zkhip_machine_desc* inner, size_t n_proofs, const zkhip_params* outer, zkhip_machine_key** key, uint32_t vk[8]);
int zkhip_machine_verifier_key_host(const zkhip_machine_desc* inner, size_t n_proofs, const zkhip_params* outer, uint32_t vk[8]);
size_t zkhip_machine_verifier_proof_size(const zkhip_machine_desc* inner, size_t n_proofs, const zkhip_params* outer);
int zkhip_prove_machine_verifier(zkhip_ctx* ctx, const zkhip_machine_key* key, const zkhip_machine_desc* inner, const uint8_t* const* proofs, const size_t* proof_lens, size_t n_proofs,
                                 const uint32_t* public_values, size_t n_public, const zkhip_params* outer, uint8_t* proof, size_t cap, size_t* len);
int zkhip_verify_machine_recursive(const zkhip_machine_desc* inner, const uint8_t* proof, size_t len, const uint32_t* public_values, size_t n_public, size_t n_proofs, const uint32_t vk[8],
                                   const zkhip_params* outer, int* reason);
/* THE TREE in one call (sp1.rs:116: core -> compress, the recursion tree; prover.rs:90: lift -> join): the shard proofs of an execution -> n_proofs /
 * proofs_per_join joins (as zkhip_prove_shard_verifier_batch makes them: join j on devices[j mod n_devices], in_flight_per_device at a time, pooled
 * contexts that keep the shape's key) -> ONE proof over the joins, proven on ctx under top_key = zkhip_machine_verifier_setup(ctx, join_machine,
 * n_joins, top_outer, ...).  join_machine describes the join machine (zkhip_shard_verifier_describe's eight chips, the join key as key_root,
 * join_outer's queries and proof-of-work bits, proofs_per_join x n_public public values).  A join's tables for the top -- which are its verification --
 * are filled on its worker's thread the moment the join exists, beside the joins still being proven: after the last join only the uploads and the
 * machine's proof remain.  The joins stay with the caller (joined + j joined_stride, joined_lens[j]; stride >= zkhip_shard_verifier_proof_size and a
 * multiple of 4), join_vk receives their key (it must equal join_machine->key_root).  The proof is the bytes zkhip_prove_machine_verifier makes from
 * the same joins; zkhip_verify_machine_recursive checks it from (join_machine, the shard proofs' public values, the top's key). */
int zkhip_prove_shard_tree(zkhip_ctx* ctx, const zkhip_machine_key* top_key, const zkhip_machine_desc* join_machine, const int* devices, int n_devices,
                           const uint8_t* const* shard_proofs, const size_t* shard_proof_lens, size_t n_proofs, size_t proofs_per_join, int log_n, uint32_t width,
                           const uint32_t* public_values, size_t n_public, const zkhip_params* inner, const zkhip_params* join_outer, const zkhip_params* top_outer,
                           int in_flight_per_device, uint8_t* joined, size_t joined_stride, size_t* joined_lens, uint32_t join_vk[8], uint8_t* proof, size_t cap, size_t* len);

/* the MAIN trace of the chip at position `which` as zkhip_prove_machine_verifier fills it on the host (canonical words, [2^log_rows][main width]; the
 * Poseidon2 chip: (input state [16], direction bit, KP) per used row -- its columns are the device's): no device, no context.  Filling the tables of an
 * inner proof is its verification: 0 (zkhip_last_error) for a proof the machine would not take.  For tests and for looking. */


/* ---- Poseidon2 parameter tables from a file: the built-in sets are this repo's own
 * ("zktls-amd/p2-bb16-v1", "...-bb24-v1"; the SP1 / RISC Zero tables of reference Cargo.lock:4030, 6172, 5057 are not
 * obtainable offline).  zkhip_load_poseidon2_params replaces the width-16 or the width-24 set (the file says which) for the
 * whole process -- provers, commitments, transcript and verifier alike -- without a rebuild.  File: JSON with "width" (16 | 24),
 * "name", "external_rc" (8 x width), "internal_rc" (13 | 21), "internal_diag" (width), canonical residues
 * (tests/golden/poseidon2*_params.json are such files).  Only while NO context exists (zkhip_ctx_destroy everything and
 * zkhip_release_cached_contexts first): a context uploads the set in effect to its device when it is created. ---- */
/* (not thread-safe against any other call into the library, host-only entries included: load at start-up) */
int zkhip_load_poseidon2_params(const char* path);
/* a counter that moves whenever the table set in effect changes (each successful load, each reset): a caller that keeps values derived from
 * the tables across calls -- a machine key, a verifying key -- stores it beside them and drops them when it differs */
uint64_t zkhip_poseidon2_params_generation(void);
int zkhip_reset_poseidon2_params(void);                 /* back to the built-in sets */
const char* zkhip_poseidon2_params_name(int width);     /* name of the set in effect (thread-local copy) */

/* ---- bincode-shaped form of a shard proof: the serde / bincode structure an upstream verifier
 * deserialises -- p3-uni-stark `Proof { commitments, opened_values, opening_proof: FriProof, degree_bits }` (reference
 * Cargo.lock:4055, 3930), which is what `proof.bytes()` carries at crates/guest-prover-sp1/src/sp1.rs:122-123.  Encoding rule:
 * bincode 1.x defaults (little-endian, u64 length prefix per Vec, arrays and structs bare).  The FIELD ORDER is [RECALLED]
 * (the crates are not in /root/reference): DESIGN.md section 6b spells it out.  Reader and writer are exact inverses on every
 * proof of zkhip_prove_shard / zkhip_prove_segment (versions 1-3).  Host only. ---- */
size_t zkhip_bincode_size(int log_n, uint32_t width, const zkhip_params* prm);
int zkhip_proof_to_bincode(const uint8_t* proof, size_t len, int log_n, uint32_t width, const zkhip_params* prm,
                           uint8_t* out, size_t cap, size_t* out_len);
int zkhip_proof_from_bincode(const uint8_t* in, size_t len, int log_n, uint32_t width, const zkhip_params* prm, size_t n_public,
                             uint8_t* proof, size_t cap, size_t* out_len);

/* The same for MULTI-CHIP proofs (versions 4-6, 9-11), in the shape of sp1-stark's `ShardProof` ([RECALLED]: reference Cargo.lock:6172 is not
 * in /root/reference): commitment { main, permutation?, quotient }, opened_values.chips[ { preprocessed{local,next}, main{local,next},
 * permutation{local,next}, quotient chunks, cumulative_sum, log_degree } ], opening_proof { fri_proof { commit_phase_commits,
 * query_proofs[ commit_phase_openings[ { sibling_value, opening_proof } ] ], final_poly, pow_witness }, query_openings[ per query one
 * BatchOpening { opened_values, opening_proof } per commitment round: preprocessed?, main, permutation?, quotient ] }, public_values --
 * preceded by this library's header words as a `Vec<u32>` envelope (upstream takes that information from the machine and the vk).
 * A multi-chip proof describes itself (its header names every chip), so no shape arguments: zkhip_chips_bincode_size(proof, len) is 0 for
 * anything that is not a complete multi-chip proof.  Writer and reader are exact inverses; the reader rebuilds the flat proof and
 * returns the public values (n_public receives their count).  Host code only. */
size_t zkhip_chips_bincode_size(const uint8_t* proof, size_t len);
int zkhip_chips_proof_to_bincode(const uint8_t* proof, size_t len, const uint32_t* public_values, size_t n_public, uint8_t* out, size_t cap, size_t* out_len);
int zkhip_chips_proof_from_bincode(const uint8_t* in, size_t len, uint8_t* proof, size_t cap, size_t* out_len,
                                   uint32_t* public_values, size_t public_cap, size_t* n_public);

/* Request digest: 8 canonical BabyBear words binding the guest input (the CBOR bytes of sp1.rs:108-109 / prover.rs:81-82) and
 * the guest program (ELF): Poseidon2 overwrite-mode sponge over 3-byte limbs, fields length-prefixed.  The glue uses them as the
 * leading public values of every shard of the request, so a proof does not transfer to another request.  Host only. */
int zkhip_request_digest(const uint8_t* input, size_t input_len, const uint8_t* program, size_t program_len, uint32_t out[8]);

/* intermediates of the last zkhip_prove_shard on this context (canonical words) */
typedef struct {
    uint32_t trace_root[8];
    uint32_t quotient_root[8];
    uint32_t alpha[4];
    uint32_t zeta[4];
    uint32_t fri_alpha[4];
    uint32_t pow_witness;
} zkhip_prove_debug;
/* Host verifiers hash the Merkle openings of 16 queries in lockstep, one query per AVX-512 lane, when the CPU has AVX-512 F + DQ
 * (checked at run time; otherwise query by query), and spread the query groups over up to 8 host threads.  This checks the batched
 * permutation against the scalar one: 1 = in use and equal, 0 = not available on this CPU, negative = mismatch; the optional outputs
 * receive the time per permutation of either form in nanoseconds. */
/* switch the batched form off (0) or back on (1) for the whole process; returns the previous setting.  For tests and A/B timing. */
int zkhip_host_simd(int enable);

#ifdef __cplusplus
}
#endif
#endif /* ZKHIP_H */
