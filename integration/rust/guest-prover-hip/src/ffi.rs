//! 1:1 declarations of include/zkhip.h (the subset the glue needs) and thin RAII wrappers.
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

use anyhow::{anyhow, Result};

#[repr(C)]
pub struct ZkhipShardJob {
    pub trace: *const u32,
    pub ld: usize,
    pub log_n: i32,
    pub width: u32,
    pub public_values: *const u32,
    pub n_public: usize,
    pub proof: *mut u8,
    pub proof_cap: usize,
    pub proof_len: usize,
    pub status: i32,
}

#[repr(C)]
pub struct ZkhipCtx {
    _private: [u8; 0],
}

/// include/zkhip.h `zkhip_params`; the last three fields select the FRI / hash shape (0 = SP1 defaults).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ZkhipParams {
    pub log_blowup: i32,
    pub num_queries: i32,
    pub pow_bits: i32,
    pub logup_pairs: i32,
    pub log_fold: i32,
    pub log_final: i32,
    pub hash_width: i32,
    /// RISC Zero's group order: > 0 commits the first `code_width` columns and the rest as two trees (code root first)
    pub code_width: i32,
}

impl ZkhipParams {
    /// sp1-stark 4.1.4 core shards
    pub const SP1_CORE: Self = Self { log_blowup: 1, num_queries: 100, pow_bits: 16, logup_pairs: 0, log_fold: 0, log_final: 0, hash_width: 0, code_width: 0 };
    /// risc0-zkp 1.2.5 segments: blowup 4, 50 queries, fold 16, 256 final coefficients, Poseidon2 width 24
    pub const RISC0: Self = Self { log_blowup: 2, num_queries: 50, pow_bits: 0, logup_pairs: 0, log_fold: 4, log_final: 8, hash_width: 24, code_width: 0 };
}

pub const ZKHIP_OK: c_int = 0;

extern "C" {
    pub fn zkhip_last_error() -> *const c_char;
    pub fn zkhip_device_count() -> c_int;
    pub fn zkhip_request_digest(input: *const u8, input_len: usize, program: *const u8, program_len: usize, out: *mut u32) -> c_int;
    /// all shards of one request over a device list, shard s on devices[s mod n] (null list + 0: every visible device)
    pub fn zkhip_prove_shards_multi(devices: *const c_int, n_devices: c_int, jobs: *mut ZkhipShardJob, n_jobs: c_int,
                                    prm: *const ZkhipParams, in_flight_per_device: c_int, host_traces: c_int) -> c_int;
    pub fn zkhip_prove_shards_air_multi(devices: *const c_int, n_devices: c_int, jobs: *mut ZkhipShardJob, n_jobs: c_int,
                                        program: *const u32, program_words: usize, prm: *const ZkhipParams, in_flight_per_device: c_int) -> c_int;
    pub fn zkhip_shard_device(shard_index: c_int, devices: *const c_int, n_devices: c_int) -> c_int;
    pub fn zkhip_ctx_create(device: c_int, stream: *mut c_void, out: *mut *mut ZkhipCtx) -> c_int;
    pub fn zkhip_ctx_destroy(ctx: *mut ZkhipCtx);
    pub fn zkhip_malloc(ctx: *mut ZkhipCtx, bytes: usize, d_ptr: *mut *mut c_void) -> c_int;
    pub fn zkhip_free(ctx: *mut ZkhipCtx, d_ptr: *mut c_void) -> c_int;
    pub fn zkhip_memcpy_h2d(ctx: *mut ZkhipCtx, d_dst: *mut c_void, h_src: *const c_void, bytes: usize) -> c_int;
    pub fn zkhip_to_monty(ctx: *mut ZkhipCtx, d_in: *const u32, d_out: *mut u32, n: usize) -> c_int;
    pub fn zkhip_gen_trace(ctx: *mut ZkhipCtx, seed: u64, shard: u64, log_n: c_int, width: u32, d_out: *mut u32, ld: usize) -> c_int;
    pub fn zkhip_gen_trace_logup(ctx: *mut ZkhipCtx, seed: u64, shard: u64, log_n: c_int, width: u32, pairs: c_int, d_out: *mut u32, ld: usize) -> c_int;
    pub fn zkhip_gen_trace_logup_cross(ctx: *mut ZkhipCtx, seed: u64, shard: u64, partner_shard: u64, log_n: c_int, width: u32, partner_width: u32, pairs: c_int,
                                       d_out: *mut u32, ld: usize) -> c_int;
    pub fn zkhip_air_synthetic(width: u32, n_public: usize, out: *mut u32, cap: usize, words: *mut usize) -> c_int;
    pub fn zkhip_machine_proof_size_keyed(log_ns: *const i32, widths: *const u32, pre_widths: *const u32, programs: *const *const u32, program_words: *const usize,
                                          tables: *const *const u32, table_words: *const usize, n_chips: c_int, prm: *const ZkhipParams, n_public: usize) -> usize;
    pub fn zkhip_proof_size(log_n: c_int, width: u32, prm: *const ZkhipParams, n_public: usize) -> usize;
    pub fn zkhip_prove_shard(
        ctx: *mut ZkhipCtx, d_trace: *const u32, ld: usize, log_n: c_int, width: u32,
        public_values: *const u32, n_public: usize, prm: *const ZkhipParams,
        proof: *mut u8, cap: usize, len: *mut usize,
    ) -> c_int;
    /// all shards of one execution, `in_flight` at a time on internal (cached) contexts; per-job status and length in `jobs`
    pub fn zkhip_prove_shards(
        device: c_int, jobs: *mut ZkhipShardJob, n_jobs: c_int, prm: *const ZkhipParams, in_flight: c_int, host_traces: c_int,
    ) -> c_int;
    pub fn zkhip_release_cached_contexts();
    pub fn zkhip_prove_segment(
        ctx: *mut ZkhipCtx, d_cols: *const u32, log_n: c_int, width: u32,
        public_values: *const u32, n_public: usize, prm: *const ZkhipParams,
        proof: *mut u8, cap: usize, len: *mut usize,
    ) -> c_int;
    pub fn zkhip_verify_shard(
        proof: *const u8, len: usize, log_n: c_int, width: u32,
        public_values: *const u32, n_public: usize, prm: *const ZkhipParams, reason: *mut c_int,
    ) -> c_int;
    // chips with their own constraint programs / the machine with interaction tables (include/zkhip.h; NULL entries: the synthetic AIR / no lookups)
    pub fn zkhip_prove_chips_air(
        ctx: *mut ZkhipCtx, chips: *const ZkhipChip, programs: *const *const u32, program_words: *const usize, n_chips: c_int,
        public_values: *const u32, n_public: usize, prm: *const ZkhipParams, proof: *mut u8, cap: usize, len: *mut usize,
    ) -> c_int;
    pub fn zkhip_machine_proof_size(
        log_ns: *const i32, widths: *const u32, programs: *const *const u32, program_words: *const usize,
        tables: *const *const u32, table_words: *const usize, n_chips: c_int, prm: *const ZkhipParams, n_public: usize,
    ) -> usize;
    pub fn zkhip_prove_machine(
        ctx: *mut ZkhipCtx, chips: *const ZkhipChip, programs: *const *const u32, program_words: *const usize,
        tables: *const *const u32, table_words: *const usize, n_chips: c_int,
        public_values: *const u32, n_public: usize, prm: *const ZkhipParams, proof: *mut u8, cap: usize, len: *mut usize,
    ) -> c_int;
    pub fn zkhip_verify_machine(
        proof: *const u8, len: usize, log_ns: *const i32, widths: *const u32, programs: *const *const u32, program_words: *const usize,
        tables: *const *const u32, table_words: *const usize, n_chips: c_int,
        public_values: *const u32, n_public: usize, prm: *const ZkhipParams, reason: *mut c_int,
    ) -> c_int;
    // the SHA-256 compression chip (include/zkhip.h): message in, digest and proof out
    pub fn zkhip_sha256_digest(message: *const u8, len: usize, digest: *mut u8);
    pub fn zkhip_sha256_proof_size(message_len: usize, prm: *const ZkhipParams) -> usize;
    pub fn zkhip_prove_sha256(
        ctx: *mut ZkhipCtx, message: *const u8, message_len: usize, prm: *const ZkhipParams, digest: *mut u8,
        proof: *mut u8, cap: usize, len: *mut usize,
    ) -> c_int;
    pub fn zkhip_verify_sha256(proof: *const u8, len: usize, digest: *const u8, message_len: u64, prm: *const ZkhipParams, reason: *mut c_int) -> c_int;
    // setup -> prove -> verify (sp1.rs:113, :116, :120): preprocessed columns committed once, the key's root is the verifying key
    pub fn zkhip_machine_setup(ctx: *mut ZkhipCtx, pre: *const ZkhipChip, n_chips: c_int, prm: *const ZkhipParams, key: *mut *mut ZkhipMachineKey, root: *mut u32) -> c_int;
    pub fn zkhip_machine_key_destroy(key: *mut ZkhipMachineKey);
    pub fn zkhip_prove_machine_keyed(
        ctx: *mut ZkhipCtx, key: *const ZkhipMachineKey, chips: *const ZkhipChip, programs: *const *const u32, program_words: *const usize,
        tables: *const *const u32, table_words: *const usize, n_chips: c_int,
        public_values: *const u32, n_public: usize, prm: *const ZkhipParams, proof: *mut u8, cap: usize, len: *mut usize,
    ) -> c_int;
    pub fn zkhip_prove_machine_keyed_at(
        ctx: *mut ZkhipCtx, key: *const ZkhipMachineKey, key_entries: *const i32, chips: *const ZkhipChip, programs: *const *const u32,
        program_words: *const usize, tables: *const *const u32, table_words: *const usize, n_chips: c_int,
        public_values: *const u32, n_public: usize, prm: *const ZkhipParams, proof: *mut u8, cap: usize, len: *mut usize,
    ) -> c_int;
    pub fn zkhip_verify_machine_keyed(
        proof: *const u8, len: usize, log_ns: *const i32, widths: *const u32, pre_widths: *const u32, root: *const u32,
        programs: *const *const u32, program_words: *const usize, tables: *const *const u32, table_words: *const usize, n_chips: c_int,
        public_values: *const u32, n_public: usize, prm: *const ZkhipParams, reason: *mut c_int,
    ) -> c_int;
    // the SHA-256 guest as a keyed machine (chip + preprocessed range table)
    pub fn zkhip_sha256_setup(ctx: *mut ZkhipCtx, prm: *const ZkhipParams, key: *mut *mut ZkhipMachineKey, vk: *mut u32) -> c_int;
    pub fn zkhip_sha256_machine_proof_size(message_len: usize, prm: *const ZkhipParams) -> usize;
    pub fn zkhip_prove_sha256_machine(
        ctx: *mut ZkhipCtx, key: *const ZkhipMachineKey, message: *const u8, message_len: usize, prm: *const ZkhipParams, digest: *mut u8,
        proof: *mut u8, cap: usize, len: *mut usize,
    ) -> c_int;
    pub fn zkhip_verify_sha256_machine(proof: *const u8, len: usize, digest: *const u8, message_len: u64, vk: *const u32, prm: *const ZkhipParams, reason: *mut c_int) -> c_int;
    // a batch of transcripts in one call (BASELINE configs[2]): job i on devices[i mod n], pooled contexts keep their proving key
    pub fn zkhip_prove_transcripts(
        devices: *const c_int, n_devices: c_int, jobs: *mut ZkhipTranscriptJob, n_jobs: c_int, prm: *const ZkhipParams,
        in_flight_per_device: c_int, verify: c_int, vk: *mut u32,
    ) -> c_int;
    // ... each job as zkhip_prove_sha256 makes it (version 7, no key): the form zkhip_prove_shard_verifier_air compresses into ONE proof
    pub fn zkhip_prove_transcripts_air(
        devices: *const c_int, n_devices: c_int, jobs: *mut ZkhipTranscriptJob, n_jobs: c_int, prm: *const ZkhipParams,
        in_flight_per_device: c_int, verify: c_int,
    ) -> c_int;
    // small jobs of a batch are proven in lock-step lanes (fibers of one thread per lane, merged kernel launches): members per batch
    // (0 / 1 = off; default 16) and lanes per device (default 6); same proof bytes either way
    pub fn zkhip_set_wait_mode(blocking: c_int, device: c_int) -> c_int;
    pub fn zkhip_set_lockstep(max_batch: c_int, lanes: c_int);
    pub fn zkhip_lockstep_stats(out: *mut u64);
    pub fn zkhip_lockstep_stack_high_water() -> u64;
    pub fn zkhip_set_fri_graph(on: c_int);
    // the shard verifier as a machine: whole shard proofs checked in-circuit (n_proofs of them by ONE outer proof); the key is a function of the shape
    pub fn zkhip_shard_verifier_setup(ctx: *mut ZkhipCtx, log_n: c_int, width: u32, n_queries: usize, inner_pow_bits: c_int, n_public: usize, n_proofs: usize,
                                      outer: *const ZkhipParams, key: *mut *mut ZkhipMachineKey, vk: *mut u32) -> c_int;
    pub fn zkhip_shard_verifier_proof_size(log_n: c_int, width: u32, n_queries: usize, inner_pow_bits: c_int, n_public: usize, n_proofs: usize, outer: *const ZkhipParams) -> usize;
    /// MACHINE MODE: the join's own output (a version-11 keyed-machine proof) verified in-circuit -- a tree of joins.  `inner` describes
    /// the inner machine (for the join machine: zkhip_shard_verifier_describe + the join key's root)
    pub fn zkhip_machine_verifier_setup(ctx: *mut ZkhipCtx, inner: *const ZkhipMachineDesc, n_proofs: usize, outer: *const ZkhipParams,
                                        key: *mut *mut ZkhipMachineKey, vk: *mut u32) -> c_int;
    pub fn zkhip_machine_verifier_key_host(inner: *const ZkhipMachineDesc, n_proofs: usize, outer: *const ZkhipParams, vk: *mut u32) -> c_int;
    pub fn zkhip_machine_verifier_proof_size(inner: *const ZkhipMachineDesc, n_proofs: usize, outer: *const ZkhipParams) -> usize;
    pub fn zkhip_prove_machine_verifier(
        ctx: *mut ZkhipCtx, key: *const ZkhipMachineKey, inner: *const ZkhipMachineDesc, proofs: *const *const u8, proof_lens: *const usize, n_proofs: usize,
        public_values: *const u32, n_public: usize, outer: *const ZkhipParams, proof: *mut u8, cap: usize, len: *mut usize,
    ) -> c_int;
    pub fn zkhip_prove_shard_tree(ctx: *mut ZkhipCtx, top_key: *const ZkhipMachineKey, join_machine: *const ZkhipMachineDesc, devices: *const c_int, n_devices: c_int,
                                  shard_proofs: *const *const u8, shard_proof_lens: *const usize, n_proofs: usize, proofs_per_join: usize, log_n: c_int, width: u32,
                                  public_values: *const u32, n_public: usize, inner: *const ZkhipParams, join_outer: *const ZkhipParams, top_outer: *const ZkhipParams,
                                  in_flight_per_device: c_int, joined: *mut u8, joined_stride: usize, joined_lens: *mut usize, join_vk: *mut u32, proof: *mut u8, cap: usize,
                                  len: *mut usize) -> c_int;
    pub fn zkhip_verify_machine_recursive(
        inner: *const ZkhipMachineDesc, proof: *const u8, len: usize, public_values: *const u32, n_public: usize, n_proofs: usize, vk: *const u32,
        outer: *const ZkhipParams, reason: *mut c_int,
    ) -> c_int;
    pub fn zkhip_poseidon2_params_generation() -> u64;
    /// the key WITHOUT a device (host cores only): what a verifier that owns no GPU derives for the shape it means
    pub fn zkhip_shard_verifier_key_host(log_n: c_int, width: u32, n_queries: usize, inner_pow_bits: c_int, n_public: usize, n_proofs: usize, outer: *const ZkhipParams,
                                         vk: *mut u32) -> c_int;
    pub fn zkhip_machine_key_host(h_traces: *const *const u32, log_ns: *const i32, pre_widths: *const u32, n_chips: c_int, prm: *const ZkhipParams, root: *mut u32) -> c_int;
    // the same machine for inner proofs of a constraint program (version 7: the SHA-256 chip's proofs ...): nine chips, the key = f(shape, program)
    pub fn zkhip_shard_verifier_setup_air(ctx: *mut ZkhipCtx, program: *const u32, program_words: usize, log_n: c_int, width: u32, n_queries: usize, inner_pow_bits: c_int,
                                          n_public: usize, n_proofs: usize, outer: *const ZkhipParams, key: *mut *mut ZkhipMachineKey, vk: *mut u32) -> c_int;
    pub fn zkhip_shard_verifier_key_host_air(program: *const u32, program_words: usize, log_n: c_int, width: u32, n_queries: usize, inner_pow_bits: c_int, n_public: usize,
                                             n_proofs: usize, outer: *const ZkhipParams, vk: *mut u32) -> c_int;
    pub fn zkhip_shard_verifier_max_proofs_air(program: *const u32, program_words: usize, log_n: c_int, width: u32, n_queries: usize, inner_pow_bits: c_int, n_public: usize,
                                               outer: *const ZkhipParams) -> usize;
    pub fn zkhip_shard_verifier_proof_size_air(program: *const u32, program_words: usize, log_n: c_int, width: u32, n_queries: usize, inner_pow_bits: c_int, n_public: usize,
                                               n_proofs: usize, outer: *const ZkhipParams) -> usize;
    pub fn zkhip_prove_shard_verifier_air(ctx: *mut ZkhipCtx, key: *const ZkhipMachineKey, program: *const u32, program_words: usize, shard_proofs: *const *const u8,
                                          shard_proof_lens: *const usize, n_proofs: usize, log_n: c_int, width: u32, public_values: *const u32, n_public: usize,
                                          inner: *const ZkhipParams, outer: *const ZkhipParams, proof: *mut u8, cap: usize, len: *mut usize) -> c_int;
    pub fn zkhip_verify_shard_recursive_air(program: *const u32, program_words: usize, proof: *const u8, len: usize, log_n: c_int, width: u32, n_queries: usize,
                                            inner_pow_bits: c_int, public_values: *const u32, n_public: usize, n_proofs: usize, vk: *const u32, outer: *const ZkhipParams,
                                            reason: *mut c_int) -> c_int;
    pub fn zkhip_shard_verifier_describe_air(program: *const u32, program_words: usize, log_n: c_int, width: u32, n_queries: usize, inner_pow_bits: c_int, n_public: usize,
                                             n_proofs: usize, which: c_int, kind: c_int, out: *mut u32, cap_words: usize, log_rows: *mut c_int, main_width: *mut u32,
                                             pre_width: *mut u32) -> usize;
    pub fn zkhip_shard_verifier_max_proofs(log_n: c_int, width: u32, n_queries: usize, inner_pow_bits: c_int, n_public: usize, outer: *const ZkhipParams) -> usize;
    pub fn zkhip_prove_shard_verifier(ctx: *mut ZkhipCtx, key: *const ZkhipMachineKey, shard_proofs: *const *const u8, shard_proof_lens: *const usize, n_proofs: usize,
                                      log_n: c_int, width: u32, public_values: *const u32, n_public: usize, inner: *const ZkhipParams, outer: *const ZkhipParams,
                                      proof: *mut u8, cap: usize, len: *mut usize) -> c_int;
    pub fn zkhip_prove_shard_verifier_batch(devices: *const c_int, n_devices: c_int, shard_proofs: *const *const u8, shard_proof_lens: *const usize, n_proofs: usize,
                                            proofs_per_join: usize, log_n: c_int, width: u32, public_values: *const u32, n_public: usize, inner: *const ZkhipParams,
                                            outer: *const ZkhipParams, in_flight_per_device: c_int, verify: c_int, joined: *mut u8, joined_stride: usize,
                                            joined_lens: *mut usize, vk: *mut u32) -> c_int;
    pub fn zkhip_verify_shard_recursive(proof: *const u8, len: usize, log_n: c_int, width: u32, n_queries: usize, inner_pow_bits: c_int, public_values: *const u32,
                                        n_public: usize, n_proofs: usize, vk: *const u32, outer: *const ZkhipParams, reason: *mut c_int) -> c_int;
    pub fn zkhip_shard_verifier_describe(log_n: c_int, width: u32, n_queries: usize, inner_pow_bits: c_int, n_public: usize, n_proofs: usize, which: c_int, kind: c_int,
                                         out: *mut u32, cap_words: usize, log_rows: *mut c_int, main_width: *mut u32, pre_width: *mut u32) -> usize;
    // the Poseidon2 permutation chip: Merkle openings (of whole rows when row_width > 0) proven in-circuit
    pub fn zkhip_p2chip_air(program: *mut u32, cap_words: usize) -> usize;
    pub fn zkhip_merkle_paths_proof_size(n_paths: usize, depth: c_int, row_width: u32, prm: *const ZkhipParams) -> usize;
    pub fn zkhip_prove_merkle_paths(
        ctx: *mut ZkhipCtx, leaves: *const u32, row_width: u32, siblings: *const u32, indices: *const u32, n_paths: usize, depth: c_int,
        root: *const u32, prm: *const ZkhipParams, proof: *mut u8, cap: usize, len: *mut usize,
    ) -> c_int;
    // SHA-256 of a message of any length as a chain of shard proofs dealt over the devices
    pub fn zkhip_sha256_sharded_count(message_len: usize, log_blocks_per_shard: c_int) -> usize;
    pub fn zkhip_sha256_shard_proof_size(log_blocks: c_int, prm: *const ZkhipParams) -> usize;
    pub fn zkhip_prove_sha256_sharded(
        devices: *const c_int, n_devices: c_int, message: *const u8, message_len: usize, log_blocks_per_shard: c_int, prm: *const ZkhipParams,
        in_flight_per_device: c_int, digest: *mut u8, chain: *mut u32, proofs: *mut u8, proof_stride: usize, proof_lens: *mut usize,
    ) -> c_int;
    pub fn zkhip_verify_sha256_sharded(
        proofs: *const u8, proof_stride: usize, proof_lens: *const usize, n_shards: usize, chain: *const u32, log_blocks_per_shard: c_int,
        digest: *const u8, message_len: u64, prm: *const ZkhipParams, bad_shard: *mut usize, reason: *mut c_int,
    ) -> c_int;
    // the chain as ONE proof (core -> compress on a statement about real data): the shards verified in-circuit (air mode of the shard verifier)
    pub fn zkhip_sha256_compress_setup(
        ctx: *mut ZkhipCtx, message_len: usize, log_blocks_per_shard: c_int, inner: *const ZkhipParams, outer: *const ZkhipParams,
        key: *mut *mut ZkhipMachineKey, vk: *mut u32,
    ) -> c_int;
    pub fn zkhip_sha256_compress_key_host(message_len: usize, log_blocks_per_shard: c_int, inner: *const ZkhipParams, outer: *const ZkhipParams, vk: *mut u32) -> c_int;
    pub fn zkhip_sha256_compressed_proof_size(message_len: usize, log_blocks_per_shard: c_int, inner: *const ZkhipParams, outer: *const ZkhipParams) -> usize;
    pub fn zkhip_prove_sha256_compressed(
        ctx: *mut ZkhipCtx, key: *const ZkhipMachineKey, devices: *const c_int, n_devices: c_int, message: *const u8, message_len: usize,
        log_blocks_per_shard: c_int, inner: *const ZkhipParams, outer: *const ZkhipParams, in_flight_per_device: c_int, digest: *mut u8,
        chain: *mut u32, proof: *mut u8, cap: usize, len: *mut usize,
    ) -> c_int;
    pub fn zkhip_verify_sha256_compressed(
        proof: *const u8, len: usize, digest: *const u8, message_len: u64, chain: *const u32, log_blocks_per_shard: c_int, vk: *const u32,
        inner: *const ZkhipParams, outer: *const ZkhipParams, reason: *mut c_int,
    ) -> c_int;
    pub fn zkhip_verify_merkle_paths(proof: *const u8, len: usize, root: *const u32, n_paths: usize, prm: *const ZkhipParams, reason: *mut c_int) -> c_int;
    // the first piece of the compress stage (sp1.rs:116): the FRI check of many shard proofs proven in-circuit by one call; the outer
    // proofs are verified with (vk, final value, challenger capacity) beside them
    pub fn zkhip_fri_indices_proof_size(layers: c_int, n_queries: usize, inner_pow_bits: c_int, prm: *const ZkhipParams) -> usize;
    pub fn zkhip_prove_fri_indices_batch(
        devices: *const c_int, n_devices: c_int, jobs: *mut ZkhipFriJob, n_jobs: c_int, log_n: c_int, width: u32, inner: *const ZkhipParams,
        outer: *const ZkhipParams, in_flight_per_device: c_int, verify: c_int,
    ) -> c_int;
    pub fn zkhip_verify_fri_indices(
        proof: *const u8, len: usize, layers: c_int, n_queries: usize, inner_pow_bits: c_int, final_value: *const u32, capacity: *const u32,
        vk: *const u32, prm: *const ZkhipParams, reason: *mut c_int,
    ) -> c_int;
}

/// one shard proof of a recursion batch (zkhip_fri_job)
#[repr(C)]
pub struct ZkhipFriJob {
    pub shard_proof: *const u8,
    pub shard_proof_len: usize,
    pub public_values: *const u32,
    pub n_public: usize,
    pub proof: *mut u8,
    pub proof_cap: usize,
    pub proof_len: usize,
    pub vk: [u32; 8],
    pub final_value: [u32; 4],
    pub capacity: [u32; 8],
    pub status: i32,
}

/// an inner keyed machine for the shard verifier's machine mode (zkhip_machine_desc)
#[repr(C)]
pub struct ZkhipMachineDesc {
    pub n_chips: i32,
    pub log_ns: *const i32,
    pub widths: *const u32,
    pub pre_widths: *const u32,
    pub programs: *const *const u32,
    pub program_words: *const usize,
    pub tables: *const *const u32,
    pub table_words: *const usize,
    pub key_root: [u32; 8],
    pub num_queries: i32,
    pub pow_bits: i32,
    pub n_public: u32,
}
/// one transcript of a batch (zkhip_transcript_job)
#[repr(C)]
pub struct ZkhipTranscriptJob {
    pub message: *const u8,
    pub message_len: usize,
    pub digest: [u8; 32],
    pub proof: *mut u8,
    pub proof_cap: usize,
    pub proof_len: usize,
    pub status: i32,
}

/// opaque proving key of a keyed machine (zkhip_machine_key)
#[repr(C)]
pub struct ZkhipMachineKey {
    _private: [u8; 0],
}

/// one table of a multi-chip shard (zkhip_chip)
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ZkhipChip {
    pub d_trace: *const u32,
    pub ld: usize,
    pub log_n: i32,
    pub width: u32,
    pub logup_pairs: i32,
    pub partner: i32,
}

/// status code -> anyhow error carrying the library's thread-local message
pub fn check(rc: c_int, what: &str) -> Result<()> {
    if rc == ZKHIP_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(zkhip_last_error()) }.to_string_lossy().into_owned();
    Err(anyhow!("{what}: zkhip error {rc}: {msg}"))
}

/// One device + one HIP stream + its workspaces.  Not shared between threads (the C library says so).
pub struct Context(*mut ZkhipCtx);

impl Context {
    pub fn new(device: i32) -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { zkhip_ctx_create(device, std::ptr::null_mut(), &mut raw) }, "zkhip_ctx_create")?;
        Ok(Self(raw))
    }
    pub fn raw(&self) -> *mut ZkhipCtx {
        self.0
    }
    pub fn alloc(&self, words: usize) -> Result<DeviceBuffer<'_>> {
        let mut p = std::ptr::null_mut();
        check(unsafe { zkhip_malloc(self.0, words * 4, &mut p) }, "zkhip_malloc")?;
        Ok(DeviceBuffer { ctx: self, ptr: p as *mut u32, words })
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { zkhip_ctx_destroy(self.0) }
    }
}

pub struct DeviceBuffer<'a> {
    ctx: &'a Context,
    pub ptr: *mut u32,
    pub words: usize,
}

impl DeviceBuffer<'_> {
    /// canonical host words -> Montgomery device words (the in-memory form of p3 `MontyField31`)
    pub fn upload_canonical(&self, host: &[u32]) -> Result<()> {
        anyhow::ensure!(host.len() == self.words, "upload size mismatch");
        check(unsafe { zkhip_memcpy_h2d(self.ctx.raw(), self.ptr as *mut c_void, host.as_ptr() as *const c_void, host.len() * 4) }, "zkhip_memcpy_h2d")?;
        check(unsafe { zkhip_to_monty(self.ctx.raw(), self.ptr, self.ptr, self.words) }, "zkhip_to_monty")
    }
}

impl Drop for DeviceBuffer<'_> {
    fn drop(&mut self) {
        unsafe {
            zkhip_free(self.ctx.raw(), self.ptr as *mut c_void);
        }
    }
}
