"""ctypes binding of libzkhip.so (C ABI: include/zkhip.h).

The product path: everything here calls the HIP library.  There is no CPU fallback --
if the shared library is missing, or no gfx950 device is visible, the calls raise.
"""
import ctypes as C
import os

import numpy as np

P = 2013265921
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzkhip.so")

u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)

EXPORTS = [
    "zkhip_version", "zkhip_last_error", "zkhip_device_count", "zkhip_ctx_create", "zkhip_ctx_destroy",
    "zkhip_ctx_sync", "zkhip_ctx_set_lde_fusion", "zkhip_ctx_stream", "zkhip_malloc", "zkhip_free", "zkhip_memcpy_h2d",
    "zkhip_memcpy_d2h", "zkhip_to_monty", "zkhip_from_monty", "zkhip_fill_uniform", "zkhip_gen_trace",
    "zkhip_gen_trace_logup", "zkhip_gen_trace_logup_cross", "zkhip_perm_trace",
    "zkhip_dft", "zkhip_coset_lde", "zkhip_ntt_pass", "zkhip_poseidon2_permute", "zkhip_hash_rows",
    "zkhip_merkle_commit", "zkhip_merkle_commit_mixed", "zkhip_merkle_commit_p24_colmajor", "zkhip_batch_interpolate_colmajor", "zkhip_batch_expand_colmajor", "zkhip_quotient_values", "zkhip_open_at", "zkhip_fri_fold", "zkhip_fri_fold_k",
    "zkhip_commit", "zkhip_proof_size", "zkhip_prove_shard", "zkhip_prove_shard_host", "zkhip_prove_shards", "zkhip_prove_shards_multi", "zkhip_shard_device", "zkhip_release_cached_contexts", "zkhip_prove_segment", "zkhip_verify_shard", "zkhip_last_prove_debug",
    "zkhip_chips_proof_size", "zkhip_prove_chips", "zkhip_verify_chips", "zkhip_request_digest",
    "zkhip_eltwise_add", "zkhip_eltwise_copy", "zkhip_eltwise_zeroize", "zkhip_eltwise_sum_ext", "zkhip_zk_shift", "zkhip_mix_poly_coeffs",
    "zkhip_batch_evaluate_any", "zkhip_gather_sample", "zkhip_scatter", "zkhip_prefix_products_ext", "zkhip_hash_rows_sha256",
    "zkhip_hash_fold_sha256", "zkhip_merkle_commit_sha256_colmajor",
    "zkhip_bincode_size", "zkhip_proof_to_bincode", "zkhip_proof_from_bincode",
    "zkhip_chips_bincode_size", "zkhip_chips_proof_to_bincode", "zkhip_chips_proof_from_bincode",
    "zkhip_load_poseidon2_params", "zkhip_reset_poseidon2_params", "zkhip_poseidon2_params_name",
    "zkhip_air_validate", "zkhip_air_digest", "zkhip_air_synthetic", "zkhip_proof_size_air", "zkhip_prove_shard_air", "zkhip_verify_shard_air",
    "zkhip_quotient_values_air",
    "zkhip_chips_proof_size_air", "zkhip_prove_chips_air", "zkhip_verify_chips_air",
    "zkhip_prove_shards_air_multi", "zkhip_selftest_host_simd", "zkhip_host_permutation_ns", "zkhip_host_simd", "zkhip_recursion_witnesses_on_host", "zkhip_machine_proof_size", "zkhip_prove_machine", "zkhip_verify_machine", "zkhip_range_table",
    "zkhip_machine_setup", "zkhip_machine_key_destroy", "zkhip_machine_proof_size_keyed", "zkhip_prove_machine_keyed", "zkhip_verify_machine_keyed", "zkhip_prove_machine_keyed_at",
    "zkhip_sha256_setup", "zkhip_sha256_machine_proof_size", "zkhip_prove_sha256_machine", "zkhip_verify_sha256_machine", "zkhip_prove_transcripts", "zkhip_prove_transcripts_air", "zkhip_sha256_machine_describe", "zkhip_machine_verifier_setup", "zkhip_machine_verifier_key_host", "zkhip_machine_verifier_proof_size", "zkhip_prove_machine_verifier", "zkhip_prove_shard_tree", "zkhip_verify_machine_recursive", "zkhip_machine_verifier_describe", "zkhip_machine_verifier_host_tables", "zkhip_sha256_compress_setup", "zkhip_sha256_compress_key_host", "zkhip_sha256_compressed_proof_size", "zkhip_prove_sha256_compressed", "zkhip_verify_sha256_compressed", "zkhip_set_wait_mode", "zkhip_set_lockstep", "zkhip_lockstep_stats", "zkhip_lockstep_stack_high_water", "zkhip_set_fri_graph", "zkhip_shard_verifier_setup", "zkhip_shard_verifier_proof_size", "zkhip_shard_verifier_max_proofs", "zkhip_prove_shard_verifier", "zkhip_prove_shard_verifier_batch", "zkhip_verify_shard_recursive", "zkhip_shard_verifier_describe", "zkhip_shard_verifier_key_host", "zkhip_machine_key_host", "zkhip_shard_verifier_setup_air", "zkhip_shard_verifier_key_host_air", "zkhip_shard_verifier_max_proofs_air", "zkhip_shard_verifier_proof_size_air", "zkhip_prove_shard_verifier_air", "zkhip_verify_shard_recursive_air", "zkhip_shard_verifier_describe_air", "zkhip_poseidon2_params_generation", "zkhip_selftest_lockstep",
    "zkhip_sha256_air_chained", "zkhip_sha256_gen_trace_chained", "zkhip_sha256_sharded_count", "zkhip_sha256_shard_proof_size", "zkhip_prove_sha256_sharded",
    "zkhip_verify_sha256_sharded",
    "zkhip_fri_view_shard", "zkhip_fri_chip_width", "zkhip_fri_chip_air",
    
    "zkhip_fri_view_path_words", "zkhip_fri_view_shard_paths", "zkhip_fri_view_transcript", "zkhip_fri_layers_chip_air", "zkhip_p2chip_air_fri_layers",
    
    "zkhip_fri_transcript_chip_air", "zkhip_p2chip_air_fri_transcript",
    
    "zkhip_fri_indices_program", "zkhip_fri_indices_key", "zkhip_fri_indices_proof_size", "zkhip_prove_fri_indices", "zkhip_verify_fri_indices",
    "zkhip_prove_fri_indices_batch", "zkhip_fri_view_all",
    "zkhip_p2chip_air", "zkhip_p2chip_gen_merkle_trace", "zkhip_merkle_paths_proof_size", "zkhip_prove_merkle_paths", "zkhip_verify_merkle_paths",
    "zkhip_sha256_air", "zkhip_sha256_digest", "zkhip_sha256_pad", "zkhip_sha256_padding_publics", "zkhip_sha256_gen_trace", "zkhip_sha256_proof_size", "zkhip_prove_sha256", "zkhip_verify_sha256",
]


class ZkHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("zkhip error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("log_blowup", C.c_int32), ("num_queries", C.c_int32), ("pow_bits", C.c_int32), ("logup_pairs", C.c_int32),
                ("log_fold", C.c_int32), ("log_final", C.c_int32), ("hash_width", C.c_int32), ("code_width", C.c_int32)]


def segment_params(num_queries=50, logup_pairs=0, log_final=8, code_width=0):
    """RISC-Zero-like shape (include/zkhip.h): blowup 4, fold by 16, 2^log_final final coefficients,
    Poseidon2 width 24, no proof of work; code_width > 0: code / data(/ accum) group commitments."""
    return Params(2, num_queries, 0, logup_pairs, 4, log_final, 24, code_width)


class Chip(C.Structure):
    _fields_ = [("d_trace", C.c_void_p), ("ld", C.c_size_t), ("log_n", C.c_int32), ("width", C.c_uint32), ("logup_pairs", C.c_int32), ("partner", C.c_int32)]


class MachineDesc(C.Structure):
    """zkhip_machine_desc: an inner keyed machine for the shard verifier's machine mode (include/zkhip.h)"""
    _fields_ = [("n_chips", C.c_int32), ("log_ns", C.POINTER(C.c_int32)), ("widths", C.POINTER(C.c_uint32)), ("pre_widths", C.POINTER(C.c_uint32)),
                ("programs", C.POINTER(C.POINTER(C.c_uint32))), ("program_words", C.POINTER(C.c_size_t)), ("tables", C.POINTER(C.POINTER(C.c_uint32))),
                ("table_words", C.POINTER(C.c_size_t)), ("key_root", C.c_uint32 * 8), ("num_queries", C.c_int32), ("pow_bits", C.c_int32), ("n_public", C.c_uint32)]


class TranscriptJob(C.Structure):
    _fields_ = [("message", u8p), ("message_len", C.c_size_t), ("digest", C.c_uint8 * 32), ("proof", u8p), ("proof_cap", C.c_size_t),
                ("proof_len", C.c_size_t), ("status", C.c_int32)]


class FriJob(C.Structure):
    _fields_ = [("shard_proof", u8p), ("shard_proof_len", C.c_size_t), ("public_values", u32p), ("n_public", C.c_size_t), ("proof", u8p),
                ("proof_cap", C.c_size_t), ("proof_len", C.c_size_t), ("vk", C.c_uint32 * 8), ("final_value", C.c_uint32 * 4),
                ("capacity", C.c_uint32 * 8), ("status", C.c_int32)]


class ShardJob(C.Structure):
    _fields_ = [("trace", C.c_void_p), ("ld", C.c_size_t), ("log_n", C.c_int32), ("width", C.c_uint32), ("public_values", u32p),
                ("n_public", C.c_size_t), ("proof", u8p), ("proof_cap", C.c_size_t), ("proof_len", C.c_size_t), ("status", C.c_int32)]


class ProveDebug(C.Structure):
    _fields_ = [
        ("trace_root", C.c_uint32 * 8),
        ("quotient_root", C.c_uint32 * 8),
        ("alpha", C.c_uint32 * 4),
        ("zeta", C.c_uint32 * 4),
        ("fri_alpha", C.c_uint32 * 4),
        ("pow_witness", C.c_uint32),
    ]


_LIB = None


def load():
    """Load libzkhip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError("libzkhip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(zktls_amd has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    L.zkhip_last_error.restype = C.c_char_p
    L.zkhip_ctx_stream.restype = C.c_void_p
    L.zkhip_ctx_stream.argtypes = [C.c_void_p]
    L.zkhip_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.zkhip_ctx_destroy.argtypes = [C.c_void_p]
    L.zkhip_ctx_destroy.restype = None
    L.zkhip_ctx_sync.argtypes = [C.c_void_p]
    L.zkhip_ctx_set_lde_fusion.argtypes = [C.c_void_p, C.c_int]
    L.zkhip_malloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.zkhip_free.argtypes = [C.c_void_p, C.c_void_p]
    L.zkhip_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.zkhip_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.zkhip_to_monty.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.zkhip_from_monty.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.zkhip_fill_uniform.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_void_p, C.c_size_t]
    L.zkhip_gen_trace.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, C.c_void_p, C.c_size_t]
    L.zkhip_gen_trace_logup.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t]
    L.zkhip_perm_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_int, u32p, u32p, C.c_void_p]
    L.zkhip_dft.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_int, C.c_int]
    L.zkhip_coset_lde.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_int, C.c_uint32]
    L.zkhip_ntt_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_int]
    L.zkhip_poseidon2_permute.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.zkhip_hash_rows.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.c_int, C.c_size_t, C.c_void_p]
    L.zkhip_merkle_commit.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_void_p]
    L.zkhip_merkle_commit_mixed.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.c_int, C.c_void_p]
    L.zkhip_merkle_commit_p24_colmajor.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    L.zkhip_batch_interpolate_colmajor.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    L.zkhip_batch_expand_colmajor.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_uint32]
    L.zkhip_quotient_values.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, u32p, C.c_void_p]
    L.zkhip_open_at.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_uint32, u32p, C.c_int, u32p]
    L.zkhip_fri_fold.argtypes = [C.c_void_p, C.c_void_p, C.c_int, u32p, C.c_void_p]
    L.zkhip_fri_fold_k.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, u32p, C.c_void_p]
    L.zkhip_proof_size.restype = C.c_size_t
    L.zkhip_proof_size.argtypes = [C.c_int, C.c_uint32, C.POINTER(Params), C.c_size_t]
    L.zkhip_prove_shard.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, u32p, C.c_size_t,
                                    C.POINTER(Params), u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zkhip_commit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, u32p]
    L.zkhip_prove_shard_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, u32p, C.c_size_t,
                                         C.POINTER(Params), u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zkhip_release_cached_contexts.restype = None
    L.zkhip_prove_shards.argtypes = [C.c_int, C.POINTER(ShardJob), C.c_int, C.POINTER(Params), C.c_int, C.c_int]
    L.zkhip_prove_shards_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(ShardJob), C.c_int, C.POINTER(Params), C.c_int, C.c_int]
    vp, sz = C.c_void_p, C.c_size_t
    L.zkhip_eltwise_add.argtypes = [vp, vp, vp, vp, sz]
    L.zkhip_eltwise_copy.argtypes = [vp, vp, vp, sz]
    L.zkhip_eltwise_zeroize.argtypes = [vp, vp, sz]
    L.zkhip_eltwise_sum_ext.argtypes = [vp, vp, vp, sz, sz]
    L.zkhip_zk_shift.argtypes = [vp, vp, sz, C.c_int, C.c_uint32]
    L.zkhip_mix_poly_coeffs.argtypes = [vp, vp, u32p, u32p, vp, vp, sz, sz, C.c_int]
    L.zkhip_batch_evaluate_any.argtypes = [vp, vp, C.c_int, vp, vp, vp, sz, C.c_int]
    L.zkhip_gather_sample.argtypes = [vp, vp, vp, sz, sz, sz]
    L.zkhip_scatter.argtypes = [vp, vp, vp, vp, vp, sz]
    L.zkhip_prefix_products_ext.argtypes = [vp, vp, sz, C.c_int]
    L.zkhip_hash_rows_sha256.argtypes = [vp, vp, sz, sz, vp]
    L.zkhip_hash_fold_sha256.argtypes = [vp, vp, vp, sz]
    L.zkhip_merkle_commit_sha256_colmajor.argtypes = [vp, vp, C.c_uint32, C.c_int, vp]
    L.zkhip_air_validate.argtypes = [u32p, C.c_size_t, C.c_uint32, C.c_size_t]
    L.zkhip_air_digest.argtypes = [u32p, C.c_size_t, u32p]
    L.zkhip_air_synthetic.argtypes = [C.c_uint32, C.c_size_t, u32p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zkhip_proof_size_air.restype = C.c_size_t
    L.zkhip_proof_size_air.argtypes = [u32p, C.c_size_t, C.c_int, C.c_uint32, C.POINTER(Params), C.c_size_t]
    L.zkhip_prove_shard_air.argtypes = [C.c_void_p, u32p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, u32p, C.c_size_t,
                                        C.POINTER(Params), u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zkhip_verify_shard_air.argtypes = [u32p, C.c_size_t, u8p, C.c_size_t, C.c_int, C.c_uint32, u32p, C.c_size_t, C.POINTER(Params), C.POINTER(C.c_int)]
    L.zkhip_quotient_values_air.argtypes = [C.c_void_p, u32p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, u32p, C.c_size_t, u32p, C.c_void_p]
    u32pp, szp = C.POINTER(u32p), C.POINTER(C.c_size_t)
    L.zkhip_chips_proof_size_air.restype = C.c_size_t
    L.zkhip_chips_proof_size_air.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_uint32), u32pp, szp, C.c_int, C.POINTER(Params), C.c_size_t]
    L.zkhip_prove_chips_air.argtypes = [C.c_void_p, C.POINTER(Chip), u32pp, szp, C.c_int, u32p, C.c_size_t, C.POINTER(Params), u8p, C.c_size_t, szp]
    L.zkhip_verify_chips_air.argtypes = [u8p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), u32pp, szp, C.c_int, u32p, C.c_size_t,
                                         C.POINTER(Params), C.POINTER(C.c_int)]
    L.zkhip_prove_shards_air_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(ShardJob), C.c_int, u32p, C.c_size_t, C.POINTER(Params), C.c_int]
    L.zkhip_selftest_host_simd.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.zkhip_host_permutation_ns.argtypes = [C.c_int]
    L.zkhip_host_permutation_ns.restype = C.c_double
    L.zkhip_range_table.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                    C.c_uint32, C.c_uint32]
    L.zkhip_machine_proof_size.restype = C.c_size_t
    L.zkhip_machine_proof_size.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_uint32), u32pp, szp, u32pp, szp, C.c_int, C.POINTER(Params), C.c_size_t]
    L.zkhip_prove_machine.argtypes = [C.c_void_p, C.POINTER(Chip), u32pp, szp, u32pp, szp, C.c_int, u32p, C.c_size_t, C.POINTER(Params), u8p, C.c_size_t, szp]
    L.zkhip_verify_machine.argtypes = [u8p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), u32pp, szp, u32pp, szp, C.c_int, u32p, C.c_size_t,
                                       C.POINTER(Params), C.POINTER(C.c_int)]
    L.zkhip_machine_setup.argtypes = [C.c_void_p, C.POINTER(Chip), C.c_int, C.POINTER(Params), C.POINTER(C.c_void_p), u32p]
    L.zkhip_machine_key_destroy.restype = None
    L.zkhip_machine_key_destroy.argtypes = [C.c_void_p]
    L.zkhip_machine_proof_size_keyed.restype = C.c_size_t
    L.zkhip_machine_proof_size_keyed.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), u32pp, szp, u32pp, szp, C.c_int,
                                                 C.POINTER(Params), C.c_size_t]
    L.zkhip_prove_machine_keyed.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Chip), u32pp, szp, u32pp, szp, C.c_int, u32p, C.c_size_t, C.POINTER(Params),
                                            u8p, C.c_size_t, szp]
    L.zkhip_verify_machine_keyed.argtypes = [u8p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), u32p, u32pp, szp, u32pp, szp,
                                             C.c_int, u32p, C.c_size_t, C.POINTER(Params), C.POINTER(C.c_int)]
    L.zkhip_prove_machine_keyed_at.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(Chip), u32pp, szp, u32pp, szp, C.c_int, u32p, C.c_size_t,
                                               C.POINTER(Params), u8p, C.c_size_t, szp]
    L.zkhip_sha256_setup.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(C.c_void_p), u32p]
    L.zkhip_sha256_machine_proof_size.restype = C.c_size_t
    L.zkhip_sha256_machine_proof_size.argtypes = [C.c_size_t, C.POINTER(Params)]
    L.zkhip_prove_sha256_machine.argtypes = [C.c_void_p, C.c_void_p, u8p, C.c_size_t, C.POINTER(Params), u8p, u8p, C.c_size_t, szp]
    L.zkhip_verify_sha256_machine.argtypes = [u8p, C.c_size_t, u8p, C.c_uint64, u32p, C.POINTER(Params), C.POINTER(C.c_int)]
    L.zkhip_chips_bincode_size.restype = C.c_size_t
    L.zkhip_chips_bincode_size.argtypes = [u8p, C.c_size_t]
    L.zkhip_chips_proof_to_bincode.argtypes = [u8p, C.c_size_t, u32p, C.c_size_t, u8p, C.c_size_t, szp]
    L.zkhip_chips_proof_from_bincode.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, szp, u32p, C.c_size_t, szp]
    L.zkhip_set_wait_mode.argtypes = [C.c_int, C.c_int]
    L.zkhip_set_wait_mode.restype = C.c_int
    L.zkhip_selftest_lockstep.argtypes = [C.c_int, C.c_int]
    L.zkhip_selftest_lockstep.restype = C.c_int
    L.zkhip_set_lockstep.argtypes = [C.c_int, C.c_int]
    L.zkhip_set_lockstep.restype = None
    L.zkhip_lockstep_stats.argtypes = [C.POINTER(C.c_uint64)]
    L.zkhip_lockstep_stats.restype = None
    L.zkhip_shard_verifier_setup.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(Params), C.POINTER(C.c_void_p), u32p]
    L.zkhip_poseidon2_params_generation.argtypes = []
    L.zkhip_poseidon2_params_generation.restype = C.c_uint64
    L.zkhip_shard_verifier_key_host.argtypes = [C.c_int, C.c_uint32, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(Params), u32p]
    L.zkhip_machine_key_host.argtypes = [C.POINTER(u32p), C.POINTER(C.c_int32), u32p, C.c_int, C.POINTER(Params), u32p]
    L.zkhip_shard_verifier_setup_air.argtypes = [C.c_void_p, u32p, C.c_size_t, C.c_int, C.c_uint32, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(Params), C.POINTER(C.c_void_p), u32p]
    L.zkhip_shard_verifier_key_host_air.argtypes = [u32p, C.c_size_t, C.c_int, C.c_uint32, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(Params), u32p]
    L.zkhip_shard_verifier_max_proofs_air.argtypes = [u32p, C.c_size_t, C.c_int, C.c_uint32, C.c_size_t, C.c_int, C.c_size_t, C.POINTER(Params)]
    L.zkhip_shard_verifier_max_proofs_air.restype = C.c_size_t
    L.zkhip_shard_verifier_proof_size_air.argtypes = [u32p, C.c_size_t, C.c_int, C.c_uint32, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(Params)]
    L.zkhip_shard_verifier_proof_size_air.restype = C.c_size_t
    L.zkhip_prove_shard_verifier_air.argtypes = [C.c_void_p, C.c_void_p, u32p, C.c_size_t, C.POINTER(u8p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_uint32, u32p, C.c_size_t, C.POINTER(Params),
                                                 C.POINTER(Params), u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zkhip_verify_shard_recursive_air.argtypes = [u32p, C.c_size_t, u8p, C.c_size_t, C.c_int, C.c_uint32, C.c_size_t, C.c_int, u32p, C.c_size_t, C.c_size_t, u32p, C.POINTER(Params), C.POINTER(C.c_int)]
    L.zkhip_shard_verifier_describe_air.argtypes = [u32p, C.c_size_t, C.c_int, C.c_uint32, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, u32p, C.c_size_t, C.POINTER(C.c_int),
                                                    C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.zkhip_shard_verifier_describe_air.restype = C.c_size_t
    L.zkhip_shard_verifier_proof_size.argtypes = [C.c_int, C.c_uint32, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(Params)]
    L.zkhip_shard_verifier_proof_size.restype = C.c_size_t
    L.zkhip_shard_verifier_max_proofs.argtypes = [C.c_int, C.c_uint32, C.c_size_t, C.c_int, C.c_size_t, C.POINTER(Params)]
    L.zkhip_shard_verifier_max_proofs.restype = C.c_size_t
    L.zkhip_prove_shard_verifier_batch.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(u8p), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t, C.c_int, C.c_uint32, u32p, C.c_size_t,
                                                   C.POINTER(Params), C.POINTER(Params), C.c_int, C.c_int, u8p, C.c_size_t, C.POINTER(C.c_size_t), u32p]
    L.zkhip_prove_shard_verifier.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(u8p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_uint32, u32p, C.c_size_t, C.POINTER(Params),
                                             C.POINTER(Params), u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zkhip_verify_shard_recursive.argtypes = [u8p, C.c_size_t, C.c_int, C.c_uint32, C.c_size_t, C.c_int, u32p, C.c_size_t, C.c_size_t, u32p, C.POINTER(Params), C.POINTER(C.c_int)]
    L.zkhip_shard_verifier_describe.argtypes = [C.c_int, C.c_uint32, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, u32p, C.c_size_t, C.POINTER(C.c_int),
                                                C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.zkhip_shard_verifier_describe.restype = C.c_size_t
    L.zkhip_set_fri_graph.argtypes = [C.c_int]
    L.zkhip_set_fri_graph.restype = None
    L.zkhip_lockstep_stack_high_water.argtypes = []
    L.zkhip_lockstep_stack_high_water.restype = C.c_uint64
    L.zkhip_prove_transcripts.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(TranscriptJob), C.c_int, C.POINTER(Params), C.c_int, C.c_int, u32p]
    L.zkhip_prove_transcripts_air.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(TranscriptJob), C.c_int, C.POINTER(Params), C.c_int, C.c_int]
    PP = C.POINTER(Params)
    MD = C.POINTER(MachineDesc)
    L.zkhip_sha256_machine_describe.argtypes = [C.c_size_t, C.c_int, C.c_int, u32p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.zkhip_sha256_machine_describe.restype = C.c_size_t
    L.zkhip_machine_verifier_setup.argtypes = [C.c_void_p, MD, C.c_size_t, PP, C.POINTER(C.c_void_p), u32p]
    L.zkhip_machine_verifier_key_host.argtypes = [MD, C.c_size_t, PP, u32p]
    L.zkhip_prove_shard_tree.argtypes = [C.c_void_p, C.c_void_p, MD, C.POINTER(C.c_int), C.c_int, C.POINTER(u8p), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t, C.c_int, C.c_uint32, u32p,
                                         C.c_size_t, PP, PP, PP, C.c_int, u8p, C.c_size_t, C.POINTER(C.c_size_t), u32p, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zkhip_machine_verifier_proof_size.argtypes = [MD, C.c_size_t, PP]
    L.zkhip_machine_verifier_proof_size.restype = C.c_size_t
    L.zkhip_prove_machine_verifier.argtypes = [C.c_void_p, C.c_void_p, MD, C.POINTER(u8p), C.POINTER(C.c_size_t), C.c_size_t, u32p, C.c_size_t, PP, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zkhip_verify_machine_recursive.argtypes = [MD, u8p, C.c_size_t, u32p, C.c_size_t, C.c_size_t, u32p, PP, C.POINTER(C.c_int)]
    L.zkhip_machine_verifier_describe.argtypes = [MD, C.c_size_t, C.c_int, C.c_int, u32p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.zkhip_machine_verifier_describe.restype = C.c_size_t
    L.zkhip_machine_verifier_host_tables.argtypes = [MD, C.POINTER(u8p), C.POINTER(C.c_size_t), C.c_size_t, u32p, C.c_size_t, C.c_int, u32p, C.c_size_t]
    L.zkhip_machine_verifier_host_tables.restype = C.c_size_t
    L.zkhip_sha256_compress_setup.argtypes = [C.c_void_p, C.c_size_t, C.c_int, PP, PP, C.POINTER(C.c_void_p), u32p]
    L.zkhip_sha256_compress_key_host.argtypes = [C.c_size_t, C.c_int, PP, PP, u32p]
    L.zkhip_sha256_compressed_proof_size.argtypes = [C.c_size_t, C.c_int, PP, PP]
    L.zkhip_sha256_compressed_proof_size.restype = C.c_size_t
    L.zkhip_prove_sha256_compressed.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, u8p, C.c_size_t, C.c_int, PP, PP, C.c_int, u8p, u32p, u8p, C.c_size_t,
                                                C.POINTER(C.c_size_t)]
    L.zkhip_verify_sha256_compressed.argtypes = [u8p, C.c_size_t, u8p, C.c_uint64, u32p, C.c_int, u32p, PP, PP, C.POINTER(C.c_int)]
    L.zkhip_fri_view_shard.argtypes = [u8p, C.c_size_t, C.c_int, C.c_uint32, u32p, C.c_size_t, C.POINTER(Params), u32p, u32p, u32p, u32p, u32p]
    L.zkhip_fri_chip_width.restype = C.c_uint32
    L.zkhip_fri_chip_width.argtypes = [C.c_int]
    L.zkhip_fri_chip_air.restype = C.c_size_t
    L.zkhip_fri_chip_air.argtypes = [C.c_int, u32p, C.c_size_t]
    L.zkhip_fri_view_path_words.restype = C.c_size_t
    L.zkhip_fri_view_path_words.argtypes = [C.c_int]
    L.zkhip_fri_view_shard_paths.argtypes = [u8p, C.c_size_t, C.c_int, C.c_uint32, u32p, C.c_size_t, C.POINTER(Params), u32p, u32p, u32p, u32p, u32p, u32p, u32p]
    L.zkhip_fri_view_transcript.argtypes = [u8p, C.c_size_t, C.c_int, C.c_uint32, u32p, C.c_size_t, C.POINTER(Params), u32p, u32p, u32p]
    L.zkhip_fri_view_all.argtypes = [u8p, C.c_size_t, C.c_int, C.c_uint32, u32p, C.c_size_t, C.POINTER(Params), u32p, u32p, u32p, u32p, u32p, u32p, u32p, u32p]
    L.zkhip_fri_indices_program.restype = C.c_size_t
    L.zkhip_fri_indices_program.argtypes = [C.c_int, C.c_int, C.c_int, u32p, C.c_size_t]
    L.zkhip_fri_indices_key.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int, u32p, u32p, C.POINTER(Params), C.POINTER(C.c_void_p), u32p]
    L.zkhip_fri_indices_proof_size.restype = C.c_size_t
    L.zkhip_fri_indices_proof_size.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(Params)]
    L.zkhip_prove_fri_indices.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, u32p, u32p, u32p, u32p, u32p, u32p, u32p, C.c_uint32,
                                          C.POINTER(Params), u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zkhip_verify_fri_indices.argtypes = [u8p, C.c_size_t, C.c_int, C.c_size_t, C.c_int, u32p, u32p, u32p, C.POINTER(Params), C.POINTER(C.c_int)]
    L.zkhip_prove_fri_indices_batch.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(FriJob), C.c_int, C.c_int, C.c_uint32, C.POINTER(Params), C.POINTER(Params),
                                                C.c_int, C.c_int]
    for f in (L.zkhip_fri_layers_chip_air, L.zkhip_p2chip_air_fri_layers, L.zkhip_fri_transcript_chip_air, L.zkhip_p2chip_air_fri_transcript):
        f.restype = C.c_size_t
        f.argtypes = [C.c_int, u32p, C.c_size_t]
    L.zkhip_p2chip_air.restype = C.c_size_t
    L.zkhip_p2chip_air.argtypes = [u32p, C.c_size_t]
    L.zkhip_p2chip_gen_merkle_trace.argtypes = [C.c_void_p, u32p, C.c_uint32, u32p, u32p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, u32p]
    L.zkhip_merkle_paths_proof_size.restype = C.c_size_t
    L.zkhip_merkle_paths_proof_size.argtypes = [C.c_size_t, C.c_int, C.c_uint32, C.POINTER(Params)]
    L.zkhip_prove_merkle_paths.argtypes = [C.c_void_p, u32p, C.c_uint32, u32p, u32p, C.c_size_t, C.c_int, u32p, C.POINTER(Params), u8p, C.c_size_t, szp]
    L.zkhip_verify_merkle_paths.argtypes = [u8p, C.c_size_t, u32p, C.c_size_t, C.POINTER(Params), C.POINTER(C.c_int)]
    L.zkhip_sha256_air_chained.restype = C.c_size_t
    L.zkhip_sha256_air_chained.argtypes = [u32p, C.c_size_t]
    L.zkhip_sha256_gen_trace_chained.argtypes = [C.c_void_p, u32p, u8p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, u32p]
    L.zkhip_sha256_sharded_count.restype = C.c_size_t
    L.zkhip_sha256_sharded_count.argtypes = [C.c_size_t, C.c_int]
    L.zkhip_sha256_shard_proof_size.restype = C.c_size_t
    L.zkhip_sha256_shard_proof_size.argtypes = [C.c_int, C.POINTER(Params)]
    L.zkhip_prove_sha256_sharded.argtypes = [C.POINTER(C.c_int), C.c_int, u8p, C.c_size_t, C.c_int, C.POINTER(Params), C.c_int, u8p, u32p, u8p, C.c_size_t, szp]
    L.zkhip_verify_sha256_sharded.argtypes = [u8p, C.c_size_t, szp, C.c_size_t, u32p, C.c_int, u8p, C.c_uint64, C.POINTER(Params), szp, C.POINTER(C.c_int)]
    L.zkhip_sha256_air.restype = C.c_size_t
    L.zkhip_sha256_air.argtypes = [u32p, C.c_size_t]
    L.zkhip_sha256_digest.restype = None
    L.zkhip_sha256_digest.argtypes = [u8p, C.c_size_t, u8p]
    L.zkhip_sha256_pad.restype = C.c_size_t
    L.zkhip_sha256_pad.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t]
    L.zkhip_sha256_gen_trace.argtypes = [C.c_void_p, u8p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_void_p, C.c_size_t, u32p]
    L.zkhip_sha256_padding_publics.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, u32p]
    L.zkhip_sha256_padding_publics.restype = None
    L.zkhip_sha256_proof_size.restype = C.c_size_t
    L.zkhip_sha256_proof_size.argtypes = [C.c_size_t, C.POINTER(Params)]
    L.zkhip_prove_sha256.argtypes = [C.c_void_p, u8p, C.c_size_t, C.POINTER(Params), u8p, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zkhip_verify_sha256.argtypes = [u8p, C.c_size_t, u8p, C.c_uint64, C.POINTER(Params), C.POINTER(C.c_int)]
    L.zkhip_load_poseidon2_params.argtypes = [C.c_char_p]
    L.zkhip_poseidon2_params_name.restype = C.c_char_p
    L.zkhip_poseidon2_params_name.argtypes = [C.c_int]
    L.zkhip_bincode_size.restype = C.c_size_t
    L.zkhip_bincode_size.argtypes = [C.c_int, C.c_uint32, C.POINTER(Params)]
    L.zkhip_proof_to_bincode.argtypes = [u8p, C.c_size_t, C.c_int, C.c_uint32, C.POINTER(Params), u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zkhip_proof_from_bincode.argtypes = [u8p, C.c_size_t, C.c_int, C.c_uint32, C.POINTER(Params), C.c_size_t, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zkhip_request_digest.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, u32p]
    L.zkhip_shard_device.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int]
    L.zkhip_prove_segment.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, u32p, C.c_size_t,
                                      C.POINTER(Params), u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zkhip_verify_shard.argtypes = [u8p, C.c_size_t, C.c_int, C.c_uint32, u32p, C.c_size_t, C.POINTER(Params), C.POINTER(C.c_int)]
    L.zkhip_last_prove_debug.argtypes = [C.c_void_p, C.POINTER(ProveDebug)]
    i32p = C.POINTER(C.c_int32)
    L.zkhip_chips_proof_size.restype = C.c_size_t
    L.zkhip_chips_proof_size.argtypes = [i32p, u32p, i32p, i32p, C.c_int, C.POINTER(Params), C.c_size_t]
    L.zkhip_gen_trace_logup_cross.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t]
    L.zkhip_prove_chips.argtypes = [C.c_void_p, C.POINTER(Chip), C.c_int, u32p, C.c_size_t, C.POINTER(Params), u8p, C.c_size_t,
                                    C.POINTER(C.c_size_t)]
    L.zkhip_verify_chips.argtypes = [u8p, C.c_size_t, i32p, u32p, i32p, i32p, C.c_int, u32p, C.c_size_t, C.POINTER(Params), C.POINTER(C.c_int)]
    _LIB = L
    return L


def check(rc):
    if rc != 0:
        raise ZkHipError(rc, load().zkhip_last_error().decode("utf-8", "replace"))


def device_count():
    return load().zkhip_device_count()


def to_monty(a):
    a = np.asarray(a, dtype=np.uint64)
    return ((a << np.uint64(32)) % np.uint64(P)).astype(np.uint32)


def from_monty(a):
    a = np.asarray(a, dtype=np.uint64)
    return ((a * np.uint64(943718400)) % np.uint64(P)).astype(np.uint32)
