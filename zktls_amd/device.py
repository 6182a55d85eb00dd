"""Thin object layer over the C ABI: a context bound to one GPU + stream, and device
buffers.  Used by the tests, bench.py and the host-side prover mirror (prover.py).
All numpy arrays crossing this layer are CANONICAL residues unless a name says monty.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import Params, ProveDebug, check, from_monty, to_monty, u32p, u8p


SHA256_WIDTH, SHA256_PUBLIC, SHA256_PADDING_PUBLIC = 640, 91, 75      # include/zkhip.h: ZKHIP_SHA256_*


class DeviceBuffer:
    """A device allocation of uint32 words owned by a Context (or wrapping a torch tensor)."""

    def __init__(self, ctx, nwords, ptr=None, owner=None):
        self.ctx = ctx
        self.nwords = int(nwords)
        self._owned = ptr is None
        self._owner = owner
        if ptr is None:
            p = C.c_void_p()
            check(ctx.lib.zkhip_malloc(ctx.handle, C.c_size_t(self.nwords * 4), C.byref(p)))
            ptr = p.value
            ctx._buffers.add(self)
        self.ptr = ptr

    def offset(self, words):
        return C.c_void_p(self.ptr + 4 * int(words))

    def upload_monty(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint32)
        assert arr.size <= self.nwords
        check(self.ctx.lib.zkhip_memcpy_h2d(self.ctx.handle, C.c_void_p(self.ptr), arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.size * 4)))

    def upload(self, canonical):
        self.upload_monty(to_monty(canonical))

    def download_monty(self, nwords=None, offset=0):
        n = self.nwords - offset if nwords is None else int(nwords)
        out = np.empty(n, dtype=np.uint32)
        check(self.ctx.lib.zkhip_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(C.c_void_p), self.offset(offset), C.c_size_t(n * 4)))
        return out

    def download(self, nwords=None, offset=0):
        return from_monty(self.download_monty(nwords, offset))

    def free(self):
        if self._owned and self.ptr and self.ctx.handle:
            self.ctx.lib.zkhip_free(self.ctx.handle, C.c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceView:
    """words of a DeviceBuffer from an offset on (a column slice of a row-major matrix is a view with the matrix's row pitch)"""

    def __init__(self, buf, words):
        self.base = buf
        self.ptr = buf.ptr + 4 * int(words)


class Context:
    def __init__(self, device=0, stream=None):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.zkhip_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.handle = h
        self.device = device
        self._buffers = weakref.WeakSet()     # allocations made through this context, freed with it

    def close(self):
        if self.handle:
            for b in list(self._buffers):
                b.free()
            self.lib.zkhip_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.lib.zkhip_ctx_sync(self.handle))

    def set_lde_fusion(self, on):
        """fused middle launch of 2^20-row LDEs on (default) / off; returns the previous setting"""
        r = self.lib.zkhip_ctx_set_lde_fusion(self.handle, 1 if on else 0)
        if r < 0:
            check(r)
        return bool(r)

    @property
    def stream(self):
        return self.lib.zkhip_ctx_stream(self.handle)

    def alloc(self, nwords):
        return DeviceBuffer(self, nwords)

    def wrap(self, tensor):
        """wrap a torch CUDA tensor (int32/uint32 storage) without copying"""
        return DeviceBuffer(self, tensor.numel(), ptr=tensor.data_ptr(), owner=tensor)

    def from_numpy(self, canonical):
        a = np.ascontiguousarray(canonical, dtype=np.uint32)
        buf = self.alloc(a.size)
        buf.upload(a)
        return buf

    # ---- synthetic data
    def fill_uniform(self, seed, log_n, width, out=None):
        out = out or self.alloc(width << log_n)
        check(self.lib.zkhip_fill_uniform(self.handle, seed, log_n, width, C.c_void_p(out.ptr), width))
        return out

    def gen_trace_logup_cross(self, seed, shard, partner_shard, log_n, width, partner_width, pairs, out=None):
        out = out or self.alloc(width << log_n)
        check(self.lib.zkhip_gen_trace_logup_cross(self.handle, seed, shard, partner_shard, log_n, width, partner_width, pairs, C.c_void_p(out.ptr), width))
        return out

    def gen_trace(self, seed, shard, log_n, width, out=None):
        out = out or self.alloc(width << log_n)
        check(self.lib.zkhip_gen_trace(self.handle, seed, shard, log_n, width, C.c_void_p(out.ptr), width))
        return out

    def gen_trace_logup(self, seed, shard, log_n, width, pairs, out=None):
        out = out or self.alloc(width << log_n)
        check(self.lib.zkhip_gen_trace_logup(self.handle, seed, shard, log_n, width, pairs, C.c_void_p(out.ptr), width))
        return out

    def perm_trace(self, trace, log_n, width, pairs, gamma, beta, out=None):
        out = out or self.alloc((4 * (pairs + 1)) << log_n)
        g = to_monty(np.asarray(gamma, dtype=np.uint32))
        b = to_monty(np.asarray(beta, dtype=np.uint32))
        check(self.lib.zkhip_perm_trace(self.handle, C.c_void_p(trace.ptr), width, log_n, width, pairs,
                                        g.ctypes.data_as(u32p), b.ctypes.data_as(u32p), C.c_void_p(out.ptr)))
        return out

    # ---- NTT / LDE
    def dft(self, src, log_n, width, inverse=False, bitrev_out=False, out=None):
        out = out or self.alloc(width << log_n)
        check(self.lib.zkhip_dft(self.handle, C.c_void_p(src.ptr), width, C.c_void_p(out.ptr), width, log_n, width,
                                 int(inverse), int(bitrev_out)))
        return out

    def coset_lde(self, src, log_n, width, log_blowup=1, shift=31, out=None, in_ld=None, out_ld=None, out_col=0):
        out_ld = out_ld or width
        in_ld = in_ld or width
        out = out or self.alloc(out_ld << (log_n + log_blowup))
        check(self.lib.zkhip_coset_lde(self.handle, C.c_void_p(src.ptr), in_ld, out.offset(out_col), out_ld,
                                       log_n, width, log_blowup, shift))
        return out

    def ntt_pass(self, src, dst, log_n, width, which):
        """which 0 / 1: stand-alone passes src -> dst; which 2..5: the LDE launches of a proof on this context's own workspaces (dst unused)"""
        check(self.lib.zkhip_ntt_pass(self.handle, C.c_void_p(src.ptr) if src is not None else None,
                                      C.c_void_p(dst.ptr) if dst is not None else None, width, log_n, width, which))

    # ---- Poseidon2 / Merkle
    def poseidon2_permute(self, states):
        check(self.lib.zkhip_poseidon2_permute(self.handle, C.c_void_p(states.ptr), states.nwords // 16))

    def _mat_args(self, mats):
        n = len(mats)
        ptrs = (C.c_void_p * n)(*[m[0].ptr for m in mats])
        lds = (C.c_size_t * n)(*[m[1] for m in mats])
        ws = (C.c_uint32 * n)(*[m[1] for m in mats])
        return ptrs, lds, ws

    def hash_rows(self, mats, height, out=None):
        """mats: list of (DeviceBuffer, width) with contiguous rows"""
        out = out or self.alloc(8 * height)
        ptrs, lds, ws = self._mat_args(mats)
        check(self.lib.zkhip_hash_rows(self.handle, ptrs, lds, ws, len(mats), height, C.c_void_p(out.ptr)))
        return out

    def merkle_commit(self, mats, log_h, out=None):
        out = out or self.alloc(8 * ((2 << log_h) - 1))
        ptrs, lds, ws = self._mat_args(mats)
        check(self.lib.zkhip_merkle_commit(self.handle, ptrs, lds, ws, len(mats), log_h, C.c_void_p(out.ptr)))
        return out

    def merkle_commit_mixed(self, mats, out=None):
        """mats: list of (DeviceBuffer, width, log_height); tree sized for the tallest"""
        log_h = max(m[2] for m in mats)
        out = out or self.alloc(8 * ((2 << log_h) - 1))
        ptrs, lds, ws = self._mat_args(mats)
        lhs = (C.c_int * len(mats))(*[m[2] for m in mats])
        check(self.lib.zkhip_merkle_commit_mixed(self.handle, ptrs, lds, ws, lhs, len(mats), C.c_void_p(out.ptr)))
        return out

    def merkle_commit_p24_colmajor(self, mat, cols, log_rows, out=None):
        out = out or self.alloc(8 * ((2 << log_rows) - 1))
        check(self.lib.zkhip_merkle_commit_p24_colmajor(self.handle, C.c_void_p(mat.ptr), cols, log_rows, C.c_void_p(out.ptr)))
        return out

    def batch_interpolate_colmajor(self, evals, count, log_size, out=None):
        out = out or self.alloc(count << log_size)
        check(self.lib.zkhip_batch_interpolate_colmajor(self.handle, C.c_void_p(evals.ptr), C.c_void_p(out.ptr), count, log_size))
        return out

    def batch_expand_colmajor(self, coeffs, count, log_size, log_blowup=2, shift=31, out=None):
        out = out or self.alloc(count << (log_size + log_blowup))
        check(self.lib.zkhip_batch_expand_colmajor(self.handle, C.c_void_p(coeffs.ptr), C.c_void_p(out.ptr), count, log_size, log_blowup, shift))
        return out

    # ---- RISC Zero Hal operators (column-major vectors; ext_field 0: x^4 = 11, 1: x^4 = -11)
    def eltwise_add(self, a, b, out=None):
        out = out or self.alloc(a.nwords)
        check(self.lib.zkhip_eltwise_add(self.handle, C.c_void_p(out.ptr), C.c_void_p(a.ptr), C.c_void_p(b.ptr), a.nwords))
        return out

    def eltwise_copy(self, a, out=None):
        out = out or self.alloc(a.nwords)
        check(self.lib.zkhip_eltwise_copy(self.handle, C.c_void_p(out.ptr), C.c_void_p(a.ptr), a.nwords))
        return out

    def eltwise_zeroize(self, io):
        check(self.lib.zkhip_eltwise_zeroize(self.handle, C.c_void_p(io.ptr), io.nwords))
        return io

    def eltwise_sum_ext(self, inp, count, out=None):
        out = out or self.alloc(4 * count)
        check(self.lib.zkhip_eltwise_sum_ext(self.handle, C.c_void_p(out.ptr), C.c_void_p(inp.ptr), count, inp.nwords // (4 * count)))
        return out

    def zk_shift(self, io, count, log_size, shift=3):
        check(self.lib.zkhip_zk_shift(self.handle, C.c_void_p(io.ptr), count, log_size, shift))
        return io

    def mix_poly_coeffs(self, out, mix_start, mix, inp, combos, input_size, count, ext_field=0):
        ms = to_monty(np.asarray(mix_start, dtype=np.uint32))
        mx = to_monty(np.asarray(mix, dtype=np.uint32))
        check(self.lib.zkhip_mix_poly_coeffs(self.handle, C.c_void_p(out.ptr), ms.ctypes.data_as(u32p), mx.ctypes.data_as(u32p),
                                             C.c_void_p(inp.ptr), C.c_void_p(combos.ptr), input_size, count, ext_field))
        return out

    def batch_evaluate_any(self, coeffs, log_size, which, xs, ext_field=0, out=None):
        n = which.nwords
        out = out or self.alloc(4 * n)
        check(self.lib.zkhip_batch_evaluate_any(self.handle, C.c_void_p(coeffs.ptr), log_size, C.c_void_p(which.ptr), C.c_void_p(xs.ptr),
                                                C.c_void_p(out.ptr), n, ext_field))
        return out

    def gather_sample(self, src, idx, size, stride, out=None):
        out = out or self.alloc(size)
        check(self.lib.zkhip_gather_sample(self.handle, C.c_void_p(out.ptr), C.c_void_p(src.ptr), idx, size, stride))
        return out

    def scatter(self, into, index, offsets, values):
        check(self.lib.zkhip_scatter(self.handle, C.c_void_p(into.ptr), C.c_void_p(index.ptr), C.c_void_p(offsets.ptr), C.c_void_p(values.ptr),
                                     index.nwords - 1))
        return into

    def prefix_products_ext(self, io, ext_field=0):
        check(self.lib.zkhip_prefix_products_ext(self.handle, C.c_void_p(io.ptr), io.nwords // 4, ext_field))
        return io

    def hash_rows_sha256(self, mat, cols, rows, out=None):
        out = out or self.alloc(8 * rows)
        check(self.lib.zkhip_hash_rows_sha256(self.handle, C.c_void_p(mat.ptr), cols, rows, C.c_void_p(out.ptr)))
        return out

    def hash_fold_sha256(self, children, count, out=None):
        out = out or self.alloc(8 * count)
        check(self.lib.zkhip_hash_fold_sha256(self.handle, C.c_void_p(children.ptr), C.c_void_p(out.ptr), count))
        return out

    def merkle_commit_sha256_colmajor(self, mat, cols, log_rows, out=None):
        out = out or self.alloc(8 * ((2 << log_rows) - 1))
        check(self.lib.zkhip_merkle_commit_sha256_colmajor(self.handle, C.c_void_p(mat.ptr), cols, log_rows, C.c_void_p(out.ptr)))
        return out

    def from_raw(self, words):
        """upload 32-bit words as they are (indices, digests: no Montgomery conversion)"""
        a = np.ascontiguousarray(words, dtype=np.uint32)
        buf = self.alloc(max(a.size, 1))
        buf.upload_monty(a)
        return buf

    # ---- STARK stages
    def quotient_values(self, lde, log_n, width, alpha, out=None):
        out = out or self.alloc(4 << (log_n + 1))
        a = to_monty(np.asarray(alpha, dtype=np.uint32))
        check(self.lib.zkhip_quotient_values(self.handle, C.c_void_p(lde.ptr), width, log_n, width,
                                             a.ctypes.data_as(u32p), C.c_void_p(out.ptr)))
        return out

    def open_at(self, lde, log_n, log_blowup, width, points):
        pts = to_monty(np.ascontiguousarray(points, dtype=np.uint32).reshape(-1, 4))
        out = np.empty((pts.shape[0], width, 4), dtype=np.uint32)
        check(self.lib.zkhip_open_at(self.handle, C.c_void_p(lde.ptr), width, log_n, log_blowup, width,
                                     pts.ctypes.data_as(u32p), pts.shape[0], out.ctypes.data_as(u32p)))
        return from_monty(out)

    def fri_fold(self, src, log_h, beta, out=None):
        out = out or self.alloc(4 << (log_h - 1))
        b = to_monty(np.asarray(beta, dtype=np.uint32))
        check(self.lib.zkhip_fri_fold(self.handle, C.c_void_p(src.ptr), log_h, b.ctypes.data_as(u32p), C.c_void_p(out.ptr)))
        return out

    def fri_fold_k(self, src, log_h, log_arity, beta, out=None):
        out = out or self.alloc(4 << (log_h - log_arity))
        b = to_monty(np.asarray(beta, dtype=np.uint32))
        check(self.lib.zkhip_fri_fold_k(self.handle, C.c_void_p(src.ptr), log_h, log_arity, b.ctypes.data_as(u32p), C.c_void_p(out.ptr)))
        return out

    def commit(self, trace, log_n, width, log_blowup=1, hash_width=16):
        """coset LDE + Merkle tree + root in one call; returns (lde, tree, root[8] canonical)"""
        h = log_n + log_blowup
        lde, tree = self.alloc(width << h), self.alloc(8 * ((2 << h) - 1))
        root = np.zeros(8, dtype=np.uint32)
        check(self.lib.zkhip_commit(self.handle, C.c_void_p(trace.ptr), width, log_n, width, log_blowup, hash_width,
                                    C.c_void_p(lde.ptr), C.c_void_p(tree.ptr), root.ctypes.data_as(u32p)))
        return lde, tree, root

    # ---- whole shard
    def prove_shard(self, trace, log_n, width, public_values=(), params=None):
        params = params or Params(1, 100, 16, 0)
        pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
        size = self.lib.zkhip_proof_size(log_n, width, C.byref(params), pv.size)
        buf = np.empty(size, dtype=np.uint8)
        got = C.c_size_t(0)
        check(self.lib.zkhip_prove_shard(self.handle, C.c_void_p(trace.ptr), width, log_n, width,
                                         pv.ctypes.data_as(u32p), pv.size, C.byref(params),
                                         buf.ctypes.data_as(u8p), size, C.byref(got)))
        return buf[: got.value]

    def prove_shard_air(self, program, trace, log_n, width, public_values=(), params=None):
        """prove against a constraint program (the AIR as data: include/zkhip.h); program: numpy u32 words"""
        params = params or Params(1, 100, 16, 0)
        prog = np.ascontiguousarray(program, dtype=np.uint32)
        pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
        size = self.lib.zkhip_proof_size_air(prog.ctypes.data_as(u32p), prog.size, log_n, width, C.byref(params), pv.size)
        if size == 0:
            check(self.lib.zkhip_air_validate(prog.ctypes.data_as(u32p), prog.size, width, pv.size))
            raise _lib.ZkHipError(-1, "prove_shard_air: bad shape (degree 4 / 5 programs need log_blowup >= 2; no logup_pairs)")
        buf = np.empty(max(size, 1), dtype=np.uint8)
        got = C.c_size_t(0)
        check(self.lib.zkhip_prove_shard_air(self.handle, prog.ctypes.data_as(u32p), prog.size, C.c_void_p(trace.ptr), width, log_n, width,
                                             pv.ctypes.data_as(u32p), pv.size, C.byref(params), buf.ctypes.data_as(u8p), size, C.byref(got)))
        return buf[: got.value]

    # ---- the SHA-256 compression chip (csrc/sha256_chip.hip)
    def sha256_gen_trace(self, blocks, n_blocks=None, out=None, message_len=None):
        """blocks: the padded message, a multiple of 64 long (sha256_pad / sha256_air.pad remember the message's length; otherwise pass
        message_len) -> (device trace [64 n_blocks][640], public values [91]: 16 digest limbs + the 75 padding values)"""
        if message_len is None:
            message_len = getattr(blocks, "message_len", None)
        if message_len is None:
            raise TypeError("sha256_gen_trace: the message length is part of the statement (pass message_len)")
        b = np.frombuffer(bytes(blocks), dtype=np.uint8)
        active = b.size // 64
        n_blocks = n_blocks or 1 << max(active - 1, 0).bit_length()
        out = out or self.alloc(64 * n_blocks * SHA256_WIDTH)
        limbs = np.zeros(SHA256_PUBLIC, dtype=np.uint32)
        check(self.lib.zkhip_sha256_gen_trace(self.handle, b.ctypes.data_as(u8p), active, n_blocks, message_len, C.c_void_p(out.ptr), SHA256_WIDTH, limbs.ctypes.data_as(u32p)))
        return out, limbs

    def prove_sha256(self, message, params=None):
        """-> (digest bytes, proof bytes): "I know a message with this SHA-256 digest" """
        params = params or Params(1, 100, 16)
        m = np.frombuffer(bytes(message), dtype=np.uint8) if len(message) else np.zeros(1, dtype=np.uint8)
        size = self.lib.zkhip_sha256_proof_size(len(message), C.byref(params))
        if size == 0:
            raise _lib.ZkHipError(-1, "prove_sha256: bad shape or message too long")
        buf = np.empty(size, dtype=np.uint8)
        digest = np.zeros(32, dtype=np.uint8)
        got = C.c_size_t(0)
        check(self.lib.zkhip_prove_sha256(self.handle, m.ctypes.data_as(u8p), len(message), C.byref(params), digest.ctypes.data_as(u8p),
                                          buf.ctypes.data_as(u8p), size, C.byref(got)))
        return digest.tobytes(), buf[: got.value]

    def p2chip_gen_merkle_trace(self, leaves, siblings, indices, log_n=None, hashed_rows=False):
        """the Poseidon2 chip's trace for a set of Merkle paths (numpy arrays of canonical words: leaves [n][8] -- or, with hashed_rows, the
        opened rows [n][8 k], hashed in-circuit --, siblings [n][depth][8], indices [n]) -> (device buffer [2^log_n][360], roots [n][8], log_n)"""
        lv = np.ascontiguousarray(leaves, dtype=np.uint32)
        sb = np.ascontiguousarray(siblings, dtype=np.uint32)
        ix = np.ascontiguousarray(indices, dtype=np.uint32)
        n, depth = lv.shape[0], sb.shape[1]
        rw = lv.shape[1] if hashed_rows else 0
        if log_n is None:
            log_n = max(5, (n * (depth + rw // 8) - 1).bit_length())
        out = self.alloc(360 << log_n)
        roots = np.zeros((n, 8), dtype=np.uint32)
        check(self.lib.zkhip_p2chip_gen_merkle_trace(self.handle, lv.ctypes.data_as(u32p), rw, sb.ctypes.data_as(u32p), ix.ctypes.data_as(u32p), n, depth, log_n,
                                                     C.c_void_p(out.ptr), 360, roots.ctypes.data_as(u32p)))
        return out, roots, log_n

    def prove_merkle_paths(self, leaves, siblings, indices, root, params=None, hashed_rows=False):
        """-> proof bytes of "I know len(leaves) Merkle paths (with hashed_rows: openings of whole rows) that end in root" (the Poseidon2 chip)"""
        params = params or Params(1, 100, 16)
        lv = np.ascontiguousarray(leaves, dtype=np.uint32)
        sb = np.ascontiguousarray(siblings, dtype=np.uint32)
        ix = np.ascontiguousarray(indices, dtype=np.uint32)
        rt = np.ascontiguousarray(root, dtype=np.uint32)
        n, depth = lv.shape[0], sb.shape[1]
        rw = lv.shape[1] if hashed_rows else 0
        size = self.lib.zkhip_merkle_paths_proof_size(n, depth, rw, C.byref(params))
        buf = np.empty(max(size, 1), dtype=np.uint8)
        got = C.c_size_t(0)
        check(self.lib.zkhip_prove_merkle_paths(self.handle, lv.ctypes.data_as(u32p), rw, sb.ctypes.data_as(u32p), ix.ctypes.data_as(u32p), n, depth,
                                                rt.ctypes.data_as(u32p), C.byref(params), buf.ctypes.data_as(u8p), size, C.byref(got)))
        return buf[: got.value]

    def sha256_setup(self, params=None):
        """zkhip_sha256_setup: the SHA-256 machine's key (the range table's preprocessed values, committed once) -> MachineKey; .root is the vk"""
        params = params or Params(1, 100, 16)
        handle = C.c_void_p()
        root = np.zeros(8, dtype=np.uint32)
        check(self.lib.zkhip_sha256_setup(self.handle, C.byref(params), C.byref(handle), root.ctypes.data_as(u32p)))
        return MachineKey(self, handle, root, [4])

    def machine_verifier_setup(self, inner, params=None, n_proofs=1):
        """zkhip_machine_verifier_setup: the key of the machine that verifies n_proofs proofs of the inner machine (InnerMachine) in-circuit"""
        params = params or Params()
        handle, root = C.c_void_p(), np.zeros(8, dtype=np.uint32)
        check(self.lib.zkhip_machine_verifier_setup(self.handle, C.byref(inner.desc), n_proofs, C.byref(params), C.byref(handle), root.ctypes.data_as(u32p)))
        return MachineKey(self, handle, root, None)

    def prove_machine_verifier(self, key, inner, proofs, public_values, params=None):
        """zkhip_prove_machine_verifier: proofs = a list of version-11 proofs of the inner machine, public_values one list per proof -> ONE proof"""
        params = params or Params()
        sps = [np.ascontiguousarray(sp, dtype=np.uint8) for sp in proofs]
        n = len(sps)
        pv = np.ascontiguousarray(np.array([list(v) for v in public_values], dtype=np.uint32).reshape(n, -1)) if inner.n_public else np.zeros((n, 1), dtype=np.uint32)
        size = self.lib.zkhip_machine_verifier_proof_size(C.byref(inner.desc), n, C.byref(params))
        if size == 0:
            check(-1)
        buf = np.empty(size, dtype=np.uint8)
        got = C.c_size_t(0)
        ptrs = (u8p * n)(*[sp.ctypes.data_as(u8p) for sp in sps])
        lens = (C.c_size_t * n)(*[sp.size for sp in sps])
        check(self.lib.zkhip_prove_machine_verifier(self.handle, key.handle, C.byref(inner.desc), ptrs, lens, n, pv.ctypes.data_as(u32p), inner.n_public, C.byref(params),
                                                    buf.ctypes.data_as(u8p), size, C.byref(got)))
        return buf[: got.value]

    def sha256_compress_setup(self, message_len, log_blocks_per_shard, inner=None, outer=None):
        """zkhip_sha256_compress_setup -> MachineKey (key.root = vk) for compressed chains of messages with this many shards"""
        inner, outer = inner or Params(1, 100, 16), outer or Params(1, 100, 16)
        handle = C.c_void_p()
        vk = np.zeros(8, dtype=np.uint32)
        check(self.lib.zkhip_sha256_compress_setup(self.handle, message_len, log_blocks_per_shard, C.byref(inner), C.byref(outer), C.byref(handle), vk.ctypes.data_as(u32p)))
        return MachineKey(self, handle, vk, None)

    def prove_sha256_compressed(self, key, message, log_blocks_per_shard, inner=None, outer=None, devices=None, in_flight=2):
        """zkhip_prove_sha256_compressed: the message as a chain of shards (dealt over `devices`), the shards verified in-circuit on this
        context's device -> (digest bytes, chain [(n_shards + 1) x 8], ONE proof)"""
        inner, outer = inner or Params(1, 100, 16), outer or Params(1, 100, 16)
        m = np.frombuffer(bytes(message), dtype=np.uint8) if len(message) else np.zeros(1, dtype=np.uint8)
        n = self.lib.zkhip_sha256_sharded_count(len(message), log_blocks_per_shard)
        size = self.lib.zkhip_sha256_compressed_proof_size(len(message), log_blocks_per_shard, C.byref(inner), C.byref(outer))
        if n == 0 or size == 0:
            raise _lib.ZkHipError(-1, "prove_sha256_compressed: bad shape")
        chain = np.zeros((n + 1, 8), dtype=np.uint32)
        digest = np.zeros(32, dtype=np.uint8)
        buf = np.empty(size, dtype=np.uint8)
        got = C.c_size_t(0)
        devs = (C.c_int * len(devices))(*devices) if devices else None
        check(self.lib.zkhip_prove_sha256_compressed(self.handle, key.handle, devs, len(devices) if devices else 0, m.ctypes.data_as(u8p), len(message), log_blocks_per_shard,
                                                     C.byref(inner), C.byref(outer), in_flight, digest.ctypes.data_as(u8p), chain.ctypes.data_as(u32p),
                                                     buf.ctypes.data_as(u8p), size, C.byref(got)))
        return digest.tobytes(), chain, buf[: got.value]

    def prove_sha256_machine(self, key, message, params=None):
        """-> (digest bytes, proof bytes): the SHA-256 chip + its range table as a keyed machine (proof version 11)"""
        params = params or Params(1, 100, 16)
        m = np.frombuffer(bytes(message), dtype=np.uint8) if len(message) else np.zeros(1, dtype=np.uint8)
        size = self.lib.zkhip_sha256_machine_proof_size(len(message), C.byref(params))
        if size == 0:
            raise _lib.ZkHipError(-1, "prove_sha256_machine: bad shape or message too long")
        buf = np.empty(size, dtype=np.uint8)
        digest = np.zeros(32, dtype=np.uint8)
        got = C.c_size_t(0)
        check(self.lib.zkhip_prove_sha256_machine(self.handle, key.handle, m.ctypes.data_as(u8p), len(message), C.byref(params), digest.ctypes.data_as(u8p),
                                                  buf.ctypes.data_as(u8p), size, C.byref(got)))
        return digest.tobytes(), buf[: got.value]

    def quotient_values_air(self, program, lde, log_n, width, public_values, alpha, out=None, log_quotient_degree=1):
        out = out or self.alloc(4 << (log_n + log_quotient_degree))
        prog = np.ascontiguousarray(program, dtype=np.uint32)
        pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
        a = to_monty(np.asarray(alpha, dtype=np.uint32))
        check(self.lib.zkhip_quotient_values_air(self.handle, prog.ctypes.data_as(u32p), prog.size, C.c_void_p(lde.ptr), width, log_n, width,
                                                 pv.ctypes.data_as(u32p), pv.size, a.ctypes.data_as(u32p), C.c_void_p(out.ptr)))
        return out

    def prove_shard_host(self, host_trace, public_values=(), params=None, host_ptr=None, log_n=None, width=None):
        """host_trace: numpy [2^log_n][width] canonical words in host memory (or a raw host pointer + shape)"""
        params = params or Params(1, 100, 16, 0)
        if host_ptr is None:
            a = np.ascontiguousarray(host_trace, dtype=np.uint32)
            log_n, width = a.shape[0].bit_length() - 1, a.shape[1]
            host_ptr = a.ctypes.data
        pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
        size = self.lib.zkhip_proof_size(log_n, width, C.byref(params), pv.size)
        buf = np.empty(size, dtype=np.uint8)
        got = C.c_size_t(0)
        check(self.lib.zkhip_prove_shard_host(self.handle, C.c_void_p(host_ptr), log_n, width, pv.ctypes.data_as(u32p), pv.size,
                                              C.byref(params), buf.ctypes.data_as(u8p), size, C.byref(got)))
        return buf[: got.value]

    def prove_segment(self, cols, log_n, width, public_values=(), params=None):
        """cols: device buffer, column-major [width][2^log_n] (RISC Zero's Hal layout)"""
        params = params or _lib.segment_params()
        pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
        size = self.lib.zkhip_proof_size(log_n, width, C.byref(params), pv.size)
        buf = np.empty(size, dtype=np.uint8)
        got = C.c_size_t(0)
        check(self.lib.zkhip_prove_segment(self.handle, C.c_void_p(cols.ptr), log_n, width,
                                           pv.ctypes.data_as(u32p), pv.size, C.byref(params),
                                           buf.ctypes.data_as(u8p), size, C.byref(got)))
        return buf[: got.value]

    def prove_chips(self, chips, public_values=(), params=None):
        """chips: [(device buffer, log_n, width[, logup_pairs[, partner]]), ...] tallest first -- one shard of several AIR
        tables (SP1's shard shape); partner = index of the chip this one exchanges lookups with, -1 for none"""
        params = params or Params(1, 100, 16, 0)
        chips = [tuple(c) + ((0,) if len(c) < 4 else ()) for c in chips]
        chips = [tuple(c) + ((-1,) if len(c) < 5 else ()) for c in chips]
        n = len(chips)
        arr = (_lib.Chip * n)(*[_lib.Chip(b.ptr, w, ln, w, pr, pa) for b, ln, w, pr, pa in chips])
        log_ns = (C.c_int32 * n)(*[c[1] for c in chips])
        widths = (C.c_uint32 * n)(*[c[2] for c in chips])
        pairs = (C.c_int32 * n)(*[c[3] for c in chips])
        partners = (C.c_int32 * n)(*[c[4] for c in chips])
        pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
        size = self.lib.zkhip_chips_proof_size(log_ns, widths, pairs, partners, n, C.byref(params), pv.size)
        if size == 0:
            check(-1)
        buf = np.empty(size, dtype=np.uint8)
        got = C.c_size_t(0)
        check(self.lib.zkhip_prove_chips(self.handle, arr, n, pv.ctypes.data_as(u32p), pv.size, C.byref(params),
                                         buf.ctypes.data_as(u8p), size, C.byref(got)))
        return buf[: got.value]

    def prove_chips_air(self, chips, programs, public_values=(), params=None):
        """chips: [(device buffer, log_n, width), ...] tallest first; programs[c]: a constraint program (numpy u32 words; degree 4 / 5 needs log_blowup >= 2)
        or None for the built-in synthetic AIR -- several different AIR tables in one proof (version 9)"""
        params = params or Params(1, 100, 16, 0)
        n = len(chips)
        arr = (_lib.Chip * n)(*[_lib.Chip(b.ptr, w, ln, w, 0, -1) for b, ln, w in chips])
        log_ns = (C.c_int32 * n)(*[c[1] for c in chips])
        widths = (C.c_uint32 * n)(*[c[2] for c in chips])
        keep, pp, pw = _program_table(programs)
        pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
        size = self.lib.zkhip_chips_proof_size_air(log_ns, widths, pp, pw, n, C.byref(params), pv.size)
        if size == 0:
            check(-1)
        buf = np.empty(size, dtype=np.uint8)
        got = C.c_size_t(0)
        check(self.lib.zkhip_prove_chips_air(self.handle, arr, pp, pw, n, pv.ctypes.data_as(u32p), pv.size, C.byref(params),
                                             buf.ctypes.data_as(u8p), size, C.byref(got)))
        return buf[: got.value]

    def range_table(self, trace, ld, rows, columns, log_table, width=4, value_col=0, mult_col=1):
        """a 2^log_table x width range table on the device: value column = row index, multiplicity column = how often the value
        appears in `columns` of `trace` (the other columns zero)"""
        out = self.from_raw(np.zeros(width << log_table, dtype=np.uint32))
        cols = (C.c_uint32 * len(columns))(*[int(c) for c in columns])
        check(self.lib.zkhip_range_table(self.handle, C.c_void_p(trace.ptr), ld, rows, cols, len(columns), log_table, C.c_void_p(out.ptr), width, value_col, mult_col))
        return out

    def prove_machine(self, chips, programs, tables, public_values=(), params=None):
        """chips: [(device buffer, log_n, width), ...] tallest first; programs[c] / tables[c]: the chip's constraint program and its
        interaction table (numpy u32 words) or None -- a machine whose tables look each other up (lookups as data, version 10)"""
        params = params or Params(1, 100, 16, 0)
        n = len(chips)
        arr = (_lib.Chip * n)(*[_lib.Chip(b.ptr, w, ln, w, 0, -1) for b, ln, w in chips])
        log_ns = (C.c_int32 * n)(*[c[1] for c in chips])
        widths = (C.c_uint32 * n)(*[c[2] for c in chips])
        kp, pp, pw = _program_table(programs)
        kt, tp, tw = _program_table(tables)
        pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
        size = self.lib.zkhip_machine_proof_size(log_ns, widths, pp, pw, tp, tw, n, C.byref(params), pv.size)
        if size == 0:
            check(-1)
        buf = np.empty(size, dtype=np.uint8)
        got = C.c_size_t(0)
        check(self.lib.zkhip_prove_machine(self.handle, arr, pp, pw, tp, tw, n, pv.ctypes.data_as(u32p), pv.size, C.byref(params),
                                           buf.ctypes.data_as(u8p), size, C.byref(got)))
        return buf[: got.value]

    def machine_setup(self, pre_chips, params=None):
        """zkhip_machine_setup: pre_chips = [(device buffer or None, log_n, preprocessed width), ...] tallest first -> MachineKey
        (.root: the 8 canonical words a verifier needs; .pre_widths)"""
        params = params or Params(1, 100, 16, 0)
        n = len(pre_chips)
        arr = (_lib.Chip * n)(*[_lib.Chip(c[0].ptr if c[0] is not None else None, c[3] if len(c) > 3 else c[2], c[1], c[2], 0, -1) for c in pre_chips])      # (buffer, log_n, width[, row pitch])
        handle = C.c_void_p()
        root = np.zeros(8, dtype=np.uint32)
        check(self.lib.zkhip_machine_setup(self.handle, arr, n, C.byref(params), C.byref(handle), root.ctypes.data_as(u32p)))
        return MachineKey(self, handle, root, [int(c[2]) for c in pre_chips])

    # ---- the FRI-fold chip (a first recursion step): trace, key, proof
    def fri_indices_key(self, view, inner_pow_bits, params=None):
        """zkhip_fri_indices_key: (query number, reduced opening), the layer roots, the SAMPLES chip's fixed columns -- no index"""
        params = params or Params(1, 100, 16)
        R, Q, betas, idx, vals, sibs, roots, paths = _fri_layers_arrays(view)
        handle, root = C.c_void_p(), np.zeros(8, dtype=np.uint32)
        check(self.lib.zkhip_fri_indices_key(self.handle, R, Q, inner_pow_bits, vals.ctypes.data_as(u32p), roots.ctypes.data_as(u32p),
                                             C.byref(params), C.byref(handle), root.ctypes.data_as(u32p)))
        return MachineKey(self, handle, root, [0, 0, 8, 12, 20])

    def prove_fri_indices(self, key, view, capacity, witness, inner_pow_bits, params=None):
        """zkhip_prove_fri_indices: the transcript machine with the query phase (proof of work, query indices) in-circuit"""
        params = params or Params(1, 100, 16)
        R, Q, betas, idx, vals, sibs, roots, paths = _fri_layers_arrays(view)
        cap8 = np.ascontiguousarray(np.array(capacity, dtype=np.uint32))
        size = self.lib.zkhip_fri_indices_proof_size(R, Q, inner_pow_bits, C.byref(params))
        buf = np.empty(size, dtype=np.uint8)
        got = C.c_size_t(0)
        check(self.lib.zkhip_prove_fri_indices(self.handle, key.handle, R, Q, inner_pow_bits, betas.ctypes.data_as(u32p), idx.ctypes.data_as(u32p),
                                               vals.ctypes.data_as(u32p), sibs.ctypes.data_as(u32p), roots.ctypes.data_as(u32p), paths.ctypes.data_as(u32p),
                                               cap8.ctypes.data_as(u32p), int(witness), C.byref(params), buf.ctypes.data_as(u8p), size, C.byref(got)))
        return buf[: got.value]

    def shard_verifier_setup(self, log_n, width, n_queries, inner_pow_bits, n_public, params=None, n_proofs=1, program=None):
        """zkhip_shard_verifier_setup: the key of the shard-verifier machine for n_proofs inner proofs of this SHAPE (no inner proof involved);
        program: the inner proofs are version-7 proofs of that constraint program (zkhip_shard_verifier_setup_air)"""
        params = params or Params(1, 100, 16)
        handle, root = C.c_void_p(), np.zeros(8, dtype=np.uint32)
        if program is None:
            check(self.lib.zkhip_shard_verifier_setup(self.handle, log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, C.byref(params), C.byref(handle), root.ctypes.data_as(u32p)))
        else:
            pg = np.ascontiguousarray(program, dtype=np.uint32)
            check(self.lib.zkhip_shard_verifier_setup_air(self.handle, pg.ctypes.data_as(u32p), pg.size, log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, C.byref(params),
                                                          C.byref(handle), root.ctypes.data_as(u32p)))
        return MachineKey(self, handle, root, None)

    def prove_shard_verifier(self, key, shard_proofs, log_n, width, public_values, inner=None, outer=None, program=None):
        """zkhip_prove_shard_verifier: the WHOLE verification of a shard proof (transcript, AIR identity, openings, FRI) proven in-circuit.
        shard_proofs: one proof (bytes / uint8 array) with its public values, or a LIST of proofs with a list of public-value lists (the join)"""
        inner, outer = inner or Params(1, 100, 16), outer or Params(1, 100, 16)
        if not isinstance(shard_proofs, (list, tuple)):
            shard_proofs, public_values = [shard_proofs], [public_values]
        sps = [np.ascontiguousarray(sp, dtype=np.uint8) for sp in shard_proofs]
        n = len(sps)
        pv = np.ascontiguousarray(np.array([list(v) for v in public_values], dtype=np.uint32).reshape(n, -1))
        n_public = pv.shape[1]
        pg = None if program is None else np.ascontiguousarray(program, dtype=np.uint32)
        if pg is None:
            size = self.lib.zkhip_shard_verifier_proof_size(log_n, width, inner.num_queries, inner.pow_bits, n_public, n, C.byref(outer))
        else:
            size = self.lib.zkhip_shard_verifier_proof_size_air(pg.ctypes.data_as(u32p), pg.size, log_n, width, inner.num_queries, inner.pow_bits, n_public, n, C.byref(outer))
        if size == 0:
            check(-1)
        buf = np.empty(size, dtype=np.uint8)
        got = C.c_size_t(0)
        ptrs = (u8p * n)(*[sp.ctypes.data_as(u8p) for sp in sps])
        lens = (C.c_size_t * n)(*[sp.size for sp in sps])
        if pg is None:
            check(self.lib.zkhip_prove_shard_verifier(self.handle, key.handle, ptrs, lens, n, log_n, width, pv.ctypes.data_as(u32p), n_public, C.byref(inner),
                                                      C.byref(outer), buf.ctypes.data_as(u8p), size, C.byref(got)))
        else:
            check(self.lib.zkhip_prove_shard_verifier_air(self.handle, key.handle, pg.ctypes.data_as(u32p), pg.size, ptrs, lens, n, log_n, width, pv.ctypes.data_as(u32p), n_public,
                                                          C.byref(inner), C.byref(outer), buf.ctypes.data_as(u8p), size, C.byref(got)))
        return buf[: got.value]

    def prove_machine_keyed(self, key, chips, programs, tables, public_values=(), params=None, key_entries=None):
        """a machine with preprocessed columns (proof version 11): `key` from machine_setup; chips as in prove_machine (main columns);
        programs / tables address the combined row [preprocessed | main].  key_entries: per chip the key entry it uses (-1: none) when the
        key holds tables only (zkhip_prove_machine_keyed_at); its preprocessed widths then come from the entries"""
        params = params or Params(1, 100, 16, 0)
        n = len(chips)
        arr = (_lib.Chip * n)(*[_lib.Chip(c[0].ptr, c[3] if len(c) > 3 else c[2], c[1], c[2], 0, -1) for c in chips])      # (buffer, log_n, width[, row pitch])
        log_ns = (C.c_int32 * n)(*[c[1] for c in chips])
        widths = (C.c_uint32 * n)(*[c[2] for c in chips])
        if key_entries is not None:
            per_chip = [key.pre_widths[e] if 0 <= e < len(key.pre_widths) else 0 for e in key_entries]
        else:
            per_chip = (list(key.pre_widths) + [0] * n)[:n]                  # (a key of another shape is the library's to refuse)
        pws = (C.c_uint32 * n)(*per_chip)
        kp, pp, pw = _program_table(programs)
        kt, tp, tw = _program_table(tables)
        pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
        size = self.lib.zkhip_machine_proof_size_keyed(log_ns, widths, pws, pp, pw, tp, tw, n, C.byref(params), pv.size)
        buf = np.empty(max(size, 1), dtype=np.uint8)
        got = C.c_size_t(0)
        if key_entries is not None:
            ke = (C.c_int32 * n)(*[int(e) for e in key_entries])
            check(self.lib.zkhip_prove_machine_keyed_at(self.handle, key.handle, ke, arr, pp, pw, tp, tw, n, pv.ctypes.data_as(u32p), pv.size, C.byref(params),
                                                        buf.ctypes.data_as(u8p), size, C.byref(got)))
        else:
            check(self.lib.zkhip_prove_machine_keyed(self.handle, key.handle, arr, pp, pw, tp, tw, n, pv.ctypes.data_as(u32p), pv.size, C.byref(params),
                                                     buf.ctypes.data_as(u8p), size, C.byref(got)))
        return buf[: got.value]

    def prove_debug(self):
        d = ProveDebug()
        check(self.lib.zkhip_last_prove_debug(self.handle, C.byref(d)))
        return {k: (np.array(getattr(d, k), dtype=np.uint32) if k != "pow_witness" else int(d.pow_witness))
                for k, _ in ProveDebug._fields_}


def prove_shards(traces, log_n, width, public_values_list, params=None, device=0, in_flight=4, host=False):
    """zkhip_prove_shards: `traces` are DeviceBuffers (Montgomery, dense) or, with host=True, numpy arrays of canonical words.
    Returns the list of proofs (numpy uint8 arrays)."""
    lib = _lib.load()
    params = params or Params(1, 100, 16, 0)
    n = len(traces)
    jobs = (_lib.ShardJob * n)()
    pvs, bufs, arrs = [], [], []
    for i, (t, pv) in enumerate(zip(traces, public_values_list)):
        pva = np.ascontiguousarray(np.array(pv, dtype=np.uint32))
        size = lib.zkhip_proof_size(log_n, width, C.byref(params), pva.size)
        buf = np.empty(max(size, 1), dtype=np.uint8)
        if host:
            arr = np.ascontiguousarray(t, dtype=np.uint32)
            arrs.append(arr)
            ptr = arr.ctypes.data
        else:
            ptr = t.ptr
        pvs.append(pva); bufs.append(buf)
        jobs[i].trace = ptr; jobs[i].ld = width; jobs[i].log_n = log_n; jobs[i].width = width
        jobs[i].public_values = pva.ctypes.data_as(u32p); jobs[i].n_public = pva.size
        jobs[i].proof = buf.ctypes.data_as(u8p); jobs[i].proof_cap = size
    check(lib.zkhip_prove_shards(int(device), jobs, n, C.byref(params), int(in_flight), 1 if host else 0))
    return [bufs[i][: jobs[i].proof_len] for i in range(n)]


def prove_shards_multi(traces, log_n, width, public_values_list, params=None, devices=None, in_flight=4, host=False):
    """zkhip_prove_shards_multi: one call, one process, several GPUs; shard s runs on devices[s % len(devices)] (all visible
    devices when `devices` is None).  Device traces must live on that device (see shard_device)."""
    lib = _lib.load()
    params = params or Params(1, 100, 16, 0)
    n = len(traces)
    jobs = (_lib.ShardJob * n)()
    keep = []
    for i, (t, pv) in enumerate(zip(traces, public_values_list)):
        pva = np.ascontiguousarray(np.array(pv, dtype=np.uint32))
        size = lib.zkhip_proof_size(log_n, width, C.byref(params), pva.size)
        buf = np.empty(max(size, 1), dtype=np.uint8)
        arr = np.ascontiguousarray(t, dtype=np.uint32) if host else None
        keep.append((pva, buf, arr))
        jobs[i].trace = arr.ctypes.data if host else t.ptr
        jobs[i].ld = width; jobs[i].log_n = log_n; jobs[i].width = width
        jobs[i].public_values = pva.ctypes.data_as(u32p); jobs[i].n_public = pva.size
        jobs[i].proof = buf.ctypes.data_as(u8p); jobs[i].proof_cap = size
    if devices is None:
        dv, nd = None, 0
    else:
        dv, nd = (C.c_int * len(devices))(*[int(d) for d in devices]), len(devices)
    check(lib.zkhip_prove_shards_multi(dv, nd, jobs, n, C.byref(params), int(in_flight), 1 if host else 0))
    return [keep[i][1][: jobs[i].proof_len] for i in range(n)]


def prove_shards_air_multi(program, traces, log_n, width, public_values_list, params=None, devices=None, in_flight=4):
    """zkhip_prove_shards_air_multi: a batch of device traces of ONE constraint program, dealt over `devices` like prove_shards_multi"""
    lib = _lib.load()
    params = params or Params(1, 100, 16, 0)
    prog = np.ascontiguousarray(program, dtype=np.uint32)
    n = len(traces)
    jobs = (_lib.ShardJob * n)()
    keep = []
    for i, (t, pv) in enumerate(zip(traces, public_values_list)):
        pva = np.ascontiguousarray(np.array(pv, dtype=np.uint32))
        size = lib.zkhip_proof_size_air(prog.ctypes.data_as(u32p), prog.size, log_n, width, C.byref(params), pva.size)
        buf = np.empty(max(size, 1), dtype=np.uint8)
        keep.append((pva, buf))
        jobs[i].trace = t.ptr
        jobs[i].ld = width; jobs[i].log_n = log_n; jobs[i].width = width
        jobs[i].public_values = pva.ctypes.data_as(u32p); jobs[i].n_public = pva.size
        jobs[i].proof = buf.ctypes.data_as(u8p); jobs[i].proof_cap = size
    if devices is None:
        dv, nd = None, 0
    else:
        dv, nd = (C.c_int * len(devices))(*[int(d) for d in devices]), len(devices)
    check(lib.zkhip_prove_shards_air_multi(dv, nd, jobs, n, prog.ctypes.data_as(u32p), prog.size, C.byref(params), int(in_flight)))
    return [keep[i][1][: jobs[i].proof_len] for i in range(n)]


def shard_device(shard_index, devices=None, n_devices=None):
    """the device ordinal zkhip_prove_shards_multi proves shard `shard_index` on"""
    lib = _lib.load()
    if devices is None:
        return lib.zkhip_shard_device(int(shard_index), None, int(n_devices))
    dv = (C.c_int * len(devices))(*[int(d) for d in devices])
    return lib.zkhip_shard_device(int(shard_index), dv, len(devices))


def verify_shard(proof, log_n, width, public_values=(), params=None):
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
    reason = C.c_int(0)
    rc = lib.zkhip_verify_shard(pr.ctypes.data_as(u8p), pr.size, log_n, width, pv.ctypes.data_as(u32p), pv.size,
                                C.byref(params), C.byref(reason))
    return rc, reason.value


def verify_shard_air(program, proof, log_n, width, public_values=(), params=None):
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    prog = np.ascontiguousarray(program, dtype=np.uint32)
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
    reason = C.c_int(0)
    rc = lib.zkhip_verify_shard_air(prog.ctypes.data_as(u32p), prog.size, pr.ctypes.data_as(u8p), pr.size, log_n, width,
                                    pv.ctypes.data_as(u32p), pv.size, C.byref(params), C.byref(reason))
    return rc, reason.value


def sha256_air_chained():
    """the chip with its initial chaining value public too (32 public values): the program of one shard of a longer message"""
    lib = _lib.load()
    n = lib.zkhip_sha256_air_chained(None, 0)
    out = np.empty(n, dtype=np.uint32)
    assert lib.zkhip_sha256_air_chained(out.ctypes.data_as(u32p), n) == n
    return out


class ShardedSha256:
    """the result of prove_sha256_sharded: digest, chaining values [n + 1][8], the shard proofs"""

    def __init__(self, digest, chain, proofs, stride, lens, log_blocks, message_len):
        self.digest, self.chain, self.buf, self.stride, self.lens, self.log_blocks, self.message_len = digest, chain, proofs, stride, lens, log_blocks, message_len

    @property
    def proofs(self):
        return [self.buf[i * self.stride:i * self.stride + int(n)] for i, n in enumerate(self.lens)]


def prove_sha256_sharded(message, log_blocks_per_shard, params=None, devices=None, in_flight=2):
    """zkhip_prove_sha256_sharded: SHA-256 of a message of any length as a chain of shard proofs dealt over `devices` (None: all visible)"""
    lib = _lib.load()
    params = params or Params(1, 100, 16)
    m = np.frombuffer(bytes(message), dtype=np.uint8) if len(message) else np.zeros(1, dtype=np.uint8)
    n = lib.zkhip_sha256_sharded_count(len(message), log_blocks_per_shard)
    stride = lib.zkhip_sha256_shard_proof_size(log_blocks_per_shard, C.byref(params))
    if n == 0 or stride == 0:
        raise _lib.ZkHipError(-1, "prove_sha256_sharded: bad shape")
    proofs = np.empty(n * stride, dtype=np.uint8)
    lens = (C.c_size_t * n)()
    chain = np.zeros((n + 1, 8), dtype=np.uint32)
    digest = np.zeros(32, dtype=np.uint8)
    devs = (C.c_int * len(devices))(*devices) if devices else None
    check(lib.zkhip_prove_sha256_sharded(devs, len(devices) if devices else 0, m.ctypes.data_as(u8p), len(message), log_blocks_per_shard, C.byref(params), in_flight,
                                         digest.ctypes.data_as(u8p), chain.ctypes.data_as(u32p), proofs.ctypes.data_as(u8p), stride, lens))
    return ShardedSha256(digest.tobytes(), chain, proofs, stride, list(lens), log_blocks_per_shard, len(message))


def verify_sha256_sharded(result, digest=None, params=None, chain=None, proofs=None, message_len=None):
    """-> (rc, failing shard, reason); digest / chain / proofs / message length default to the result's own"""
    lib = _lib.load()
    params = params or Params(1, 100, 16)
    n = len(result.lens)
    lens = (C.c_size_t * n)(*result.lens)
    ch = np.ascontiguousarray(result.chain if chain is None else chain, dtype=np.uint32)
    buf = np.ascontiguousarray(result.buf if proofs is None else proofs, dtype=np.uint8)
    dg = np.frombuffer(bytes(result.digest if digest is None else digest), dtype=np.uint8)
    bad, reason = C.c_size_t(0), C.c_int(0)
    rc = lib.zkhip_verify_sha256_sharded(buf.ctypes.data_as(u8p), result.stride, lens, n, ch.ctypes.data_as(u32p), result.log_blocks, dg.ctypes.data_as(u8p),
                                         result.message_len if message_len is None else message_len, C.byref(params), C.byref(bad), C.byref(reason))
    return rc, bad.value, reason.value


class InnerMachine:
    """zkhip_machine_desc built from Python: chips = [{ln, W, Pw, prog, tab}] tallest first, the machine's key, how its proofs are made"""
    def __init__(self, chips, key_root, n_queries, pow_bits, n_public):
        n = len(chips)
        self.keep = [(np.ascontiguousarray(c["prog"], dtype=np.uint32), np.ascontiguousarray(c["tab"], dtype=np.uint32)) for c in chips]
        self.log_ns = (C.c_int32 * n)(*[int(c["ln"]) for c in chips])
        self.widths = (C.c_uint32 * n)(*[int(c["W"]) for c in chips])
        self.pre_widths = (C.c_uint32 * n)(*[int(c["Pw"]) for c in chips])
        self.progs = (C.POINTER(C.c_uint32) * n)(*[p.ctypes.data_as(u32p) for p, _ in self.keep])
        self.prog_words = (C.c_size_t * n)(*[p.size for p, _ in self.keep])
        self.tabs = (C.POINTER(C.c_uint32) * n)(*[t.ctypes.data_as(u32p) for _, t in self.keep])
        self.tab_words = (C.c_size_t * n)(*[t.size for _, t in self.keep])
        self.desc = _lib.MachineDesc(n, self.log_ns, self.widths, self.pre_widths, self.progs, self.prog_words, self.tabs, self.tab_words,
                                     (C.c_uint32 * 8)(*[int(v) for v in key_root]), int(n_queries), int(pow_bits), int(n_public))
        self.n_public = int(n_public)


def sha256_inner_machine(message_len, key_root, params=None):
    """the keyed SHA-256 machine of messages of this length (zkhip_prove_transcripts' proofs) as an InnerMachine for machine mode"""
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    chips = []
    for which in range(2):
        ln, w, pw = C.c_int(0), C.c_uint32(0), C.c_uint32(0)
        parts = []
        for kind in range(2):
            n = lib.zkhip_sha256_machine_describe(message_len, which, kind, None, 0, C.byref(ln), C.byref(w), C.byref(pw))
            out = np.zeros(max(n, 1), dtype=np.uint32)
            assert n and lib.zkhip_sha256_machine_describe(message_len, which, kind, out.ctypes.data_as(u32p), n, C.byref(ln), C.byref(w), C.byref(pw)) == n
            parts.append(out[:n])
        chips.append(dict(ln=ln.value, W=w.value, Pw=pw.value, prog=parts[0], tab=parts[1]))
    return InnerMachine(chips, key_root, params.num_queries, params.pow_bits, SHA256_PUBLIC)


def machine_verifier_describe(inner, which, kind, n_proofs=1):
    """zkhip_machine_verifier_describe -> (words, log_rows, main width, preprocessed width); kind 0 program, 1 interaction table, 2 preprocessed trace"""
    lib = _lib.load()
    ln, mw, pw = C.c_int(0), C.c_uint32(0), C.c_uint32(0)
    n = lib.zkhip_machine_verifier_describe(C.byref(inner.desc), n_proofs, which, kind, None, 0, C.byref(ln), C.byref(mw), C.byref(pw))
    out = np.zeros(max(n, 1), dtype=np.uint32)
    lib.zkhip_machine_verifier_describe(C.byref(inner.desc), n_proofs, which, kind, out.ctypes.data_as(u32p), n, C.byref(ln), C.byref(mw), C.byref(pw))
    return out[:n], ln.value, mw.value, pw.value


def machine_verifier_host_tables(inner, proofs, public_values, which):
    """zkhip_machine_verifier_host_tables: the main trace of the chip at position `which` as the prover fills it on the host (canonical words, flat); None: refused"""
    lib = _lib.load()
    sps = [np.ascontiguousarray(sp, dtype=np.uint8) for sp in proofs]
    n = len(sps)
    pv = np.ascontiguousarray(np.array([list(v) for v in public_values], dtype=np.uint32).reshape(n, -1)) if inner.n_public else np.zeros((n, 1), dtype=np.uint32)
    ptrs = (u8p * n)(*[sp.ctypes.data_as(u8p) for sp in sps])
    lens = (C.c_size_t * n)(*[sp.size for sp in sps])
    words = lib.zkhip_machine_verifier_host_tables(C.byref(inner.desc), ptrs, lens, n, pv.ctypes.data_as(u32p), inner.n_public, which, None, 0)
    if words == 0:
        return None
    out = np.zeros(words, dtype=np.uint32)
    assert lib.zkhip_machine_verifier_host_tables(C.byref(inner.desc), ptrs, lens, n, pv.ctypes.data_as(u32p), inner.n_public, which, out.ctypes.data_as(u32p), words) == words
    return out


def machine_verifier_key_host(inner, params=None, n_proofs=1):
    params = params or Params()
    vk = np.zeros(8, dtype=np.uint32)
    check(_lib.load().zkhip_machine_verifier_key_host(C.byref(inner.desc), n_proofs, C.byref(params), vk.ctypes.data_as(u32p)))
    return vk


def verify_machine_recursive(inner, proof, public_values, vk, params=None, n_proofs=1):
    """zkhip_verify_machine_recursive: the outer proof against (the inner machine's description, the inner proofs' public values, the key) -> (rc, reason)"""
    params = params or Params()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    pv = np.ascontiguousarray(np.array(list(public_values) or [0], dtype=np.uint32))
    k = np.ascontiguousarray(np.array(vk, dtype=np.uint32))
    reason = C.c_int(0)
    rc = _lib.load().zkhip_verify_machine_recursive(C.byref(inner.desc), pr.ctypes.data_as(u8p), pr.size, pv.ctypes.data_as(u32p), inner.n_public, n_proofs, k.ctypes.data_as(u32p),
                                                    C.byref(params), C.byref(reason))
    return rc, reason.value


def sha256_compress_key_host(message_len, log_blocks_per_shard, inner=None, outer=None):
    """zkhip_sha256_compress_key_host: the key of the compressed chain for a message of this length, without a device -> 8 words"""
    inner, outer = inner or Params(1, 100, 16), outer or Params(1, 100, 16)
    vk = np.zeros(8, dtype=np.uint32)
    check(_lib.load().zkhip_sha256_compress_key_host(message_len, log_blocks_per_shard, C.byref(inner), C.byref(outer), vk.ctypes.data_as(u32p)))
    return vk


def verify_sha256_compressed(proof, digest, message_len, chain, log_blocks_per_shard, vk, inner=None, outer=None):
    """zkhip_verify_sha256_compressed: "digest = SHA-256 of a message of message_len bytes" from ONE proof, the chain and the key -> (rc, reason)"""
    inner, outer = inner or Params(1, 100, 16), outer or Params(1, 100, 16)
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    dg = np.frombuffer(bytes(digest), dtype=np.uint8)
    ch = np.ascontiguousarray(chain, dtype=np.uint32)
    k = np.ascontiguousarray(np.array(vk, dtype=np.uint32))
    reason = C.c_int(0)
    # the C entry reads 8 (n_shards + 1) chain words, n_shards following from the length: another number of shards is another statement
    if ch.size != 8 * (_lib.load().zkhip_sha256_sharded_count(message_len, log_blocks_per_shard) + 1):
        return -6, 1                                               # ZKHIP_ERR_VERIFY
    rc = _lib.load().zkhip_verify_sha256_compressed(pr.ctypes.data_as(u8p), pr.size, dg.ctypes.data_as(u8p), message_len, ch.ctypes.data_as(u32p), log_blocks_per_shard,
                                                    k.ctypes.data_as(u32p), C.byref(inner), C.byref(outer), C.byref(reason))
    return rc, reason.value


def sha256_air():
    """the SHA-256 compression chip's constraint program (u32 words)"""
    lib = _lib.load()
    n = lib.zkhip_sha256_air(None, 0)
    out = np.empty(n, dtype=np.uint32)
    assert lib.zkhip_sha256_air(out.ctypes.data_as(u32p), n) == n
    return out


class PaddedMessage(bytes):
    """padded blocks that remember the message's length (the statement's other half)"""
    message_len = None


def sha256_pad(message):
    lib = _lib.load()
    m = np.frombuffer(bytes(message), dtype=np.uint8) if len(message) else np.zeros(1, dtype=np.uint8)
    n = lib.zkhip_sha256_pad(m.ctypes.data_as(u8p), len(message), None, 0)
    out = np.empty(n, dtype=np.uint8)
    assert lib.zkhip_sha256_pad(m.ctypes.data_as(u8p), len(message), out.ctypes.data_as(u8p), n) == n
    res = PaddedMessage(out.tobytes())
    res.message_len = len(message)
    return res


def sha256_padding_publics(message_len, first_block=0, n_active=None):
    """zkhip_sha256_padding_publics: the 75 public values a verifier derives from the length (whole message: all its blocks)"""
    if n_active is None:
        n_active = (message_len + 8) // 64 + 1 - first_block
    out = np.zeros(SHA256_PADDING_PUBLIC, dtype=np.uint32)
    _lib.load().zkhip_sha256_padding_publics(message_len, first_block, n_active, out.ctypes.data_as(u32p))
    return out


def verify_sha256(proof, digest, params=None, message_len=None):
    """the statement: digest = SHA-256 of a message of message_len bytes"""
    if message_len is None:
        raise TypeError("verify_sha256: the message length is part of the statement")
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    dg = np.frombuffer(bytes(digest), dtype=np.uint8)
    reason = C.c_int(0)
    rc = lib.zkhip_verify_sha256(pr.ctypes.data_as(u8p), pr.size, dg.ctypes.data_as(u8p), message_len, C.byref(params), C.byref(reason))
    return rc, reason.value


def prove_transcripts(messages, params=None, devices=None, in_flight=4, verify=False, keyed=True):
    """zkhip_prove_transcripts: every message proven as the keyed SHA-256 machine in one call, dealt over `devices` (None: all visible);
    verify: each proof is checked against the key inside the call -> (vk, [(digest bytes, proof bytes), ...]).
    keyed=False: zkhip_prove_transcripts_air -- the chip alone (version-7 proofs, what prove_sha256 makes); vk is None"""
    lib = _lib.load()
    params = params or Params(1, 100, 16)
    n = len(messages)
    jobs = (_lib.TranscriptJob * max(n, 1))()
    keep = []
    for i, m in enumerate(messages):
        msg = np.frombuffer(bytes(m), dtype=np.uint8) if len(m) else np.zeros(1, dtype=np.uint8)
        size = lib.zkhip_sha256_machine_proof_size(len(m), C.byref(params)) if keyed else lib.zkhip_sha256_proof_size(len(m), C.byref(params))
        buf = np.empty(max(size, 1), dtype=np.uint8)
        keep.append((msg, buf))
        jobs[i].message = msg.ctypes.data_as(u8p); jobs[i].message_len = len(m)
        jobs[i].proof = buf.ctypes.data_as(u8p); jobs[i].proof_cap = size
    vk = np.zeros(8, dtype=np.uint32)
    devs = (C.c_int * len(devices))(*devices) if devices else None
    if keyed:
        check(lib.zkhip_prove_transcripts(devs, len(devices) if devices else 0, jobs, n, C.byref(params), in_flight, 1 if verify else 0, vk.ctypes.data_as(u32p)))
    else:
        check(lib.zkhip_prove_transcripts_air(devs, len(devices) if devices else 0, jobs, n, C.byref(params), in_flight, 1 if verify else 0))
    return (vk if keyed else None), [(bytes(jobs[i].digest), keep[i][1][: jobs[i].proof_len]) for i in range(n)]


def prove_fri_indices_batch(shard_proofs, log_n, width, public_values, inner=None, outer=None, devices=None, in_flight=4, verify=False):
    """zkhip_prove_fri_indices_batch: the query-phase machine of every shard proof in one call (the compress-like step), dealt over `devices`
    -> [(proof bytes, vk [8], final value [4], capacity [8]), ...]; public_values: one list per shard proof"""
    lib = _lib.load()
    inner, outer = inner or Params(1, 100, 16), outer or Params(1, 100, 16)
    n = len(shard_proofs)
    size = lib.zkhip_fri_indices_proof_size(log_n, inner.num_queries, inner.pow_bits, C.byref(outer))
    jobs = (_lib.FriJob * max(n, 1))()
    keep = []
    for i, sp in enumerate(shard_proofs):
        a = np.ascontiguousarray(sp, dtype=np.uint8)
        pv = np.ascontiguousarray(np.array(public_values[i], dtype=np.uint32))
        buf = np.empty(max(size, 1), dtype=np.uint8)
        keep.append((a, pv, buf))
        jobs[i].shard_proof = a.ctypes.data_as(u8p); jobs[i].shard_proof_len = a.size
        jobs[i].public_values = pv.ctypes.data_as(u32p); jobs[i].n_public = pv.size
        jobs[i].proof = buf.ctypes.data_as(u8p); jobs[i].proof_cap = size
    devs = (C.c_int * len(devices))(*devices) if devices else None
    check(lib.zkhip_prove_fri_indices_batch(devs, len(devices) if devices else 0, jobs, n, log_n, width, C.byref(inner), C.byref(outer), in_flight, 1 if verify else 0))
    return [(keep[i][2][: jobs[i].proof_len], list(jobs[i].vk), list(jobs[i].final_value), list(jobs[i].capacity)) for i in range(n)]


def prove_shard_verifier_batch(shard_proofs, proofs_per_join, log_n, width, public_values, inner=None, outer=None, devices=None, in_flight=4, verify=False):
    """zkhip_prove_shard_verifier_batch: len(shard_proofs) / proofs_per_join joins of one shape, dealt over `devices`, `in_flight` at a time on each
    -> ([join proof bytes, ...], vk [8]); public_values: one list per shard proof"""
    lib = _lib.load()
    inner, outer = inner or Params(1, 100, 16), outer or Params(1, 100, 16)
    sps = [np.ascontiguousarray(sp, dtype=np.uint8) for sp in shard_proofs]
    n = len(sps)
    pv = np.ascontiguousarray(np.array([list(v) for v in public_values], dtype=np.uint32).reshape(max(n, 1), -1))
    n_public = pv.shape[1]
    size = lib.zkhip_shard_verifier_proof_size(log_n, width, inner.num_queries, inner.pow_bits, n_public, proofs_per_join, C.byref(outer))
    if size == 0:
        check(-1)
    n_joins = n // proofs_per_join if proofs_per_join else 0
    buf = np.empty((max(n_joins, 1), size), dtype=np.uint8)
    lens = (C.c_size_t * max(n_joins, 1))()
    ptrs = (u8p * max(n, 1))(*[sp.ctypes.data_as(u8p) for sp in sps])
    plens = (C.c_size_t * max(n, 1))(*[sp.size for sp in sps])
    vk = np.zeros(8, dtype=np.uint32)
    devs = (C.c_int * len(devices))(*devices) if devices else None
    check(lib.zkhip_prove_shard_verifier_batch(devs, len(devices) if devices else 0, ptrs, plens, n, proofs_per_join, log_n, width, pv.ctypes.data_as(u32p), n_public,
                                               C.byref(inner), C.byref(outer), in_flight, 1 if verify else 0, buf.ctypes.data_as(u8p), size, lens, vk.ctypes.data_as(u32p)))
    return [buf[j, : lens[j]] for j in range(n_joins)], vk


def prove_shard_tree(ctx, top_key, join_machine, shard_proofs, proofs_per_join, log_n, width, public_values, inner=None, join_outer=None, top_outer=None, devices=None, in_flight=4):
    """zkhip_prove_shard_tree: shard proofs -> joins of proofs_per_join (in flight, dealt over `devices`) -> ONE proof over the joins on `ctx`
    -> (top proof bytes, [join proof bytes, ...], join vk [8]); join_machine: the InnerMachine of the join machine; public_values: one list per shard proof"""
    lib = _lib.load()
    inner, join_outer, top_outer = inner or Params(1, 100, 16), join_outer or Params(1, 100, 16), top_outer or Params(1, 100, 16)
    sps = [np.ascontiguousarray(sp, dtype=np.uint8) for sp in shard_proofs]
    n = len(sps)
    pv = np.ascontiguousarray(np.array([list(v) for v in public_values], dtype=np.uint32).reshape(max(n, 1), -1))
    n_public = pv.shape[1]
    jsize = lib.zkhip_shard_verifier_proof_size(log_n, width, inner.num_queries, inner.pow_bits, n_public, proofs_per_join, C.byref(join_outer))
    n_joins = n // proofs_per_join if proofs_per_join else 0
    tsize = lib.zkhip_machine_verifier_proof_size(C.byref(join_machine.desc), max(n_joins, 1), C.byref(top_outer))
    if jsize == 0 or tsize == 0:
        check(-1)
    jbuf = np.empty((max(n_joins, 1), jsize), dtype=np.uint8)
    jlens = (C.c_size_t * max(n_joins, 1))()
    ptrs = (u8p * max(n, 1))(*[sp.ctypes.data_as(u8p) for sp in sps])
    plens = (C.c_size_t * max(n, 1))(*[sp.size for sp in sps])
    jvk = np.zeros(8, dtype=np.uint32)
    top = np.empty(tsize, dtype=np.uint8)
    got = C.c_size_t(0)
    devs = (C.c_int * len(devices))(*devices) if devices else None
    check(lib.zkhip_prove_shard_tree(ctx.handle, top_key.handle, C.byref(join_machine.desc), devs, len(devices) if devices else 0, ptrs, plens, n, proofs_per_join, log_n, width,
                                     pv.ctypes.data_as(u32p), n_public, C.byref(inner), C.byref(join_outer), C.byref(top_outer), in_flight, jbuf.ctypes.data_as(u8p), jsize, jlens,
                                     jvk.ctypes.data_as(u32p), top.ctypes.data_as(u8p), tsize, C.byref(got)))
    return top[: got.value], [jbuf[j, : jlens[j]] for j in range(n_joins)], jvk


LKUP_MAGIC, LKUP_SEND, LKUP_RECEIVE = 0x50554B4C, 0, 1


def interaction_table(interactions):
    """[(LKUP_SEND | LKUP_RECEIVE, multiplicity column or None for the constant 1, bus, [value columns]), ...] -> the flat u32 table of include/zkhip.h"""
    body = []
    for sign, mult, bus, cols in interactions:
        body += [sign, 0xFFFFFFFF if mult is None else mult, bus % _lib.P, len(cols)] + list(cols)
    return np.array([LKUP_MAGIC, len(interactions), 3 + len(body)] + body, dtype=np.uint32)


SP1_SHAPED_SPEC = ((20, 96, 3, 1), (20, 32, 3, 0), (19, 64, 2, -1), (18, 128, 4, -1), (16, 256, 8, -1), (14, 40, 1, -1))
SP1_SHAPED_PRE = ((4, 32),)
BUS_SP1 = 300


class Sp1ShapedShard:
    """SP1's shard structure as a KEYED machine (proof version 11; sp1-core-machine's shards, reference Cargo.lock:5822): chips of mixed heights
    (spec: (log_n, width, LogUp pairs, partner) tallest first), each under the synthetic AIR as a constraint program, pair q of a chip sends the
    (a, b) of column group 2q and receives at group 2q + 1 -- in-table (partner -1), or ACROSS two tables of one height that look each other up
    (chip c sends on bus BUS_SP1 + 16 c + q, its partner receives there: the two cumulative sums cancel) --, and preprocessed leading columns
    on the chips of `pre` ((chip, columns), ...: what setup commits once, sp1.rs:113).  Traces are generated on the device
    (zkhip_gen_trace_logup / _cross, stream seed + 100 shard + chip)."""

    def __init__(self, spec=SP1_SHAPED_SPEC, pre=SP1_SHAPED_PRE, n_public=9):
        self.spec, self.pre, self.n_public = [tuple(c) for c in spec], dict(pre), int(n_public)
        self.programs = [air_synthetic(w, n_public) for _, w, _, _ in self.spec]
        self.tables = []
        for c, (ln, w, pairs, partner) in enumerate(self.spec):
            it = []
            for q in range(pairs):
                it.append((LKUP_SEND, None, BUS_SP1 + 16 * c + q, [8 * q, 8 * q + 1]))
                it.append((LKUP_RECEIVE, None, BUS_SP1 + 16 * (c if partner < 0 else partner) + q, [8 * q + 4, 8 * q + 5]))
            self.tables.append(interaction_table(it) if it else None)
        self.log_ns = [c[0] for c in self.spec]
        self.pre_widths = [self.pre.get(c, 0) for c in range(len(self.spec))]
        self.widths = [w - pw for (_, w, _, _), pw in zip(self.spec, self.pre_widths)]                  # main widths
        self.cells = sum(w << ln for ln, w, _, _ in self.spec)

    def gen_traces(self, ctx, seed, shard):
        """-> the full traces [pre | main] of one shard as DeviceBuffers"""
        out = []
        for c, (ln, w, pairs, partner) in enumerate(self.spec):
            if partner < 0:
                out.append(ctx.gen_trace_logup(seed, 100 * shard + c, ln, w, pairs))
            else:
                out.append(ctx.gen_trace_logup_cross(seed, 100 * shard + c, 100 * shard + partner, ln, w, self.spec[partner][1], pairs))
        return out

    def main_chips(self, traces):
        """the chips argument of Context.prove_machine_keyed: the main columns as views of the full traces"""
        return [(DeviceView(t, pw), ln, w, w + pw) for t, ln, w, pw in zip(traces, self.log_ns, self.widths, self.pre_widths)]

    KEY_SHARD = 9999

    def setup(self, ctx, seed, params=None):
        """the key (zkhip_machine_setup) over the preprocessed columns: the leading columns of the chips of `pre`, generated from a stream of their
        own (shard KEY_SHARD) so that every shard of the execution is proven against ONE key, as a program's shards are.  -> (key, the traces it views)"""
        keep = []
        for c, (ln, w, pairs, partner) in enumerate(self.spec):
            keep.append(ctx.gen_trace_logup(seed, 100 * self.KEY_SHARD + c, ln, w, pairs) if self.pre_widths[c] else None)
            assert not (self.pre_widths[c] and (partner >= 0 or self.pre_widths[c] % 8)), "preprocessed columns: whole in-table pairs"
        return ctx.machine_setup([(t, ln, pw, (w + pw) if pw else 0) for t, ln, w, pw in zip(keep, self.log_ns, self.widths, self.pre_widths)], params), keep

    def inner_machine(self, key_root, params=None):
        params = params or Params(1, 100, 16, 0)
        chips = [dict(ln=ln, W=w, Pw=pw, prog=g, tab=t) for ln, w, pw, g, t in zip(self.log_ns, self.widths, self.pre_widths, self.programs, self.tables)]
        return InnerMachine(chips, key_root, params.num_queries, params.pow_bits, self.n_public)


def recursion_witnesses_on_host(enable):
    """zkhip_recursion_witnesses_on_host: 1 = the recursion machines fill their per-query tables on host threads (rounds 4 - 5), 0 = device kernels; returns the previous setting"""
    return int(_lib.load().zkhip_recursion_witnesses_on_host(1 if enable else 0))


def set_lockstep(max_batch, lanes=0):
    """zkhip_set_lockstep: members per lock-step batch of small transcripts (0 / 1 = off), batches in flight per device (0 = keep)"""
    _lib.load().zkhip_set_lockstep(int(max_batch), int(lanes))


def lockstep_stats():
    """(merged launches, member launch requests, rendezvous with differing requests, ns waited at rendezvous, ns issuing merged
    launches, ns in votes) since the library was loaded"""
    out = (C.c_uint64 * 6)()
    _lib.load().zkhip_lockstep_stats(out)
    return tuple(int(v) for v in out)


def p2chip_air():
    """the Poseidon2 chip's constraint program for the parameter set in effect"""
    lib = _lib.load()
    n = lib.zkhip_p2chip_air(None, 0)
    out = np.empty(n, dtype=np.uint32)
    assert lib.zkhip_p2chip_air(out.ctypes.data_as(u32p), n) == n
    return out


def verify_merkle_paths(proof, root, n_paths, params=None):
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    rt = np.ascontiguousarray(root, dtype=np.uint32)
    reason = C.c_int(0)
    rc = lib.zkhip_verify_merkle_paths(pr.ctypes.data_as(u8p), pr.size, rt.ctypes.data_as(u32p), n_paths, C.byref(params), C.byref(reason))
    return rc, reason.value


def verify_sha256_machine(proof, digest, vk, params=None, message_len=None):
    if message_len is None:
        raise TypeError("verify_sha256_machine: the message length is part of the statement")
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    dg = np.frombuffer(bytes(digest), dtype=np.uint8)
    k = np.ascontiguousarray(np.array(vk, dtype=np.uint32))
    reason = C.c_int(0)
    rc = lib.zkhip_verify_sha256_machine(pr.ctypes.data_as(u8p), pr.size, dg.ctypes.data_as(u8p), message_len, k.ctypes.data_as(u32p), C.byref(params), C.byref(reason))
    return rc, reason.value


def air_synthetic(width, n_public):
    lib = _lib.load()
    n = C.c_size_t(0)
    check(lib.zkhip_air_synthetic(width, n_public, None, 0, C.byref(n)))
    out = np.empty(n.value, dtype=np.uint32)
    check(lib.zkhip_air_synthetic(width, n_public, out.ctypes.data_as(u32p), out.size, C.byref(n)))
    return out


def _program_table(programs):
    n = len(programs)
    keep = [(np.ascontiguousarray(p, dtype=np.uint32) if p is not None else None) for p in programs]
    pp = (u32p * n)(*[(p.ctypes.data_as(u32p) if p is not None else None) for p in keep])
    pw = (C.c_size_t * n)(*[(p.size if p is not None else 0) for p in keep])
    return keep, pp, pw


def fri_view_shard(proof, log_n, width, public_values=(), params=None):
    """zkhip_fri_view_shard: what the FRI check of a fold-by-2 shard proof reads -> {"betas": [R][4], "final": [4], "queries": [(index, value[4],
    siblings [R][4])]} (canonical), or raises if the proof is rejected.  Host only."""
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
    R, Q = log_n, params.num_queries
    betas, final = np.zeros(4 * R, dtype=np.uint32), np.zeros(4, dtype=np.uint32)
    idx, vals, sibs = np.zeros(Q, dtype=np.uint32), np.zeros(4 * Q, dtype=np.uint32), np.zeros(4 * Q * R, dtype=np.uint32)
    check(lib.zkhip_fri_view_shard(pr.ctypes.data_as(u8p), pr.size, log_n, width, pv.ctypes.data_as(u32p), pv.size, C.byref(params),
                                   betas.ctypes.data_as(u32p), final.ctypes.data_as(u32p), idx.ctypes.data_as(u32p), vals.ctypes.data_as(u32p),
                                   sibs.ctypes.data_as(u32p)))
    return {"betas": betas.reshape(R, 4).tolist(), "final": final.tolist(),
            "queries": [(int(idx[q]), vals[4 * q:4 * q + 4].tolist(), sibs[4 * q * R:4 * (q + 1) * R].reshape(R, 4).tolist()) for q in range(Q)]}


def fri_view_transcript(proof, log_n, width, public_values=(), params=None):
    """zkhip_fri_view_transcript -> (roots [R][8], betas [R][4], capacity [8], pending inputs): the Fiat-Shamir side of the FRI view
    (fri_view_witness: the proof-of-work witness, the transcript array's last word)"""
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
    R = log_n
    roots, betas, tr = np.zeros(8 * R, dtype=np.uint32), np.zeros(4 * R, dtype=np.uint32), np.zeros(10, dtype=np.uint32)
    check(lib.zkhip_fri_view_transcript(pr.ctypes.data_as(u8p), pr.size, log_n, width, pv.ctypes.data_as(u32p), pv.size, C.byref(params),
                                        roots.ctypes.data_as(u32p), betas.ctypes.data_as(u32p), tr.ctypes.data_as(u32p)))
    return roots.reshape(R, 8).tolist(), betas.reshape(R, 4).tolist(), tr[:8].tolist(), int(tr[8])


def fri_view_witness(proof, log_n, width, public_values=(), params=None):
    """the proof-of-work witness of a shard proof (zkhip_fri_view_transcript's transcript[9])"""
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
    roots, betas, tr = np.zeros(8 * log_n, dtype=np.uint32), np.zeros(4 * log_n, dtype=np.uint32), np.zeros(10, dtype=np.uint32)
    check(lib.zkhip_fri_view_transcript(pr.ctypes.data_as(u8p), pr.size, log_n, width, pv.ctypes.data_as(u32p), pv.size, C.byref(params),
                                        roots.ctypes.data_as(u32p), betas.ctypes.data_as(u32p), tr.ctypes.data_as(u32p)))
    return int(tr[9])


def fri_view_all(proof, log_n, width, public_values=(), params=None):
    """zkhip_fri_view_all: the view of fri_view_shard_paths, the challenger's capacity and the proof-of-work witness from ONE pass over the proof
    -> (view, capacity [8], witness)"""
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
    R, Q = log_n, params.num_queries
    pw = lib.zkhip_fri_view_path_words(R)
    betas, final, idx, vals, sibs = (np.zeros(n, dtype=np.uint32) for n in (4 * R, 4, Q, 4 * Q, 4 * Q * R))
    roots, paths, tr = np.zeros(8 * R, dtype=np.uint32), np.zeros(pw * Q, dtype=np.uint32), np.zeros(10, dtype=np.uint32)
    check(lib.zkhip_fri_view_all(pr.ctypes.data_as(u8p), pr.size, log_n, width, pv.ctypes.data_as(u32p), pv.size, C.byref(params),
                                 *[a.ctypes.data_as(u32p) for a in (betas, final, idx, vals, sibs, roots, paths, tr)]))
    view = {"betas": betas.reshape(R, 4).tolist(), "final": final.tolist(),
            "queries": [(int(idx[q]), vals[4 * q:4 * q + 4].tolist(), sibs[4 * q * R:4 * (q + 1) * R].reshape(R, 4).tolist()) for q in range(Q)],
            "roots": roots.reshape(R, 8).tolist(), "paths": []}
    per_q = paths.reshape(Q, pw) if Q else paths
    for q in range(Q):
        off, pq = 0, []
        for l in range(R):
            pq.append(per_q[q, off:off + 8 * (R - l)].reshape(R - l, 8).tolist())
            off += 8 * (R - l)
        view["paths"].append(pq)
    return view, tr[:8].tolist(), int(tr[9])


def fri_view_shard_paths(proof, log_n, width, public_values=(), params=None):
    """zkhip_fri_view_shard_paths: the view of fri_view_shard plus "roots": [R][8] and, per query, "paths": [R] lists of (R - l) digests"""
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
    R, Q = log_n, params.num_queries
    per = int(lib.zkhip_fri_view_path_words(R))
    betas, final = np.zeros(4 * R, dtype=np.uint32), np.zeros(4, dtype=np.uint32)
    idx, vals, sibs = np.zeros(Q, dtype=np.uint32), np.zeros(4 * Q, dtype=np.uint32), np.zeros(4 * Q * R, dtype=np.uint32)
    roots, paths = np.zeros(8 * R, dtype=np.uint32), np.zeros(per * Q, dtype=np.uint32)
    check(lib.zkhip_fri_view_shard_paths(pr.ctypes.data_as(u8p), pr.size, log_n, width, pv.ctypes.data_as(u32p), pv.size, C.byref(params),
                                         betas.ctypes.data_as(u32p), final.ctypes.data_as(u32p), idx.ctypes.data_as(u32p), vals.ctypes.data_as(u32p),
                                         sibs.ctypes.data_as(u32p), roots.ctypes.data_as(u32p), paths.ctypes.data_as(u32p)))
    view = {"betas": betas.reshape(R, 4).tolist(), "final": final.tolist(), "roots": roots.reshape(R, 8).tolist(),
            "queries": [(int(idx[q]), vals[4 * q:4 * q + 4].tolist(), sibs[4 * q * R:4 * (q + 1) * R].reshape(R, 4).tolist()) for q in range(Q)], "paths": []}
    for q in range(Q):
        off, pq = q * per, []
        for l in range(R):
            n = 8 * (R - l)
            pq.append(paths[off:off + n].reshape(R - l, 8).tolist())
            off += n
        view["paths"].append(pq)
    return view


def _fri_layers_arrays(view):
    R, Q, betas, idx, vals, sibs = _fri_view_arrays(view)
    roots = np.ascontiguousarray(np.array(view["roots"], dtype=np.uint32).reshape(-1))
    paths = np.ascontiguousarray(np.array([w for pq in view["paths"] for layer in pq for dg in layer for w in dg], dtype=np.uint32))
    return R, Q, betas, idx, vals, sibs, roots, paths


def fri_layers_programs(layers):
    """-> (the Poseidon2 chip's FRI-layers variant, the fold chip's wired form) as the library builds them"""
    lib = _lib.load()
    out = []
    for f in (lib.zkhip_p2chip_air_fri_layers, lib.zkhip_fri_layers_chip_air):
        n = f(layers, None, 0)
        buf = np.zeros(n, dtype=np.uint32)
        assert n and f(layers, buf.ctypes.data_as(u32p), n) == n
        out.append(buf)
    return out


def fri_transcript_programs(layers):
    """-> (the Poseidon2 chip's transcript variant, the fold chip's wired form with the transcript machine's public values)"""
    lib = _lib.load()
    out = []
    for f in (lib.zkhip_p2chip_air_fri_transcript, lib.zkhip_fri_transcript_chip_air):
        n = f(layers, None, 0)
        buf = np.zeros(n, dtype=np.uint32)
        assert n and f(layers, buf.ctypes.data_as(u32p), n) == n
        out.append(buf)
    return out


def fri_indices_programs(layers, inner_pow_bits):
    """-> (the Poseidon2 chip with query-phase rows, the SAMPLES chip): what the query-phase machine adds to the transcript machine"""
    lib = _lib.load()
    out = []
    for which in (0, 1):
        n = lib.zkhip_fri_indices_program(which, layers, inner_pow_bits, None, 0)
        buf = np.zeros(n, dtype=np.uint32)
        assert n and lib.zkhip_fri_indices_program(which, layers, inner_pow_bits, buf.ctypes.data_as(u32p), n) == n
        out.append(buf)
    return out


def verify_shard_recursive(proof, log_n, width, n_queries, inner_pow_bits, public_values, vk, params=None, n_proofs=1, program=None):
    """zkhip_verify_shard_recursive (host): the inner proofs' shape and public values (a flat list: proof 0's, then proof 1's, ...) and the shape's key --
    no byte of an inner proof"""
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32).ravel())
    k = np.ascontiguousarray(np.array(vk, dtype=np.uint32))
    reason = C.c_int(0)
    if program is None:
        rc = lib.zkhip_verify_shard_recursive(pr.ctypes.data_as(u8p), pr.size, log_n, width, n_queries, inner_pow_bits, pv.ctypes.data_as(u32p), pv.size // max(n_proofs, 1), n_proofs,
                                              k.ctypes.data_as(u32p), C.byref(params), C.byref(reason))
    else:
        pg = np.ascontiguousarray(program, dtype=np.uint32)
        rc = lib.zkhip_verify_shard_recursive_air(pg.ctypes.data_as(u32p), pg.size, pr.ctypes.data_as(u8p), pr.size, log_n, width, n_queries, inner_pow_bits, pv.ctypes.data_as(u32p),
                                                  pv.size // max(n_proofs, 1), n_proofs, k.ctypes.data_as(u32p), C.byref(params), C.byref(reason))
    return rc, reason.value


def shard_verifier_key_host(log_n, width, n_queries, inner_pow_bits, n_public, params=None, n_proofs=1, program=None):
    """zkhip_shard_verifier_key_host[_air]: the key of the shape (and program) computed on the HOST (no context, no device) -> 8 canonical words"""
    params = params or Params()
    vk = np.zeros(8, dtype=np.uint32)
    if program is None:
        check(_lib.load().zkhip_shard_verifier_key_host(log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, C.byref(params), vk.ctypes.data_as(u32p)))
    else:
        pg = np.ascontiguousarray(program, dtype=np.uint32)
        check(_lib.load().zkhip_shard_verifier_key_host_air(pg.ctypes.data_as(u32p), pg.size, log_n, width, n_queries, inner_pow_bits, n_public, n_proofs, C.byref(params), vk.ctypes.data_as(u32p)))
    return vk


def machine_key_host(traces, log_ns, params=None):
    """zkhip_machine_key_host: traces[c] = None or a [2^log_n][pre_width] array of MONTGOMERY words (host) -> the key's 8 canonical words"""
    params = params or Params()
    arrs = [None if t is None else np.ascontiguousarray(t, dtype=np.uint32) for t in traces]
    ptrs = (u32p * len(arrs))(*[None if a is None else a.ctypes.data_as(u32p) for a in arrs])
    lns = (C.c_int32 * len(arrs))(*[int(x) for x in log_ns])
    pws = np.array([0 if a is None else a.shape[1] for a in arrs], dtype=np.uint32)
    root = np.zeros(8, dtype=np.uint32)
    check(_lib.load().zkhip_machine_key_host(ptrs, lns, pws.ctypes.data_as(u32p), len(arrs), C.byref(params), root.ctypes.data_as(u32p)))
    return root


def shard_verifier_max_proofs(log_n, width, n_queries, inner_pow_bits, n_public, outer=None, program=None):
    """zkhip_shard_verifier_max_proofs[_air] (host): how many shard proofs of this shape ONE join takes (outer: the outer proof's params; None = blowup 2)"""
    o = C.byref(outer) if outer is not None else None
    if program is None:
        return int(_lib.load().zkhip_shard_verifier_max_proofs(log_n, width, n_queries, inner_pow_bits, n_public, o))
    pg = np.ascontiguousarray(program, dtype=np.uint32)
    return int(_lib.load().zkhip_shard_verifier_max_proofs_air(pg.ctypes.data_as(u32p), pg.size, log_n, width, n_queries, inner_pow_bits, n_public, o))


def shard_verifier_describe(log_n, width, n_queries, inner_pow_bits, n_public, which, kind, n_proofs=1, program=None):
    """zkhip_shard_verifier_describe[_air] -> (words, log_rows, main width, preprocessed width); kind 0 program, 1 interaction table, 2 preprocessed trace"""
    lib = _lib.load()
    ln, mw, pw = C.c_int(0), C.c_uint32(0), C.c_uint32(0)
    shape = (log_n, width, n_queries, inner_pow_bits, n_public, n_proofs)
    if program is None:
        fn = lambda *a: lib.zkhip_shard_verifier_describe(*shape, *a)
    else:
        pg = np.ascontiguousarray(program, dtype=np.uint32)
        fn = lambda *a: lib.zkhip_shard_verifier_describe_air(pg.ctypes.data_as(u32p), pg.size, *shape, *a)
    n = fn(which, kind, None, 0, C.byref(ln), C.byref(mw), C.byref(pw))
    out = np.zeros(max(n, 1), dtype=np.uint32)
    fn(which, kind, out.ctypes.data_as(u32p), n, C.byref(ln), C.byref(mw), C.byref(pw))
    return out[:n], ln.value, mw.value, pw.value


def verify_fri_indices(proof, final, capacity, layers, n_queries, inner_pow_bits, vk, params=None):
    """zkhip_verify_fri_indices: the final value, the challenger's capacity, the key -- no challenge, no index"""
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    f = np.ascontiguousarray(np.array(final, dtype=np.uint32))
    c8 = np.ascontiguousarray(np.array(capacity, dtype=np.uint32))
    k = np.ascontiguousarray(np.array(vk, dtype=np.uint32))
    reason = C.c_int(0)
    rc = lib.zkhip_verify_fri_indices(pr.ctypes.data_as(u8p), pr.size, layers, n_queries, inner_pow_bits, f.ctypes.data_as(u32p), c8.ctypes.data_as(u32p),
                                      k.ctypes.data_as(u32p), C.byref(params), C.byref(reason))
    return rc, reason.value






def _fri_view_arrays(view):
    R, Q = len(view["betas"]), len(view["queries"])
    betas = np.ascontiguousarray(np.array(view["betas"], dtype=np.uint32).reshape(-1))
    idx = np.ascontiguousarray(np.array([q[0] for q in view["queries"]], dtype=np.uint32))
    vals = np.ascontiguousarray(np.array([q[1] for q in view["queries"]], dtype=np.uint32).reshape(-1))
    sibs = np.ascontiguousarray(np.array([q[2] for q in view["queries"]], dtype=np.uint32).reshape(-1))
    return R, Q, betas, idx, vals, sibs


def fri_chip_air(layers):
    lib = _lib.load()
    n = lib.zkhip_fri_chip_air(layers, None, 0)
    out = np.zeros(n, dtype=np.uint32)
    assert n and lib.zkhip_fri_chip_air(layers, out.ctypes.data_as(u32p), n) == n
    return out




class MachineKey:
    """the proving key of a keyed machine (device-resident preprocessed traces, LDEs and tree) + what the verifier needs of it"""

    def __init__(self, ctx, handle, root, pre_widths):
        self.ctx, self.handle, self.root, self.pre_widths = ctx, handle, root, pre_widths

    def close(self):
        if self.handle:
            self.ctx.lib.zkhip_machine_key_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def verify_machine_keyed(proof, log_ns, widths, pre_widths, root, programs, tables, public_values=(), params=None):
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
    rt = np.ascontiguousarray(np.array(root, dtype=np.uint32))
    n = len(log_ns)
    ln = (C.c_int32 * n)(*[int(x) for x in log_ns])
    ws = (C.c_uint32 * n)(*[int(x) for x in widths])
    pws = (C.c_uint32 * n)(*[int(x) for x in pre_widths])
    kp, pp, pw = _program_table(programs)
    kt, tp, tw = _program_table(tables)
    reason = C.c_int(0)
    rc = lib.zkhip_verify_machine_keyed(pr.ctypes.data_as(u8p), pr.size, ln, ws, pws, rt.ctypes.data_as(u32p), pp, pw, tp, tw, n,
                                        pv.ctypes.data_as(u32p), pv.size, C.byref(params), C.byref(reason))
    return rc, reason.value


def verify_machine(proof, log_ns, widths, programs, tables, public_values=(), params=None):
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
    n = len(log_ns)
    ln = (C.c_int32 * n)(*[int(x) for x in log_ns])
    ws = (C.c_uint32 * n)(*[int(x) for x in widths])
    kp, pp, pw = _program_table(programs)
    kt, tp, tw = _program_table(tables)
    reason = C.c_int(0)
    rc = lib.zkhip_verify_machine(pr.ctypes.data_as(u8p), pr.size, ln, ws, pp, pw, tp, tw, n, pv.ctypes.data_as(u32p), pv.size, C.byref(params), C.byref(reason))
    return rc, reason.value


def verify_chips_air(proof, log_ns, widths, programs, public_values=(), params=None):
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
    n = len(log_ns)
    ln = (C.c_int32 * n)(*[int(x) for x in log_ns])
    ws = (C.c_uint32 * n)(*[int(x) for x in widths])
    keep, pp, pw = _program_table(programs)
    reason = C.c_int(0)
    rc = lib.zkhip_verify_chips_air(pr.ctypes.data_as(u8p), pr.size, ln, ws, pp, pw, n, pv.ctypes.data_as(u32p), pv.size, C.byref(params), C.byref(reason))
    return rc, reason.value


def verify_chips(proof, log_ns, widths, public_values=(), params=None, pairs=None, partners=None):
    params = params or Params(1, 100, 16)
    lib = _lib.load()
    pr = np.ascontiguousarray(proof, dtype=np.uint8)
    pv = np.ascontiguousarray(np.array(public_values, dtype=np.uint32))
    n = len(log_ns)
    ln = (C.c_int32 * n)(*[int(x) for x in log_ns])
    ws = (C.c_uint32 * n)(*[int(x) for x in widths])
    prs = (C.c_int32 * n)(*[int(x) for x in pairs]) if pairs is not None else None
    pas = (C.c_int32 * n)(*[int(x) for x in partners]) if partners is not None else None
    reason = C.c_int(0)
    rc = lib.zkhip_verify_chips(pr.ctypes.data_as(u8p), pr.size, ln, ws, prs, pas, n, pv.ctypes.data_as(u32p), pv.size, C.byref(params), C.byref(reason))
    return rc, reason.value
