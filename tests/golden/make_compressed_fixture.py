#!/usr/bin/env python3
"""Makes tests/golden/proofs/compressed_blob_8x8x3.bin ON A GPU BOX: one call of the host mirror's compress stage (zktls_guest_prove_compressed,
zktls_amd/host: core -> compress, sp1.rs:116) for a plan of three 2^8 x 8 shards -- the blob holds ONE outer proof and the key of the shape.
tests/test_host_mirror_cpu.py then checks it with NO device: the key from zktls_compress_key_host (host cores), zktls_verify_compressed_blob.
    gpurun -- python3 tests/golden/make_compressed_fixture.py gpurun_out/compressed_blob_8x8x3.bin [gpurun_out/compressed_tree_blob_5x8x5.bin]   (then copy to tests/golden/proofs/)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CBOR, ELF = b"\xa1transcript", b"\x7fELFprog"
PLAN = (8, 8, 3, 10, 8)          # log_n, width, shards, queries, proof-of-work bits


class Plan(C.Structure):
    _fields_ = [("log_n", C.c_int32), ("width", C.c_uint32), ("shards", C.c_uint32), ("num_queries", C.c_int32), ("pow_bits", C.c_int32)]


if __name__ == "__main__":
    out_path = sys.argv[1]
    L = C.CDLL(os.path.join(ROOT, "zktls_amd", "libzktls_guest_prover.so"))
    u8pp, szp = C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)
    L.zktls_guest_prove_compressed.argtypes = [C.c_int, C.c_int, C.POINTER(Plan), C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, u8pp, szp, u8pp, szp, C.c_char_p, C.c_size_t]
    plan = Plan(*PLAN)
    out, outn, pr, prn = C.POINTER(C.c_uint8)(), C.c_size_t(), C.POINTER(C.c_uint8)(), C.c_size_t()
    err = C.create_string_buffer(512)
    rc = L.zktls_guest_prove_compressed(0, 2, C.byref(plan), CBOR, len(CBOR), ELF, len(ELF), C.byref(out), C.byref(outn), C.byref(pr), C.byref(prn), err, 512)
    assert rc == 0, err.value
    blob = C.string_at(pr, prn.value)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    open(out_path, "wb").write(blob)
    print("wrote %d bytes to %s" % (len(blob), out_path))
    if len(sys.argv) > 2:
        # ... and a TREE blob: five 2^5 x 8 shards (2 queries, no proof of work) in joins of at most two (zktls_set_compress_join_size) -> three joins -> ONE
        # proof above them (machine mode: csrc/machine_verifier.inl; blob flag TREE).  tests/golden/proofs/compressed_tree_blob_5x8x5.bin
        L.zktls_set_compress_join_size.argtypes = [C.c_uint32]
        L.zktls_set_compress_join_size(2)
        tplan = Plan(5, 8, 5, 2, 0)
        rc = L.zktls_guest_prove_compressed(0, 2, C.byref(tplan), CBOR, len(CBOR), ELF, len(ELF), C.byref(out), C.byref(outn), C.byref(pr), C.byref(prn), err, 512)
        assert rc == 0, err.value
        tblob = C.string_at(pr, prn.value)
        open(sys.argv[2], "wb").write(tblob)
        print("wrote %d bytes to %s" % (len(tblob), sys.argv[2]))
