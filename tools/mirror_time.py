"""Throughput of the host mirror (prove + CPU verification of every shard, as sp1.rs:116-120) on a request of `shards` full-size shards:
python tools/mirror_time.py [shards=8]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "zktls_amd", "libzktls_guest_prover.so"))


class Plan(C.Structure):
    _fields_ = [("log_n", C.c_int32), ("width", C.c_uint32), ("shards", C.c_uint32), ("num_queries", C.c_int32), ("pow_bits", C.c_int32)]


L.zktls_guest_prove.argtypes = [C.c_int, C.c_int, C.POINTER(Plan), C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t),
                                C.c_char_p, C.c_size_t]
L.zktls_free.argtypes = [C.c_void_p]
shards = int(sys.argv[1]) if len(sys.argv) > 1 else 8
plan = Plan(20, 256, shards, 100, 16)
for i in range(3):
    out, outn, pr, prn = C.POINTER(C.c_uint8)(), C.c_size_t(), C.POINTER(C.c_uint8)(), C.c_size_t()
    err = C.create_string_buffer(512)
    t0 = time.perf_counter()
    rc = L.zktls_guest_prove(0, 2, C.byref(plan), b"\xa1input", 6, b"\x7fELF", 4, C.byref(out), C.byref(outn), C.byref(pr), C.byref(prn), err, 512)
    dt = time.perf_counter() - t0
    assert rc == 0, err.value
    print("request %d: %d shards of 2^20 x 256 proven AND verified in %.1f ms = %.1f ms per shard" % (i, shards, dt * 1e3, dt * 1e3 / shards))
    L.zktls_free(out); L.zktls_free(pr)
