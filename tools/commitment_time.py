"""End-to-end latency of the input-commitment guest through the host mirror (GPU box): python tools/commitment_time.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
L = C.CDLL(os.path.join(ROOT, "zktls_amd", "libzktls_guest_prover.so"))
L.zktls_guest_prove_commitment.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                           C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.POINTER(C.POINTER(C.c_uint8)),
                                           C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
L.zktls_free.argtypes = [C.c_void_p]
cbor = open(os.path.join(ROOT, "tests", "golden", "reference", "guest_input0.cbor"), "rb").read()
for i in range(6):
    out, outn, pr, prn = C.POINTER(C.c_uint8)(), C.c_size_t(), C.POINTER(C.c_uint8)(), C.c_size_t()
    err = C.create_string_buffer(512)
    t0 = time.perf_counter()
    rc = L.zktls_guest_prove_commitment(0, 0, 2, 100, 16, cbor, len(cbor), b"\x7fELF", 4, C.byref(out), C.byref(outn), C.byref(pr), C.byref(prn), err, 512)
    dt = time.perf_counter() - t0
    assert rc == 0, err.value
    print("request %d: %d-byte transcript, %.1f ms end to end (prove + verify), proof %d bytes" % (i, len(cbor), dt * 1e3, prn.value))
    L.zktls_free(out); L.zktls_free(pr)

# the same through setup -> prove -> verify: the keyed SHA-256 machine (chip + preprocessed range table); the first request pays setup
L.zktls_guest_prove_commitment_keyed.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                                 C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.POINTER(C.POINTER(C.c_uint8)),
                                                 C.POINTER(C.c_size_t), C.c_char_p, C.c_char_p, C.c_size_t]
for i in range(6):
    out, outn, pr, prn = C.POINTER(C.c_uint8)(), C.c_size_t(), C.POINTER(C.c_uint8)(), C.c_size_t()
    err, vk = C.create_string_buffer(512), C.create_string_buffer(64)
    t0 = time.perf_counter()
    rc = L.zktls_guest_prove_commitment_keyed(0, 2, 100, 16, cbor, len(cbor), b"\x7fELF", 4, C.byref(out), C.byref(outn), C.byref(pr), C.byref(prn), vk, err, 512)
    dt = time.perf_counter() - t0
    assert rc == 0, err.value
    print("keyed request %d: %.1f ms end to end (setup%s + prove + verify), proof %d bytes" % (i, dt * 1e3, "" if i == 0 else " cached", prn.value))
    L.zktls_free(out); L.zktls_free(pr)
