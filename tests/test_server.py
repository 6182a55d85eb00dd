"""The prover server (zktls_amd/server/moongate_hip.cpp; SURVEY.md 8b plug point 4, 8f-3): Twirp-over-HTTP transport of the
endpoint the reference's SP1 path already calls (sp1.rs:86-90, prove.rs:45-47).  CPU part: routing, protobuf framing, Twirp error
mapping, and the loud failure without a device.  GPU part: ProveCore returns a batch of verified proofs."""
import http.client
import json
import os
import socket
import struct
import subprocess
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "zktls_amd", "moongate-hip")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture()
def server():
    port = _free_port()
    proc = subprocess.Popen([BIN, "--port", str(port)], stderr=subprocess.PIPE)
    line = proc.stderr.readline()                      # "listening on ..."
    assert b"listening" in line, line
    yield port
    proc.terminate()
    proc.wait(timeout=30)


def pb_bytes(data):
    n, out = len(data), bytearray([0x0A])
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            break
    return bytes(out) + data


def pb_field1(msg):
    if not msg:
        return b""
    assert msg[0] == 0x0A
    n, sh, p = 0, 0, 1
    while True:
        b = msg[p]; p += 1
        n |= (b & 0x7F) << sh; sh += 7
        if not b & 0x80:
            break
    assert p + n == len(msg)
    return msg[p:p + n]


def call(port, method, body, ctype="application/protobuf", verb="POST", path=None, timeout=600):
    c = http.client.HTTPConnection("127.0.0.1", port, timeout=timeout)
    c.request(verb, path or "/twirp/api.ProverService/" + method, body=body, headers={"Content-Type": ctype})
    r = c.getresponse()
    data = r.read()
    c.close()
    return r.status, r.getheader("Content-Type"), data


def prove_payload(log_n, width, shards, queries, pow_bits, cbor, elf, backend=0, device=0):
    return (b"ZKMG" + struct.pack("<IiIIiiIi", 1, log_n, width, shards, queries, pow_bits, backend, device)
            + struct.pack("<I", len(cbor)) + cbor + struct.pack("<I", len(elf)) + elf)


def compress_payload(log_n, width, shards, queries, pow_bits, cbor, elf, blob, device=0, per_join=0):
    return (b"ZKMC" + struct.pack("<IiIIiiiI", 1, log_n, width, shards, queries, pow_bits, device, per_join)
            + struct.pack("<I", len(cbor)) + cbor + struct.pack("<I", len(elf)) + elf + struct.pack("<I", len(blob)) + blob)


def test_routing_framing_and_twirp_errors(server):
    from zktls_amd import _lib
    port = server
    st, ct, body = call(port, "Ready", b"")
    assert st == 200 and ct == "application/protobuf"
    assert body == (b"\x08\x01" if _lib.device_count() > 0 else b"")
    # Setup: result = the request digest of the program (the same words the glue binds proofs to)
    elf = b"\x7fELFprogram" * 3
    st, ct, body = call(port, "Setup", pb_bytes(elf))
    assert st == 200
    import ctypes as C
    d = (C.c_uint32 * 8)()
    assert _lib.load().zkhip_request_digest(None, 0, elf, len(elf), d) == 0
    assert pb_field1(body) == bytes(d)
    # Twirp errors: JSON {"code", "msg"}, status by code
    for args, status, code in ((dict(method="Nope", body=b""), 404, "bad_route"),
                               (dict(method="Ready", body=b"", verb="GET"), 404, "bad_route"),
                               (dict(method="Ready", body=b"{}", ctype="application/json"), 404, "bad_route"),
                               (dict(method="Ready", body=b"", path="/other/Ready"), 404, "bad_route"),
                               (dict(method="Shrink", body=pb_bytes(b"x")), 501, "unimplemented"),
                               (dict(method="Wrap", body=pb_bytes(b"x")), 501, "unimplemented"),
                               (dict(method="Compress", body=pb_bytes(b"x")), 400, "invalid_argument"),
                               (dict(method="Compress", body=pb_bytes(compress_payload(6, 8, 2, 10, 4, b"in", b"elf", b"blob")[:-1])), 400, "invalid_argument"),
                               (dict(method="Compress", body=pb_bytes(compress_payload(6, 8, 0, 10, 4, b"in", b"elf", b"blob"))), 400, "invalid_argument"),
                               (dict(method="Compress", body=pb_bytes(compress_payload(6, 8, 2, 10, 4, b"in", b"elf", b"blob", per_join=5000))), 400, "invalid_argument"),
                               (dict(method="Compress", body=pb_bytes(compress_payload(40, 8, 2, 10, 4, b"in", b"elf", b"blob"))), 400, "invalid_argument"),
                               (dict(method="Setup", body=b"\x0a\x05ab"), 400, "malformed"),
                               (dict(method="Setup", body=pb_bytes(b"")), 400, "invalid_argument"),
                               (dict(method="ProveCore", body=pb_bytes(b"not a payload")), 400, "invalid_argument"),
                               (dict(method="ProveCore", body=pb_bytes(prove_payload(6, 8, 1, 10, 4, b"in", b"elf")[:-1])), 400, "invalid_argument")):
        st, ct, body = call(port, **args)
        assert st == status and ct == "application/json", (args, st, body)
        assert json.loads(body)["code"] == code
    if _lib.device_count() == 0:
        st, ct, body = call(port, "ProveCore", pb_bytes(prove_payload(6, 8, 2, 10, 4, b"input", b"elf")))
        assert st == 503 and json.loads(body)["code"] == "unavailable" and "no CPU fallback" in json.loads(body)["msg"]
        st, ct, body = call(port, "Compress", pb_bytes(compress_payload(6, 8, 2, 10, 4, b"input", b"elf", b"blob")))
        assert st == 503 and json.loads(body)["code"] == "unavailable" and "no CPU fallback" in json.loads(body)["msg"]
    else:
        st, ct, body = call(port, "Compress", pb_bytes(compress_payload(6, 8, 2, 10, 4, b"input", b"elf", b"not a batch blob")))
        assert st == 400 and json.loads(body)["code"] == "invalid_argument" and "batch blob" in json.loads(body)["msg"]


def test_plan_from_the_wire_is_bounded_before_anything_is_sized_by_it(server):
    """advice r2 (medium): the plan fields come from an unauthenticated peer -- shards = 2^32 - 1 used to value-initialise ~100 GB
    of proof slots.  Every out-of-range field is an invalid_argument, answered at once, and the server keeps serving."""
    port = server
    bad = [dict(shards=0xFFFFFFFF), dict(shards=5000), dict(log_n=40), dict(log_n=2), dict(width=0), dict(width=4096), dict(queries=0),
           dict(queries=100000), dict(pow_bits=64), dict(device=1 << 20), dict(device=-7), dict(shards=4096, log_n=22, width=1024)]
    for kw in bad:
        a = dict(log_n=10, width=16, shards=2, queries=10, pow_bits=4, device=0)
        a.update(kw)
        t0 = time.time()
        st, ct, body = call(port, "ProveCore", pb_bytes(prove_payload(a["log_n"], a["width"], a["shards"], a["queries"], a["pow_bits"], b"in", b"elf",
                                                                      device=a["device"])), timeout=30)
        assert st == 400 and json.loads(body)["code"] == "invalid_argument", (kw, st, body)
        assert time.time() - t0 < 5
    st, _, _ = call(port, "Ready", b"")
    assert st == 200


def test_declared_body_that_never_arrives_allocates_nothing_and_times_out():
    """a peer that announces a large body and stalls: the reader grows its buffer only with bytes that arrive, the socket has a
    receive timeout, and the next client is served"""
    port = _free_port()
    proc = subprocess.Popen([BIN, "--port", str(port)], stderr=subprocess.PIPE)
    try:
        assert b"listening" in proc.stderr.readline()
        s = socket.create_connection(("127.0.0.1", port))
        s.sendall(b"POST /twirp/api.ProverService/Setup HTTP/1.1\r\nContent-Type: application/protobuf\r\nContent-Length: 60000000\r\n\r\nabc")
        time.sleep(0.5)
        import resource  # noqa: F401  (the check below reads the server's RSS from /proc)
        rss_kb = int([ln for ln in open("/proc/%d/status" % proc.pid) if ln.startswith("VmRSS")][0].split()[1])
        assert rss_kb < 200 * 1024, rss_kb            # 60 MB announced, nothing allocated for it
        s.close()                                      # the stalled peer goes away (otherwise SO_RCVTIMEO ends the wait after 20 s)
        st, _, _ = call(port, "Ready", b"", timeout=60)
        assert st == 200
        # over the cap: refused outright
        s = socket.create_connection(("127.0.0.1", port))
        s.sendall(b"POST /twirp/api.ProverService/Setup HTTP/1.1\r\nContent-Type: application/protobuf\r\nContent-Length: 99999999999\r\n\r\n")
        assert b"malformed" in s.recv(4096)
        s.close()
    finally:
        proc.terminate()
        proc.wait(timeout=30)


@pytest.mark.gpu
def test_prove_core_returns_verified_proofs(server):
    from zktls_amd._lib import Params
    from zktls_amd.device import verify_shard
    port = server
    cbor, elf = b"\xa1transcript" * 9, b"\x7fELFguest"
    st, ct, body = call(port, "ProveCore", pb_bytes(prove_payload(10, 16, 3, 20, 8, cbor, elf)))
    assert st == 200, body
    res = pb_field1(body)
    on = struct.unpack_from("<I", res)[0]
    output, blob = res[4:4 + on], res[4 + on:]
    digest = list(struct.unpack("<8I", output))
    magic, version, flags, count = struct.unpack_from("<4I", blob)
    assert (magic, version, flags, count) == (0x42544B5A, 2, 1, 3)              # "ZKTB", flagged SYNTHETIC
    off = 16
    for s in range(3):
        ln = struct.unpack_from("<I", blob, off)[0]
        proof = np.frombuffer(blob[off + 4:off + 4 + ln], dtype=np.uint8)
        assert verify_shard(proof, 10, 16, digest + [s], Params(1, 20, 8)) == (0, 0)
        off += 4 + ln
    assert off == len(blob)
    # the RISC Zero shape through the same endpoint, and a second request on the same server
    st, ct, body = call(port, "ProveCore", pb_bytes(prove_payload(12, 8, 1, 100, 16, cbor, elf, backend=1)))
    assert st == 200
    st, ct, body = call(port, "ProveCore", pb_bytes(prove_payload(10, 16, 1, 100000, 8, cbor, elf)))
    assert st == 400 and json.loads(body)["code"] == "invalid_argument"           # the library's own argument check comes back as a Twirp error


@pytest.mark.gpu
def test_prove_core_input_commitment_guest(server):
    """shards = 0 in the payload: the SHA-256-of-the-input guest; output = the digest, one chip proof in the blob"""
    import hashlib
    from zktls_amd._lib import Params
    from zktls_amd.device import verify_sha256
    cbor = bytes((3 * i + 1) & 0xff for i in range(5000))
    st, ct, body = call(server, "ProveCore", pb_bytes(prove_payload(0, 0, 0, 16, 5, cbor, b"\x7fELFguest")))
    assert st == 200, body
    res = pb_field1(body)
    on = struct.unpack_from("<I", res)[0]
    output, blob = res[4:4 + on], res[4 + on:]
    assert output == hashlib.sha256(cbor).digest()
    magic, version, flags, count = struct.unpack_from("<4I", blob)
    assert (magic, version, flags, count) == (0x42544B5A, 2, 2, 2)              # "ZKTB", flagged INPUT_SHA256: the proof, then the input's length
    ln = struct.unpack_from("<I", blob, 16)[0]
    assert 20 + ln + 12 == len(blob) and struct.unpack_from("<IQ", blob, 20 + ln) == (8, len(cbor))
    assert verify_sha256(np.frombuffer(blob[20:20 + ln], dtype=np.uint8), output, Params(1, 16, 5), len(cbor)) == (0, 0)


def prove_payload_v2(shards, queries, pow_bits, cbor, elf, flags, backend=0, device=0, log_n=0, width=0):
    return (b"ZKMG" + struct.pack("<IiIIiiIiI", 2, log_n, width, shards, queries, pow_bits, backend, device, flags)
            + struct.pack("<I", len(cbor)) + cbor + struct.pack("<I", len(elf)) + elf)


def test_prove_core_v2_argument_checks(server):
    for payload in (prove_payload_v2(1, 10, 4, b"in", b"elf", 1),            # KEYED with shards > 0
                    prove_payload_v2(0, 10, 4, b"in", b"elf", 1, backend=1),   # KEYED in the RISC Zero shape
                    prove_payload_v2(0, 10, 4, b"in", b"elf", 3),              # KEYED and COMPRESS together
                    prove_payload_v2(2, 10, 4, b"in", b"elf", 2, backend=1, log_n=8, width=8),   # COMPRESS in the RISC Zero shape
                    prove_payload_v2(0, 10, 4, b"in", b"elf", 4)):             # unknown flag
        st, ct, body = call(server, "ProveCore", pb_bytes(payload))
        assert st == 400 and json.loads(body)["code"] == "invalid_argument"


@pytest.mark.gpu
def test_prove_core_keyed_commitment_guest(server):
    """payload version 2 with KEYED: setup -> prove -> verify in the server; the response carries the 64-byte vk a CPU-only client needs"""
    import ctypes as C
    import hashlib
    L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zktls_amd", "libzktls_guest_prover.so"))
    L.zktls_verify_commitment_blob.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int)]
    cbor = bytes((5 * i + 2) & 0xff for i in range(7000))
    st, ct, body = call(server, "ProveCore", pb_bytes(prove_payload_v2(0, 16, 5, cbor, b"\x7fELFguest", 1)))
    assert st == 200, body
    res = pb_field1(body)
    on = struct.unpack_from("<I", res)[0]
    output = res[4:4 + on]
    vn = struct.unpack_from("<I", res, 4 + on)[0]
    vk, blob = res[8 + on:8 + on + vn], res[8 + on + vn:]
    assert output == hashlib.sha256(cbor).digest() and vn == 64
    assert struct.unpack_from("<4I", blob) == (0x42544B5A, 2, 6, 2)                # INPUT_SHA256 | KEYED: the proof, then the input's length
    reason = C.c_int(0)
    assert L.zktls_verify_commitment_blob(blob, len(blob), output, vk, 64, 16, 5, C.byref(reason)) == 0
    assert L.zktls_verify_commitment_blob(blob, len(blob), hashlib.sha256(b"x").digest(), vk, 64, 16, 5, C.byref(reason)) != 0
    # version 2 without the flag: the plain chip proof, an empty vk
    st, ct, body = call(server, "ProveCore", pb_bytes(prove_payload_v2(0, 16, 5, cbor, b"\x7fELFguest", 0)))
    res = pb_field1(body)
    on = struct.unpack_from("<I", res)[0]
    assert st == 200 and struct.unpack_from("<I", res, 4 + on)[0] == 0 and struct.unpack_from("<4I", res, 8 + on) == (0x42544B5A, 2, 2, 2)


@pytest.mark.gpu
def test_compress_is_a_step_of_its_own_and_a_flag_of_prove_core(server):
    """sp1-cuda's prove_core -> compress pair (behind sp1.rs:116): ProveCore returns the shard proofs, Compress takes that blob and returns ONE proof
    that verifies them in-circuit -- the same bytes ProveCore returns with the COMPRESS flag; with two shard proofs per join the five shards take three
    joins and ONE proof above them (the tree).  A verifier with no device checks either blob from (plan, input, ELF, key derived on the host)."""
    import ctypes as C
    L = C.CDLL(os.path.join(ROOT, "zktls_amd", "libzktls_guest_prover.so"))

    class Plan(C.Structure):
        _fields_ = [("log_n", C.c_int32), ("width", C.c_uint32), ("shards", C.c_uint32), ("num_queries", C.c_int32), ("pow_bits", C.c_int32)]
    L.zktls_verify_compressed_blob.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Plan), C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    L.zktls_compress_key_host.argtypes = [C.POINTER(Plan), C.POINTER(C.c_uint32), C.c_char_p, C.c_size_t]
    L.zktls_set_compress_join_size.argtypes = [C.c_uint32]
    L.zktls_batch_flags.argtypes = [C.c_char_p, C.c_size_t]
    cbor, elf = b"\xa1transcript" * 9, b"\x7fELFguest"
    shape = (8, 8, 5, 4, 2)
    st, ct, body = call(server, "ProveCore", pb_bytes(prove_payload(*shape, cbor, elf)))
    assert st == 200, body
    res = pb_field1(body)
    on = struct.unpack_from("<I", res)[0]
    core = res[4 + on:]
    plan, key, err, reason = Plan(*shape), (C.c_uint32 * 8)(), C.create_string_buffer(512), C.c_int(0)
    for per_join, flags in ((0, 1 | 16), (2, 1 | 16 | 32)):
        st, ct, body = call(server, "Compress", pb_bytes(compress_payload(*shape, cbor, elf, core, per_join=per_join)))
        assert st == 200, body
        blob = pb_field1(body)
        assert L.zktls_batch_flags(blob, len(blob)) == flags                                  # (at this toy size the one proof is larger than the five it verifies)
        L.zktls_set_compress_join_size(per_join)
        try:
            assert L.zktls_compress_key_host(C.byref(plan), key, err, 512) == 0, err.value
            assert L.zktls_verify_compressed_blob(blob, len(blob), C.byref(plan), cbor, len(cbor), elf, len(elf), key, C.byref(reason)) == 0
            assert L.zktls_verify_compressed_blob(blob, len(blob), C.byref(plan), cbor + b"!", len(cbor) + 1, elf, len(elf), key, C.byref(reason)) == -2
        finally:
            L.zktls_set_compress_join_size(0)
        if per_join == 0:
            st, ct, body = call(server, "ProveCore", pb_bytes(prove_payload_v2(shape[2], shape[3], shape[4], cbor, elf, 2, log_n=shape[0], width=shape[1])))
            assert st == 200, body
            res2 = pb_field1(body)
            on2 = struct.unpack_from("<I", res2)[0]
            assert struct.unpack_from("<I", res2, 4 + on2)[0] == 0 and res2[8 + on2:] == blob          # (version 2: an empty vk, then the blob)
    # another request's shard proofs do not compress under this one: the machine's tables are the verification
    st, ct, body = call(server, "Compress", pb_bytes(compress_payload(*shape, cbor + b"!", elf, core)))
    assert st == 400 and json.loads(body)["code"] == "invalid_argument", body
    # ... nor does a blob with a shard missing
    st, ct, body = call(server, "Compress", pb_bytes(compress_payload(8, 8, 4, 4, 2, cbor, elf, core)))
    assert st == 400 and json.loads(body)["code"] == "invalid_argument"
    st, _, _ = call(server, "Ready", b"")
    assert st == 200
