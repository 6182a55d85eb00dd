"""Poseidon2 parameter tables from a file (csrc/params.cpp): dropping in another table needs no
rebuild.  CPU part: the host side of the library (request digest, transcript, verifier) follows the loaded set, checked
against the pure-Python permutation of tests/pyref.py run with the same file.  GPU part: the device permutation, the Merkle
kernels and a whole proof follow it too."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import pyref
from zktls_amd import _lib
from zktls_amd._lib import Params
from zktls_amd.device import verify_shard

HERE = os.path.dirname(os.path.abspath(__file__))
P = 2013265921


def _variant(tmp_path, width):
    """the committed parameter file of that width with every constant changed (still canonical)"""
    src = "poseidon2_params.json" if width == 16 else "poseidon2_24_params.json"
    d = json.load(open(os.path.join(HERE, "golden", src)))
    d["name"] = "test-variant-%d" % width
    d["external_rc"] = [[(v * 7 + 11 * i + r) % P for i, v in enumerate(row)] for r, row in enumerate(d["external_rc"])]
    d["internal_rc"] = [(v * 5 + 3) % P for v in d["internal_rc"]]
    d["internal_diag"] = [(v + 12345 + i) % P for i, v in enumerate(d["internal_diag"])]
    path = os.path.join(str(tmp_path), "variant%d.json" % width)
    json.dump(d, open(path, "w"))
    return path, d


def _skip_if_contexts_alive(L):
    # the set can only change while no context exists; a GPU session running the whole suite holds the shared fixture's context
    if L.zkhip_reset_poseidon2_params() != 0 and b"contexts exist" in L.zkhip_last_error():
        pytest.skip("a context is alive in this process")


def _digest(L, cbor, elf):
    out = (C.c_uint32 * 8)()
    assert L.zkhip_request_digest(cbor, len(cbor), elf, len(elf), out) == 0
    return list(out)


def _py_digest(params, cbor, elf):
    old = pyref.PARAMS
    pyref.PARAMS = params
    me, mi = pyref.ME, pyref.MI
    try:
        pyref.MI = [[(1 + (params["internal_diag"][i] if i == j else 0)) % P for j in range(16)] for i in range(16)]
        words = [0x5A4B54]
        for blob in (cbor, elf):
            words += [len(blob) & 0xFFFFFF, (len(blob) >> 24) & 0xFFFFFF]
            words += [int.from_bytes(blob[i:i + 3], "little") for i in range(0, len(blob), 3)]
        st = [0] * 16
        for i in range(0, len(words), 8):
            chunk = words[i:i + 8]
            st[:len(chunk)] = chunk
            st = pyref.poseidon2(st)
        return st[:8]
    finally:
        pyref.PARAMS, pyref.ME, pyref.MI = old, me, mi


def test_loader_changes_the_host_side_and_reset_restores_it(tmp_path):
    L = _lib.load()
    _skip_if_contexts_alive(L)
    cbor, elf = b"\xa2input" * 5, b"\x7fELF...."
    builtin = _digest(L, cbor, elf)
    assert L.zkhip_poseidon2_params_name(16) == b"zktls-amd/p2-bb16-v1"
    g = json.load(open(os.path.join(HERE, "golden", "oracle_kat.json")))["golden_proof_files"]["v1_6x8"]
    proof = np.frombuffer(open(os.path.join(HERE, "golden", "proofs", "v1_6x8.bin"), "rb").read(), dtype=np.uint8)
    assert verify_shard(proof, g["log_n"], g["width"], g["public"], Params(*g["shape"])) == (0, 0)
    assert _air_verdict(L) == 0
    try:
        # the committed file IS the built-in set: loading it changes nothing
        assert L.zkhip_load_poseidon2_params(os.path.join(HERE, "golden", "poseidon2_params.json").encode()) == 0
        assert _digest(L, cbor, elf) == builtin
        path, d = _variant(tmp_path, 16)
        assert L.zkhip_load_poseidon2_params(path.encode()) == 0
        assert L.zkhip_poseidon2_params_name(16) == b"test-variant-16" and L.zkhip_poseidon2_params_name(24) == b"zktls-amd/p2-bb24-v1"
        got = _digest(L, cbor, elf)
        assert got != builtin and got == _py_digest(d, cbor, elf)          # follows the file, value pinned by the Python permutation
        # a proof made under the built-in set no longer verifies: transcript and Merkle hashes changed
        assert verify_shard(proof, g["log_n"], g["width"], g["public"], Params(*g["shape"]))[0] == -6
        # the program-digest cache of the prover / verifier follows the table set (it is keyed by the set's generation)
        assert _air_verdict(L) == -6
    finally:
        assert L.zkhip_reset_poseidon2_params() == 0
    assert _digest(L, cbor, elf) == builtin
    assert verify_shard(proof, g["log_n"], g["width"], g["public"], Params(*g["shape"])) == (0, 0)
    assert _air_verdict(L) == 0


def _air_verdict(L):
    """a constraint-program proof made by the oracle under the BUILT-IN set, through the product's verifier (whose program digest
    comes from its cache): 0 = accepted"""
    import airs
    import oracle_lib as O
    from zktls_amd.device import verify_shard_air
    prog = airs.fibonacci_program()
    if not hasattr(_air_verdict, "proof"):
        t, pub = airs.fibonacci_trace(6, 3, 5)
        _air_verdict.pub = pub
        _air_verdict.proof = O.prove_shard_air(prog, t, pub, O.default_params(1, 4, 4))
    return verify_shard_air(prog, _air_verdict.proof, 6, 4, _air_verdict.pub, Params(1, 4, 4))[0]


def test_loader_rejects_bad_files(tmp_path):
    L = _lib.load()
    _skip_if_contexts_alive(L)
    base = json.load(open(os.path.join(HERE, "golden", "poseidon2_params.json")))

    def load(d):
        path = os.path.join(str(tmp_path), "bad.json")
        json.dump(d, open(path, "w"))
        return L.zkhip_load_poseidon2_params(path.encode())
    assert L.zkhip_load_poseidon2_params(b"/nonexistent/params.json") == -1
    assert L.zkhip_load_poseidon2_params(None) == -1
    for mutate in (lambda d: d.update(width=12), lambda d: d.update(internal_rc=d["internal_rc"][:-1]),
                   lambda d: d.update(internal_diag=d["internal_diag"] + [1]), lambda d: d["external_rc"][3].__setitem__(5, P),
                   lambda d: d.update(rounds_p=14), lambda d: d.update(p=P + 2), lambda d: d.pop("external_rc"),
                   lambda d: d["internal_rc"].__setitem__(0, -1)):
        d = json.loads(json.dumps(base))
        mutate(d)
        assert load(d) == -1, d.keys()
    assert load(base) == 0                      # and a good one still loads afterwards
    assert L.zkhip_reset_poseidon2_params() == 0


def _device_follows_the_loaded_tables(tmp_path):
    from zktls_amd.device import Context
    L = _lib.load()
    L.zkhip_release_cached_contexts()
    states = np.arange(64, dtype=np.uint32).reshape(4, 16) * 1000003 % P
    c0 = Context(0)
    builtin_perm = c0.from_numpy(states)
    c0.poseidon2_permute(builtin_perm)
    builtin_perm = builtin_perm.download().reshape(4, 16)
    # the parameter set cannot change under a live context
    path16, d16 = _variant(tmp_path, 16)
    assert L.zkhip_load_poseidon2_params(path16.encode()) == -1 and b"contexts exist" in L.zkhip_last_error()
    trace_seed, log_n, width = 0x5A4B544C53, 9, 16
    proof_builtin = c0.prove_shard(c0.gen_trace(trace_seed, 1, log_n, width), log_n, width, [1], Params(1, 12, 6))
    c0.close()
    path24, d24 = _variant(tmp_path, 24)
    try:
        assert L.zkhip_load_poseidon2_params(path16.encode()) == 0
        assert L.zkhip_load_poseidon2_params(path24.encode()) == 0
        c1 = Context(0)
        buf = c1.from_numpy(states)
        c1.poseidon2_permute(buf)
        got = buf.download().reshape(4, 16)
        old = (pyref.PARAMS, pyref.ME, pyref.MI)
        pyref.PARAMS = d16
        pyref.MI = [[(1 + (d16["internal_diag"][i] if i == j else 0)) % P for j in range(16)] for i in range(16)]
        try:
            want = [pyref.poseidon2(list(map(int, s))) for s in states]
        finally:
            pyref.PARAMS, pyref.ME, pyref.MI = old
        assert got.tolist() == want and got.tolist() != builtin_perm.tolist()
        # a whole proof under the loaded tables (both hash widths): accepted by the host verifier under the same tables, different bytes
        trace = c1.gen_trace(trace_seed, 1, log_n, width)
        p16 = c1.prove_shard(trace, log_n, width, [1], Params(1, 12, 6))
        assert verify_shard(p16, log_n, width, [1], Params(1, 12, 6)) == (0, 0)
        assert p16.tobytes() != proof_builtin.tobytes()
        assert verify_shard(proof_builtin, log_n, width, [1], Params(1, 12, 6))[0] == -6
        shape24 = (2, 10, 0, 0, 4, 1, 24)
        p24 = c1.prove_shard(trace, log_n, width, [1], Params(*shape24))
        assert verify_shard(p24, log_n, width, [1], Params(*shape24)) == (0, 0)
        c1.close()
    finally:
        L.zkhip_release_cached_contexts()
        assert L.zkhip_reset_poseidon2_params() == 0
    c2 = Context(0)
    again = c2.prove_shard(c2.gen_trace(trace_seed, 1, log_n, width), log_n, width, [1], Params(1, 12, 6))
    assert again.tobytes() == proof_builtin.tobytes()           # back on the built-in set: the same bytes as before
    c2.close()


@pytest.mark.gpu
def test_device_follows_the_loaded_tables(tmp_path):
    """in a child process: the parameter set can only change while no context exists, and this pytest session holds one"""
    import subprocess
    import sys
    code = ("import sys; sys.path[:0] = [%r, %r]; import test_params_loader as t; t._device_follows_the_loaded_tables(%r); print('child ok')"
            % (os.path.dirname(HERE), HERE, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout + r.stderr
