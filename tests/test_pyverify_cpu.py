"""The committed golden proofs (tests/golden/proofs/*.bin, one per proof version; produced by the oracle, see the provenance
block of oracle_kat.json) under THREE verifiers that share no code:
  * tests/pyverify.py  -- pure Python, written from the protocol description in DESIGN.md sections 3 and 6;
  * the product's host verifier (zkhip_verify_shard, C++, Montgomery arithmetic) -- runs on the CPU, no device needed;
  * the oracle's verifier (C, canonical arithmetic).
This does not pin parity with upstream SP1 / RISC Zero (nothing can, offline: SURVEY.md section 4); it shrinks the room for
a protocol misreading shared by the oracle and the product, which were written by the same hand."""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

import pyverify
from zktls_amd._lib import Params
from zktls_amd.device import verify_shard

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "oracle_kat.json")))
GOLDEN = KAT["golden_proof_files"]


def load(name):
    return open(os.path.join(HERE, "golden", "proofs", name + ".bin"), "rb").read()


def test_fixtures_carry_their_provenance():
    prov = KAT["provenance"]
    assert len(prov["oracle_commit"]) == 40 and prov["generator"] == "tests/golden/make_golden.py"
    assert "parity unpinned" in prov["note"]
    for name, g in GOLDEN.items():
        b = load(name)
        assert len(b) == g["bytes"] and hashlib.sha256(b).hexdigest() == g["sha256"]


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_three_independent_verifiers_accept_the_golden_proof(name, oracle):
    g = GOLDEN[name]
    b = load(name)
    assert pyverify.verify(b, g["log_n"], g["width"], g["public"], *g["shape"]) is True
    arr = np.frombuffer(b, dtype=np.uint8)
    assert verify_shard(arr, g["log_n"], g["width"], g["public"], Params(*g["shape"])) == (0, 0)
    assert oracle.verify_shard(arr, g["log_n"], g["width"], g["public"], oracle.default_params(*g["shape"])) == 0


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_verifiers_agree_on_corrupted_proofs(name, oracle):
    """every single-word corruption tried is rejected by all three (the failing check may differ, the verdict may not)"""
    g = GOLDEN[name]
    b = load(name)
    n_words = len(b) // 4
    rng = np.random.default_rng(len(b))
    offsets = sorted(set([8, 12, 20, n_words // 3, n_words // 2, n_words - 2] + rng.integers(8, n_words, 6).tolist()))
    for off in offsets:
        bad = bytearray(b)
        v = struct.unpack_from("<I", bad, 4 * off)[0]
        struct.pack_into("<I", bad, 4 * off, (v + 1) % pyverify.P)
        with pytest.raises(pyverify.Reject):
            pyverify.verify(bytes(bad), g["log_n"], g["width"], g["public"], *g["shape"])
        arr = np.frombuffer(bytes(bad), dtype=np.uint8)
        assert verify_shard(arr, g["log_n"], g["width"], g["public"], Params(*g["shape"]))[0] == -6
        assert oracle.verify_shard(arr, g["log_n"], g["width"], g["public"], oracle.default_params(*g["shape"])) != 0
    # wrong public values / wrong parameters
    with pytest.raises(pyverify.Reject):
        pyverify.verify(b, g["log_n"], g["width"], [1, 2, 4], *g["shape"])
    shape = list(g["shape"])
    shape[1] += 1
    with pytest.raises(pyverify.Reject):
        pyverify.verify(b, g["log_n"], g["width"], g["public"], *shape)


def test_python_transcript_matches_the_oracle_challenger(oracle):
    """the duplex sponge of pyverify.Transcript against the oracle's challenger KAT (oracle_kat.json)"""
    ts = pyverify.Transcript()
    ts.observe_many(range(1, 12))
    assert [ts.sample() for _ in range(10)] == KAT["challenger_samples"]
    assert ts.sample_bits(12) == KAT["challenger_bits"]
