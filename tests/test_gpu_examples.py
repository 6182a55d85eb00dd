"""The plain-C++ users of the C ABI (examples/, built by __graft_entry__.build()) run on the GPU box."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run(*argv):
    exe = os.path.join(ROOT, "examples", argv[0])
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    p = subprocess.run([exe] + [str(a) for a in argv[1:]], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


def test_compress_shards_example():
    out = run("compress_shards", 6, 12, 16)
    assert re.search(r"6 shards of 2\^12 x 16: shard proofs .* ONE proof that verifies them all .* verified on the host .* from \(shape, 6 public values, key\) alone", out), out


def test_compress_tree_example():
    out = run("compress_tree", 6, 3, 10, 16)
    assert re.search(r"6 shards of 2\^10 x 16: .* 2 joins of 3: .* ONE proof over the joins .* verified on the host .* keys derived on the host", out), out


def test_prove_shard_example():
    out = run("prove_shard", 12, 16, 3, "batch")
    assert "mean" in out and "proof bytes" in out, out
