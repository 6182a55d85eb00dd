import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    oracle_lib.set_threads(min(8, os.cpu_count() or 1))
    return oracle_lib


@pytest.fixture(scope="session")
def oracle_shard_proof(oracle):
    """full-size oracle proofs, each made ONCE per session (a 2^20-row oracle proof is ~10 s on the box's cores; several tests compare HIP bytes
    with the same one): f(seed, shard, log_n, width, public values, "sp1" | "r0") -> proof bytes of the oracle's prover on the oracle's trace"""
    made = {}

    def get(seed, shard, log_n, width, public_values, shape="sp1"):
        key = (int(seed), int(shard), int(log_n), int(width), tuple(int(v) for v in public_values), shape)
        if key not in made:
            prev = min(8, os.cpu_count() or 1)
            oracle.set_threads(min(os.cpu_count() or 1, 96))
            try:
                prm = oracle.default_params(1, 100, 16) if shape == "sp1" else oracle.segment_params()
                made[key] = oracle.prove_shard(oracle.gen_trace(seed, shard, log_n, width), list(public_values), prm).tobytes()
            finally:
                oracle.set_threads(prev)
        return made[key]
    return get


@pytest.fixture(scope="session")
def ctx():
    """One HIP context for the GPU tests.  Fails loudly (no skip, no fallback) when the
    HIP library is missing or no gfx950 device is usable."""
    from zktls_amd.device import Context
    c = Context(0)
    yield c
    c.close()
