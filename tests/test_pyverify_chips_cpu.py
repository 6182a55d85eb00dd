"""Multi-chip proofs (versions 4, 5, 6, 9, 10, 11) under THREE verifiers that share no code: tests/pyverify_chips.py (pure Python,
from DESIGN.md section 6), the product's host verifier (C++, Montgomery arithmetic) and the oracle's (C, canonical arithmetic)."""
import struct

import numpy as np
import pytest

import airs
import machines as M
import pyverify_chips as V
from pyverify import Reject
from zktls_amd._lib import Params
from zktls_amd.device import verify_chips, verify_chips_air, verify_machine, verify_machine_keyed

SEED = 0x5A4B544C53
P = 2013265921


def cases(O):
    fib = airs.fibonacci_program()
    ft, fpub = airs.fibonacci_trace(6, 3, 5)
    cnt = airs.counter_program(8).copy()
    cnt[4] = 3
    mt, mp, mtab, mpub = M.range_machine(5, 6)
    kt, kpre, kp, ktab, kpub = M.byte_machine(6, 3)
    rt, rpre, rp, rtab, rpub = M.random_keyed_machine(303)
    return {
        "v11": dict(traces=kt, pre=kpre, pub=kpub, programs=kp, tables=ktab),
        "v11r": dict(traces=rt, pre=rpre, pub=rpub, programs=rp, tables=rtab),
        "v4": dict(traces=[O.gen_trace(SEED, 1, 7, 8), O.gen_trace(SEED, 2, 6, 12), O.gen_trace(SEED, 3, 6, 4)], pub=[1, 2]),
        "v5": dict(traces=[O.gen_trace_logup(SEED, 1, 7, 16, 2), O.gen_trace(SEED, 2, 6, 8)], pub=[5], pairs=[2, 0]),
        "v6": dict(traces=[O.gen_trace_logup_cross(SEED, 0, 1, 6, 16, 8, 1), O.gen_trace_logup_cross(SEED, 1, 0, 6, 8, 16, 1), O.gen_trace(SEED, 2, 5, 4)],
                   pub=[5], pairs=[1, 1, 0], partners=[1, 0, -1]),
        "v9": dict(traces=[airs.counter_trace(7, 8, 3, 5)[0], O.gen_trace(SEED, 4, 7, 4), ft], pub=fpub, programs=[cnt, None, fib]),
        "v10": dict(traces=mt, pub=mpub, programs=mp, tables=mtab),
    }


def prove(O, c, oprm):
    if "pre" in c:
        c["root"] = O.machine_setup(c["pre"], [t.shape[0].bit_length() - 1 for t in c["traces"]], oprm)
        c["pre_widths"] = [0 if p is None else p.shape[1] for p in c["pre"]]
        return O.prove_machine_keyed(c["traces"], c["pre"], c["programs"], c["tables"], c["pub"], oprm)
    if "tables" in c:
        return O.prove_machine(c["traces"], c["programs"], c["tables"], c["pub"], oprm)
    if "programs" in c:
        return O.prove_chips_air(c["traces"], c["programs"], c["pub"], oprm)
    return O.prove_chips(c["traces"], c["pub"], oprm, c.get("pairs"), c.get("partners"))


def product_verdict(c, proof, lns, ws, prm):
    if "pre" in c:
        return verify_machine_keyed(proof, lns, ws, c["pre_widths"], c["root"], c["programs"], c["tables"], c["pub"], prm)[0]
    if "tables" in c:
        return verify_machine(proof, lns, ws, c["programs"], c["tables"], c["pub"], prm)[0]
    if "programs" in c:
        return verify_chips_air(proof, lns, ws, c["programs"], c["pub"], prm)[0]
    return verify_chips(proof, lns, ws, c["pub"], prm, c.get("pairs"), c.get("partners"))[0]


@pytest.mark.parametrize("name", ["v4", "v5", "v6", "v9", "v10", "v11", "v11r"])
@pytest.mark.parametrize("shape", [(1, 4, 3), (2, 3, 0)])
def test_three_verifiers_agree(oracle, name, shape):
    O = oracle
    c = cases(O)[name]
    lns, ws = [t.shape[0].bit_length() - 1 for t in c["traces"]], [t.shape[1] for t in c["traces"]]
    proof = prove(O, c, O.default_params(*shape))
    kw = {k: c[k] for k in ("pairs", "partners", "programs", "tables", "pre_widths") if k in c}
    if "root" in c:
        kw["pre_root"] = c["root"].tolist()
    assert np.frombuffer(proof.tobytes(), dtype=np.uint32)[1] == int(name[1:].rstrip("r"))
    assert V.verify(proof.tobytes(), lns, ws, c["pub"], *shape, **kw) is True
    assert product_verdict(c, proof, lns, ws, Params(*shape)) == 0
    n_words = proof.size // 4
    rng = np.random.default_rng(n_words)
    for off in sorted(set([9, 20, 40, n_words // 2, n_words - 2] + rng.integers(8, n_words, 5).tolist())):
        bad = bytearray(proof.tobytes())
        v = struct.unpack_from("<I", bad, 4 * off)[0]
        struct.pack_into("<I", bad, 4 * off, (v + 1) % P)
        with pytest.raises(Reject):
            V.verify(bytes(bad), lns, ws, c["pub"], *shape, **kw)
        assert product_verdict(c, np.frombuffer(bytes(bad), dtype=np.uint8), lns, ws, Params(*shape)) == -6, off
    other_pub = list(c["pub"])
    other_pub[0] = (other_pub[0] + 1) % P
    with pytest.raises(Reject):
        V.verify(proof.tobytes(), lns, ws, other_pub, *shape, **kw)
