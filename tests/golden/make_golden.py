#!/usr/bin/env python3
"""Regenerate tests/golden/oracle_kat.json from the CPU oracle.

The reference (the3cloud/zktls) holds NO golden vector for the prover path (SURVEY.md
section 4), so these fixtures are produced by this repo's own oracle ("parity
unpinned"): they freeze the oracle against regressions and give the GPU tests fixed
expected values that travel to the GPU box.  Independent pinning of the primitives comes
from tests/pyref.py (pure-Python first-principles definitions), checked in
tests/test_oracle.py.  Run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

SEED = 0x5A4B544C53


def main():
    O.build()
    out = {}
    out["poseidon2_iota"] = O.poseidon2(np.arange(16, dtype=np.uint32)).tolist()
    out["poseidon2_zero"] = O.poseidon2(np.zeros(16, dtype=np.uint32)).tolist()
    out["sponge_1_to_20"] = O.sponge_hash(np.arange(1, 21, dtype=np.uint32)).tolist()
    out["compress"] = O.compress(np.arange(8, dtype=np.uint32), np.arange(8, 16, dtype=np.uint32)).tolist()
    out["ntt_1_to_8"] = O.ntt(np.arange(1, 9, dtype=np.uint32).reshape(8, 1)).ravel().tolist()
    m = O.fill_uniform(SEED, 6, 4)
    out["fill_uniform_6x4_first8"] = m.ravel()[:8].tolist()
    lde = O.coset_lde(m, 1, 31)
    out["lde_6x4_sha256"] = hashlib.sha256(lde.tobytes()).hexdigest()
    out["lde_6x4_root"] = O.merkle_tree([lde])[-1].tolist()
    # bigger commit used by the GPU tests (two-pass NTT path: log_n = 12)
    m12 = O.fill_uniform(SEED + 1, 12, 24)
    lde12 = O.coset_lde(m12, 1, 31)
    out["lde_12x24_sha256"] = hashlib.sha256(lde12.tobytes()).hexdigest()
    out["lde_12x24_root"] = O.merkle_tree([lde12])[-1].tolist()
    ch = O.OracleChallenger()
    ch.observe(np.arange(1, 12, dtype=np.uint32))
    out["challenger_samples"] = [int(ch.sample()) for _ in range(10)]
    out["challenger_bits"] = int(ch.sample_bits(12))
    ch2 = O.OracleChallenger()
    ch2.observe(np.arange(5, dtype=np.uint32))
    out["challenger_grind8"] = int(ch2.grind(8))
    proofs = {}
    for name, (log_n, w, q, pw) in {"p6x8": (6, 8, 10, 8), "p10x16": (10, 16, 100, 16), "p12x32": (12, 32, 100, 16)}.items():
        t = O.gen_trace(SEED, 3, log_n, w)
        prm = O.default_params(1, q, pw)
        pf = O.prove_shard(t, [7, 8, 9], prm)
        assert O.verify_shard(pf, log_n, w, [7, 8, 9], prm) == 0
        d = O.prove_debug()
        proofs[name] = {
            "log_n": log_n, "width": w, "num_queries": q, "pow_bits": pw, "shard": 3, "public": [7, 8, 9],
            "bytes": int(pf.size), "sha256": hashlib.sha256(pf.tobytes()).hexdigest(),
            "trace_root": d["trace_root"].tolist(), "quotient_root": d["quotient_root"].tolist(),
            "alpha": d["alpha"].tolist(), "zeta": d["zeta"].tolist(), "fri_alpha": d["fri_alpha"].tolist(),
            "pow_witness": d["pow_witness"],
        }
    out["proofs"] = proofs
    # other proof-system shapes: (log_blowup, queries, pow_bits, logup_pairs, log_fold, log_final, hash_width);
    # "r0_*" are RISC Zero's parameters (blowup 4, fold 16, Poseidon2 width 24, no PoW) at small sizes
    shapes = {}
    for name, (log_n, w, shape) in {"r0_10x16": (10, 16, (2, 50, 0, 0, 4, 2, 24)), "r0_12x32": (12, 32, (2, 50, 0, 0, 4, 8, 24)),
                                    "r0_lookup_10x32": (10, 32, (2, 20, 0, 2, 4, 6, 24)), "blowup4_9x8": (9, 8, (2, 10, 8, 0, 1, 0, 16)),
                                    "fold8_9x8": (9, 8, (1, 10, 8, 0, 3, 0, 16)),
                                    # 8th field = code_width: RISC Zero's code / data(/ accum) / check group order (proof version 8)
                                    "r0_groups_12x32": (12, 32, (2, 50, 0, 0, 4, 8, 24, 8)),
                                    "r0_groups_lookup_10x32": (10, 32, (2, 20, 0, 2, 4, 6, 24, 12))}.items():
        pairs = shape[3]
        t = O.gen_trace_logup(SEED, 5, log_n, w, pairs) if pairs else O.gen_trace(SEED, 5, log_n, w)
        prm = O.default_params(*shape)
        pf = O.prove_shard(t, [1, 2, 3], prm)
        assert O.verify_shard(pf, log_n, w, [1, 2, 3], prm) == 0
        shapes[name] = {"log_n": log_n, "width": w, "shape": list(shape), "shard": 5, "public": [1, 2, 3],
                        "bytes": int(pf.size), "sha256": hashlib.sha256(pf.tobytes()).hexdigest()}
    out["shape_proofs"] = shapes
    # shards of several chips with different heights (SP1's shard shape): [(log_n, width), ...] tallest first
    chipsets = {}
    for name, (chips, prm) in {"two_chips": ([(10, 16), (8, 8)], (1, 10, 4)),
                               "five_chips": ([(10, 16), (10, 8), (7, 12), (7, 4), (5, 8)], (1, 20, 8)),
                               "blowup4_chips": ([(9, 8), (8, 8), (7, 8), (6, 8), (5, 8)], (2, 10, 0))}.items():
        traces = [O.gen_trace(SEED, i, ln, w) for i, (ln, w) in enumerate(chips)]
        params = O.default_params(*prm)
        pf = O.prove_chips(traces, [3, 4], params)
        assert O.verify_chips(pf, [c[0] for c in chips], [c[1] for c in chips], [3, 4], params) == 0
        chipsets[name] = {"chips": [list(c) for c in chips], "params": list(prm), "public": [3, 4],
                          "bytes": int(pf.size), "sha256": hashlib.sha256(pf.tobytes()).hexdigest()}
    out["chip_proofs"] = chipsets
    # chips with lookups: (log_n, width, pairs, partner); partner >= 0 = the two chips look each other up
    lk = {}
    for name, (chips, prm) in {"local_lookups": ([(10, 16, 2, -1), (8, 8, 0, -1), (7, 24, 3, -1)], (1, 12, 4)),
                               "cross_lookups": ([(9, 16, 2, 1), (9, 24, 2, 0), (7, 8, 1, -1), (6, 4, 0, -1)], (1, 12, 4))}.items():
        traces = []
        for i, (ln, w, pr, pa) in enumerate(chips):
            traces.append(O.gen_trace_logup_cross(SEED, i, pa, ln, w, chips[pa][1], pr) if pa >= 0 else
                          (O.gen_trace_logup(SEED, i, ln, w, pr) if pr else O.gen_trace(SEED, i, ln, w)))
        params = O.default_params(*prm)
        prs, pas = [c[2] for c in chips], [c[3] for c in chips]
        cross = any(p >= 0 for p in pas)
        pf = O.prove_chips(traces, [3, 4], params, prs, pas if cross else None)
        assert O.verify_chips(pf, [c[0] for c in chips], [c[1] for c in chips], [3, 4], params, prs, pas if cross else None) == 0
        lk[name] = {"chips": [list(c) for c in chips], "params": list(prm), "public": [3, 4],
                    "bytes": int(pf.size), "sha256": hashlib.sha256(pf.tobytes()).hexdigest()}
    out["chip_lookup_proofs"] = lk
    # machines: tables with their own constraint programs that look each other up (tests/machines.py); proof versions 9 and 10
    import machines
    mach = {}
    for name, (args, prm) in {"range_5_6": (("range", 5, 6, 1), (1, 8, 4)), "range_6_8_blowup4": (("range", 6, 8, 2), (2, 6, 0)),
                              "random_3": (("random", 3), (1, 5, 3)), "random_7": (("random", 7), (3, 4, 2))}.items():
        tr, pg, tb, pub = machines.range_machine(*args[1:]) if args[0] == "range" else machines.random_machine(args[1])
        params = O.default_params(*prm)
        pf = O.prove_machine(tr, pg, tb, pub, params)
        lns, ws = [t.shape[0].bit_length() - 1 for t in tr], [t.shape[1] for t in tr]
        assert O.verify_machine(pf, lns, ws, pg, tb, pub, params) == 0
        mach[name] = {"machine": list(args), "params": list(prm), "bytes": int(pf.size), "sha256": hashlib.sha256(pf.tobytes()).hexdigest()}
    out["machine_proofs"] = mach
    # keyed machines: tables with preprocessed columns, committed once by setup (proof version 11); the key's root is part of the fixture
    keyed = {}
    for name, (args, prm) in {"byte_6_3": (("byte", 6, 3, 1), (1, 8, 4)), "byte_8_4_blowup8": (("byte", 8, 4, 2), (3, 4, 0)),
                              "random_keyed_301": (("random_keyed", 301), (1, 5, 3)), "random_keyed_305": (("random_keyed", 305), (2, 4, 2))}.items():
        tr, pre, pg, tb, pub = machines.byte_machine(*args[1:]) if args[0] == "byte" else machines.random_keyed_machine(args[1])
        params = O.default_params(*prm)
        lns, ws = [t.shape[0].bit_length() - 1 for t in tr], [t.shape[1] for t in tr]
        pws = [0 if p is None else p.shape[1] for p in pre]
        root = O.machine_setup(pre, lns, params)
        pf = O.prove_machine_keyed(tr, pre, pg, tb, pub, params)
        assert O.verify_machine_keyed(pf, lns, ws, pws, root, pg, tb, pub, params) == 0
        keyed[name] = {"machine": list(args), "params": list(prm), "root": [int(v) for v in root], "bytes": int(pf.size),
                       "sha256": hashlib.sha256(pf.tobytes()).hexdigest()}
    out["keyed_machine_proofs"] = keyed
    # the Poseidon2 chip (tests/poseidon2_air.py): its program's digest and one proof of Merkle paths
    import poseidon2_air
    leaves, sibs, idx, root = poseidon2_air.tree_paths(4, 7, seed=11)
    trace, _ = poseidon2_air.merkle_trace(leaves, sibs, idx)
    prog = poseidon2_air.program()
    params = O.default_params(1, 8, 4)
    pf = O.prove_shard_air(prog, trace, root + [7], params)
    assert O.verify_shard_air(prog, pf, 5, poseidon2_air.WIDTH, root + [7], params) == 0
    out["p2chip"] = {"paths": [4, 7, 11], "params": [1, 8, 4], "program_words": int(prog.size), "program_digest": [int(v) for v in O.air_digest(prog)],
                     "root": [int(v) for v in root], "bytes": int(pf.size), "sha256": hashlib.sha256(pf.tobytes()).hexdigest()}
    # complete proof BYTES of small shards, one per proof version (tests/golden/proofs/*.bin): what the independent pure-Python
    # verifier (tests/pyverify.py, written from DESIGN.md sections 3 and 6) and the product's host verifier check on the CPU, and
    # what the HIP prover must reproduce byte for byte on the GPU.  shape = (log_blowup, queries, pow_bits, logup_pairs,
    # log_fold, log_final, hash_width)
    os.makedirs(os.path.join(HERE, "proofs"), exist_ok=True)
    golden = {}
    for name, (log_n, w, shape) in {"v1_6x8": (6, 8, (1, 4, 4, 0, 0, 0, 0)), "v1_10x16": (10, 16, (1, 6, 8, 0, 0, 0, 0)),
                                    "v2_lookup_7x16": (7, 16, (1, 3, 4, 1, 0, 0, 0)),
                                    "v3_r0_9x8": (9, 8, (2, 3, 0, 0, 4, 1, 24)), "v3_fold8_9x8": (9, 8, (1, 3, 5, 0, 3, 0, 16)),
                                    "v3_blowup4_lookup_8x16": (8, 16, (2, 3, 3, 2, 1, 0, 16)),
                                    "v8_groups_r0_lookup_8x16": (8, 16, (2, 3, 0, 1, 4, 0, 24, 4))}.items():
        t = O.gen_trace_logup(SEED, 5, log_n, w, shape[3]) if shape[3] else O.gen_trace(SEED, 5, log_n, w)
        prm = O.default_params(*shape)
        pf = O.prove_shard(t, [1, 2, 3], prm)
        assert O.verify_shard(pf, log_n, w, [1, 2, 3], prm) == 0
        with open(os.path.join(HERE, "proofs", name + ".bin"), "wb") as f:
            f.write(pf.tobytes())
        golden[name] = {"log_n": log_n, "width": w, "shape": list(shape), "seed": SEED, "shard": 5, "public": [1, 2, 3],
                        "bytes": int(pf.size), "sha256": hashlib.sha256(pf.tobytes()).hexdigest()}
    out["golden_proof_files"] = golden
    # provenance: which state of oracle/ produced this file
    import subprocess
    root = os.path.dirname(os.path.dirname(HERE))
    try:
        commit = subprocess.run(["git", "log", "-1", "--format=%H", "--", "oracle"], cwd=root, capture_output=True, text=True, check=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "oracle"], cwd=root, capture_output=True, text=True, check=True).stdout.strip())
    except Exception:
        commit, dirty = "unknown", False
    src = hashlib.sha256()
    for fn in sorted(os.listdir(os.path.join(root, "oracle"))):
        if fn.endswith((".c", ".h")):
            src.update(open(os.path.join(root, "oracle", fn), "rb").read())
    out["provenance"] = {"oracle_commit": commit, "oracle_tree_dirty": dirty, "oracle_sources_sha256": src.hexdigest(),
                         "generator": "tests/golden/make_golden.py", "note": "every fixture in this file and in proofs/ was produced by this repo's own CPU oracle at that commit (parity unpinned)"}
    with open(os.path.join(HERE, "oracle_kat.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote oracle_kat.json")


if __name__ == "__main__":
    main()
