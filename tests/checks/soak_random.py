"""exploration: a long pseudo-random sweep on the GPU -- every proof must equal the oracle's bytes and verify.
usage: python tests/checks/soak_random.py [seconds]"""
import sys, time
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import oracle_lib as O
import airs
from zktls_amd.device import Context, verify_shard, verify_chips, verify_shard_air, verify_chips_air, verify_machine, verify_machine_keyed, verify_shard_recursive
import machines
import poseidon2_air
import recursion_air
from zktls_amd._lib import Params
O.set_threads(8)
P = O.P


def with_quintic_identity(prog, col):
    """the program with one more constraint, x^5 - x^5 = 0 on column `col` written as two degree-5 terms: true on every trace, and enough
    to give the table four quotient chunks"""
    extra = [O.SEL_ALL, 2, 1, 5] + [O.air_var(col)] * 5 + [P - 1, 5] + [O.air_var(col)] * 5
    out = np.concatenate([prog, np.array(extra, dtype=np.uint32)])
    out[3] += 1
    out[5] = out.size
    return out
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(time.time()))
ctx = Context(0)
t0 = time.time(); n_single = n_chips = n_air = n_machine = n_lookup = n_keyed = n_p2 = n_rec = n_rec_air = n_rec_machine = 0
SEED = int(rng.integers(1, 2**40))
while time.time() - t0 < budget:
    r_kind = rng.random()
    if r_kind < 0.004:
        # the shard verifier machine / the join: 1..3 shard proofs of a random small shape verified inside ONE outer proof; key and bytes against
        # the oracle's keyed-machine prover on the Python restatement's arrays (tests/recursion_air.py); the verifier takes no inner proof
        log_n, width, q, pb = int(rng.integers(5, 9)), 8 * int(rng.integers(1, 5)), int(rng.integers(1, 9)), int(rng.integers(0, 5))      # (the machine takes widths in multiples of 8)
        npub, nproofs = int(rng.integers(0, 4)), int(rng.integers(1, 4))
        pubs = [[int(x) for x in rng.integers(0, 2013265921, npub)] for _ in range(nproofs)]
        oshape = (1, int(rng.integers(4, 12)), int(rng.integers(0, 6)))
        iprm, prm, oprm = Params(1, q, pb), Params(*oshape), O.default_params(*oshape)
        inner = []
        for p_ in range(nproofs):
            tr = ctx.gen_trace(SEED, 900 + p_, log_n, width)
            inner.append(ctx.prove_shard(tr, log_n, width, pubs[p_], iprm)); tr.free()
        key = ctx.shard_verifier_setup(log_n, width, q, pb, npub, prm, n_proofs=nproofs)
        sh, mains, pres, progs, tabs, pv = recursion_air.machine([x.tobytes() for x in inner], log_n, width, pubs, q, pb)
        lns = [m.shape[0].bit_length() - 1 for m in mains]
        assert key.root.tolist() == O.machine_setup(pres, lns, oprm).tolist(), ("shard verifier key", log_n, width, q, pb, npub, nproofs, oshape)
        outer = ctx.prove_shard_verifier(key, inner, log_n, width, pubs, iprm, prm)
        assert outer.tobytes() == O.prove_machine_keyed(mains, pres, progs, tabs, pv, oprm).tobytes(), ("shard verifier", log_n, width, q, pb, npub, nproofs, oshape)
        assert verify_shard_recursive(outer, log_n, width, q, pb, [v for p_ in pubs for v in p_], key.root, prm, n_proofs=nproofs) == (0, 0)
        key.close()
        n_rec += 1
    elif r_kind < 0.008:
        # the same machine in AIR MODE: 1..3 version-7 proofs of a pseudo-random constraint program (degree <= 3, all three selectors, public
        # values, next-row variables) verified inside ONE outer proof; key and bytes against the oracle on the restatement's arrays
        log_n, width, q, pb = int(rng.integers(5, 9)), 8 * int(rng.integers(1, 4)), int(rng.integers(1, 7)), int(rng.integers(0, 5))
        nproofs = int(rng.integers(1, 4))
        pseed = int(rng.integers(0, 2**31))
        oshape = (1, int(rng.integers(4, 12)), int(rng.integers(0, 6)))
        iprm, prm, oprm = Params(1, q, pb), Params(*oshape), O.default_params(*oshape)
        prog, trace, pub = airs.random_program_and_trace(pseed, log_n, width, int(rng.choice([2, 3])))
        if O.air_log_quotient_degree(prog) != 1: continue
        # (one program, hence one key: the proofs differ in their public values -- the counter's start -- through other seeds of the SAME structure
        # is not what random_program_and_trace offers, so the join repeats the statement with fresh proofs of it)
        inner, pubs = [], []
        for p_ in range(nproofs):
            inner.append(ctx.prove_shard_air(prog, ctx.from_numpy(trace), log_n, width, pub, iprm)); pubs.append(list(pub))
        key = ctx.shard_verifier_setup(log_n, width, q, pb, 3, prm, n_proofs=nproofs, program=prog)
        sh, mains, pres, progs, tabs, pv = recursion_air.machine([x.tobytes() for x in inner], log_n, width, pubs, q, pb, program=prog)
        lns = [m.shape[0].bit_length() - 1 for m in mains]
        assert key.root.tolist() == O.machine_setup(pres, lns, oprm).tolist(), ("air-mode key", pseed, log_n, width, q, pb, nproofs, oshape)
        outer = ctx.prove_shard_verifier(key, inner, log_n, width, pubs, iprm, prm, program=prog)
        assert outer.tobytes() == O.prove_machine_keyed(mains, pres, progs, tabs, pv, oprm).tobytes(), ("air-mode join", pseed, log_n, width, q, pb, nproofs, oshape)
        assert verify_shard_recursive(outer, log_n, width, q, pb, [v for p_ in pubs for v in p_], key.root, prm, n_proofs=nproofs, program=prog) == (0, 0)
        key.close()
        n_rec_air += 1
    elif r_kind < 0.011:
        # MACHINE MODE: a proof of a pseudo-random keyed machine (2 .. 5 tables of random heights, tuples of 1 .. 8 values, preprocessed columns) verified
        # in-circuit; key and bytes against the oracle on the restatement's arrays (tests/recursion_machine.py)
        import recursion_machine as RMM
        from zktls_amd.device import InnerMachine, verify_machine_recursive
        mseed = int(rng.integers(0, 2**31))
        mains_, pres_, progs_, tabs_, pub_ = machines.random_keyed_machine(mseed)
        if any(p_ is None for p_ in progs_) or any(t_ is None for t_ in tabs_): continue
        q_, pb_ = int(rng.integers(1, 5)), int(rng.integers(0, 3))
        lns_ = [m_.shape[0].bit_length() - 1 for m_ in mains_]
        iprm_ = O.default_params(1, q_, pb_)
        ivk = [int(x) for x in O.machine_setup(pres_, lns_, iprm_)]
        iproof = O.prove_machine_keyed(mains_, pres_, progs_, tabs_, pub_, iprm_)
        chips_ = [dict(ln=lns_[c], W=mains_[c].shape[1], Pw=0 if pres_[c] is None else pres_[c].shape[1], prog=progs_[c], tab=tabs_[c]) for c in range(len(mains_))]
        oshape = (1, int(rng.integers(4, 12)), int(rng.integers(0, 6)))
        im_ = InnerMachine(chips_, ivk, q_, pb_, len(pub_))
        key = ctx.machine_verifier_setup(im_, Params(*oshape), 1)
        sh_, m2, p2, g2, t2, pv2 = RMM.machine(chips_, ivk, [iproof.tobytes()], [pub_], q_, pb_)
        lns2 = [m_.shape[0].bit_length() - 1 for m_ in m2]
        assert key.root.tolist() == O.machine_setup(p2, lns2, O.default_params(*oshape)).tolist(), ("machine-mode key", mseed, q_, pb_, oshape)
        outer = ctx.prove_machine_verifier(key, im_, [iproof], [pub_], Params(*oshape))
        assert outer.tobytes() == O.prove_machine_keyed(m2, p2, g2, t2, pv2, O.default_params(*oshape)).tobytes(), ("machine mode", mseed, q_, pb_, oshape)
        assert verify_machine_recursive(im_, outer, pv2, key.root, Params(*oshape), 1) == (0, 0)
        key.close()
        n_rec_machine += 1
    elif r_kind < 0.013:
        # THE TREE in one call (zkhip_prove_shard_tree: joins in flight on pooled contexts, each one's tables for the top filled on its worker's thread): a random
        # small shape, 2 .. 3 joins of 1 .. 3 shard proofs; the joins and the top against the step-by-step entries (whose bytes the kinds above hold to the oracle's)
        from zktls_amd.device import InnerMachine, prove_shard_tree, shard_verifier_describe, verify_machine_recursive
        log_n, width, q, pb = int(rng.integers(5, 8)), 8 * int(rng.integers(1, 4)), int(rng.integers(1, 6)), int(rng.integers(0, 4))
        npub, J, nj = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(2, 4))
        pubs = [[int(x) for x in rng.integers(0, 2013265921, npub)] for _ in range(J * nj)]
        jsh, tsh = (1, int(rng.integers(2, 6)), int(rng.integers(0, 3))), (1, int(rng.integers(4, 12)), int(rng.integers(0, 6)))
        inner = [ctx.prove_shard(ctx.gen_trace(SEED, int(rng.integers(0, 2**31)), log_n, width), log_n, width, pubs[i], Params(1, q, pb)) for i in range(J * nj)]
        jkey = ctx.shard_verifier_setup(log_n, width, q, pb, npub, Params(*jsh), n_proofs=J)
        joins = [ctx.prove_shard_verifier(jkey, inner[J * j:J * (j + 1)], log_n, width, pubs[J * j:J * (j + 1)], Params(1, q, pb), Params(*jsh)) for j in range(nj)]
        chips_ = []
        for i in range(8):
            p_, ln_, mw_, pw_ = shard_verifier_describe(log_n, width, q, pb, npub, i, 0, J)
            t_, _, _, _ = shard_verifier_describe(log_n, width, q, pb, npub, i, 1, J)
            chips_.append(dict(ln=ln_, W=mw_, Pw=pw_, prog=p_, tab=t_))
        im_ = InnerMachine(chips_, jkey.root, jsh[1], jsh[2], J * npub)
        tkey = ctx.machine_verifier_setup(im_, Params(*tsh), nj)
        jflat = [[v for p_ in pubs[J * j:J * (j + 1)] for v in p_] for j in range(nj)]
        top = ctx.prove_machine_verifier(tkey, im_, joins, jflat, Params(*tsh))
        top1, joins1, jvk1 = prove_shard_tree(ctx, tkey, im_, inner, J, log_n, width, pubs, Params(1, q, pb), Params(*jsh), Params(*tsh), devices=[0], in_flight=int(rng.integers(1, 4)))
        assert top1.tobytes() == top.tobytes() and [x.tobytes() for x in joins1] == [x.tobytes() for x in joins] and jvk1.tolist() == jkey.root.tolist(), ("tree in one call", log_n, width, q, pb, npub, J, nj, jsh, tsh)
        assert verify_machine_recursive(im_, top1, [v for f_ in jflat for v in f_], tkey.root, Params(*tsh), nj) == (0, 0)
        jkey.close(), tkey.close()
        n_rec_machine += 1
    elif r_kind < 0.015:
        # THE PLUG POINT for shards in SP1's shard structure (zktls_amd/host MachinePlan): a random small machine (2 .. 5 chips, LogUp pairs in-table and across
        # two tables of one height, preprocessed columns) x 1 .. 4 shards through setup -> prove -> verify; plain: every shard proof and the key against the oracle
        # on tests/machines.py's traces; compressed (joins of at most 1 .. 3): the blob checked on the host from (plan, input, ELF, vk), another request refused
        import ctypes as C_
        import test_host_mirror_machine as TM
        if "mirror" not in globals():
            mirror = TM.lib.__wrapped__()
        hs = sorted((int(x) for x in rng.integers(5, 10, int(rng.integers(2, 6)))), reverse=True)
        spec = []
        for ln in hs:
            w = 8 * int(rng.integers(1, 5))
            spec.append([ln, w, int(rng.integers(1, w // 8 + 1)), -1])
        for i in range(len(spec) - 1):
            if spec[i][0] == spec[i + 1][0] and spec[i][3] < 0 and rng.random() < 0.7:
                pr_ = min(spec[i][2], spec[i + 1][2])
                spec[i][2] = spec[i + 1][2] = pr_
                spec[i][3], spec[i + 1][3] = i + 1, i
                break
        cand = [c for c, ch in enumerate(spec) if ch[3] < 0 and ch[1] >= 16]
        if not cand: continue                                # (a keyed machine has preprocessed columns somewhere)
        must = int(rng.choice(cand))
        pre = tuple((c, 8) for c in cand if c == must or rng.random() < 0.3)
        spec = [tuple(c) for c in spec]
        shards, q, pb = int(rng.integers(1, 5)), int(rng.integers(1, 5)), int(rng.integers(0, 3))
        cbor, elf = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)), b"\x7fELF" + bytes(rng.integers(0, 256, 8, dtype=np.uint8))
        compress = rng.random() < 0.5
        plan = TM.mplan(spec, pre, shards, q, pb, in_flight=int(rng.integers(0, 3)))
        if compress: mirror.zktls_set_compress_join_size(int(rng.integers(1, 4)))
        try:
            rc, err, out, blob, vk = TM.prove(mirror, 2, plan, cbor=cbor, elf=elf, compress=1 if compress else 0, setup_first=int(rng.integers(0, 2)))
            assert rc == 0, ("mirror machine", spec, pre, shards, q, pb, err)
            dg = TM.digest_words(mirror, cbor, elf)
            if not compress:
                ent = TM.entries(mirror, blob)[0]
                assert len(ent) == shards
                seed_, kseed_ = TM.stream_seed(dg), TM.stream_seed(TM.digest_words(mirror, b"", elf))
                oprm_ = O.default_params(1, q, pb)
                for s_ in range(shards):
                    mains_, pres_, progs_, tabs_, _ = machines.sp1_shaped_machine(spec, seed=seed_, shard=s_, pre=pre, n_public=9, key_seed=kseed_)
                    assert vk[:32] == O.machine_setup(pres_, [c[0] for c in spec], oprm_).tobytes(), ("mirror machine key", spec, pre)
                    assert ent[s_] == O.prove_machine_keyed(mains_, pres_, progs_, tabs_, dg + [s_], oprm_).tobytes(), ("mirror machine shard", spec, pre, shards, q, pb, s_)
            assert TM.check(mirror, blob, plan, vk, cbor=cbor, elf=elf) == 0, ("mirror machine verify", spec, pre, shards, q, pb, compress)
            assert TM.check(mirror, blob, plan, vk, cbor=cbor + b"!", elf=elf) == -2
        finally:
            mirror.zktls_set_compress_join_size(0)
        n_mirror = globals().get("n_mirror", 0) + 1
    elif r_kind < 0.017:
        # the Poseidon2 chip: random Merkle paths of a random tree; device trace against the Python restatement, proof bytes against the oracle
        depth, n_paths = int(rng.integers(1, 6)), int(rng.integers(1, 12))
        leaves, sibs, idx, root = poseidon2_air.tree_paths(depth, n_paths, seed=int(rng.integers(0, 2**31)))
        trace, _ = poseidon2_air.merkle_trace(leaves, sibs, idx)
        shape = (int(rng.integers(1, 4)), int(rng.integers(1, 12)), int(rng.integers(0, 7)))
        d, droots, log_n = ctx.p2chip_gen_merkle_trace(leaves, sibs, idx)
        assert (d.download().reshape(-1, poseidon2_air.WIDTH) == trace).all(), ("p2chip trace", depth, n_paths)
        d.free()
        pf = ctx.prove_merkle_paths(leaves, sibs, idx, root, Params(*shape))
        assert pf.tobytes() == O.prove_shard_air(poseidon2_air.program(), trace, root + [n_paths], O.default_params(*shape)).tobytes(), ("p2chip", depth, n_paths, shape)
        n_p2 += 1
    elif r_kind < 0.03:
        # a keyed machine: some tables have preprocessed columns, committed once by setup (version 11); two proofs against one key
        if rng.random() < 0.4:
            traces, pre, progs, tables, pub = machines.byte_machine(int(rng.integers(5, 11)), int(rng.integers(3, 6)), seed=int(rng.integers(0, 2**31)))
        else:
            traces, pre, progs, tables, pub = machines.random_keyed_machine(int(rng.integers(0, 2**31)))
        lns, ws = [t.shape[0].bit_length() - 1 for t in traces], [t.shape[1] for t in traces]
        pws = [0 if p_ is None else p_.shape[1] for p_ in pre]
        prm = (int(rng.integers(1, 4)), int(rng.integers(1, 12)), int(rng.integers(0, 7)))
        if prm[0] >= 2 and rng.random() < 0.5:                                   # one table's program gains a degree-5 constraint: four quotient chunks for that table
            k = int(rng.integers(0, len(traces)))
            progs = list(progs)
            progs[k] = with_quintic_identity(progs[k], 0)
        dev = [ctx.from_numpy(t) for t in traces]
        dpre = [None if p_ is None else ctx.from_numpy(p_) for p_ in pre]
        key = ctx.machine_setup(list(zip(dpre, lns, pws)), Params(*prm))
        assert key.root.tolist() == O.machine_setup(pre, lns, O.default_params(*prm)).tolist(), ("key", lns, pws, prm)
        for d in dpre:
            if d is not None: d.free()                                           # the key holds its own copies
        want = O.prove_machine_keyed(traces, pre, progs, tables, pub, O.default_params(*prm)).tobytes()
        for _ in range(2):
            pf = ctx.prove_machine_keyed(key, list(zip(dev, lns, ws)), progs, tables, pub, Params(*prm))
            assert pf.tobytes() == want, ("keyed machine", lns, ws, pws, prm)
        assert verify_machine_keyed(pf, lns, ws, pws, key.root, progs, tables, pub, Params(*prm)) == (0, 0)
        key.close()
        for d in dev: d.free()
        n_keyed += 1
    elif r_kind < 0.05:
        # a machine whose tables look each other up (interaction tables: multiplicities, buses, 1- and 2-tuples), plus bystanders
        lt = int(rng.integers(5, 9)); lu = int(rng.integers(lt, 11))
        if rng.random() < 0.5:
            traces, progs, tables, pub = machines.range_machine(lt, lu, seed=int(rng.integers(0, 2**31)))
        else:
            traces, progs, tables, pub = machines.random_machine(int(rng.integers(0, 2**31)))
        extra = []
        for i in range(int(rng.integers(0, 3)) if len(pub) == 3 else 0):        # synthetic bystanders (range machine only: they share its public values)
            hgt = int(rng.integers(5, lu + 2)); extra.append(O.gen_trace(SEED, 50 + i, hgt, 4 * int(rng.integers(1, 6))))
        allt = sorted([(t, p_, tb) for t, p_, tb in zip(traces, progs, tables)] + [(t, None, None) for t in extra], key=lambda e: -e[0].shape[0])
        traces, progs, tables = [e[0] for e in allt], [e[1] for e in allt], [e[2] for e in allt]
        lns, ws = [t.shape[0].bit_length() - 1 for t in traces], [t.shape[1] for t in traces]
        if max(lns.count(x) for x in lns) > 4: continue
        prm = (int(rng.integers(1, 4)), int(rng.integers(1, 12)), int(rng.integers(0, 7)))
        dev = [ctx.from_numpy(t) for t in traces]
        pf = ctx.prove_machine(list(zip(dev, lns, ws)), progs, tables, pub, Params(*prm))
        assert pf.tobytes() == O.prove_machine(traces, progs, tables, pub, O.default_params(*prm)).tobytes(), ("lookup machine", lns, ws, prm)
        assert verify_machine(pf, lns, ws, progs, tables, pub, Params(*prm)) == (0, 0)
        for d in dev: d.free()
        n_lookup += 1
    elif r_kind < 0.1:
        # a machine: several tables with their own constraint programs (or the synthetic AIR) in one proof -- version 9.
        # Table 0 is a pseudo-random degree-<=3 AIR; counter tables and synthetic tables share its three public values.
        n = int(rng.integers(1, 7))
        hs = sorted((int(x) for x in rng.integers(5, 11, n)), reverse=True)
        if max(hs.count(x) for x in hs) > 4: continue
        w0 = 4 * int(rng.integers(1, 6))
        prog0, t0_, pub = airs.random_program_and_trace(int(rng.integers(0, 2**31)), hs[0], w0, int(rng.choice([2, 3])))
        traces, progs = [t0_], [prog0]
        for i, hgt in enumerate(hs[1:], 1):
            w = 4 * int(rng.integers(1, 10))
            if rng.random() < 0.5:
                cp = airs.counter_program(w).copy(); cp[4] = 3
                traces.append(airs.counter_trace(hgt, w, pub[0], pub[1])[0]); progs.append(cp)
            else:
                traces.append(O.gen_trace(SEED, i, hgt, w)); progs.append(None)
        prm = (int(rng.integers(1, 4)), int(rng.integers(1, 12)), int(rng.integers(0, 7)))
        dev = [ctx.from_numpy(t) for t in traces]
        lns, ws = [t.shape[0].bit_length() - 1 for t in traces], [t.shape[1] for t in traces]
        pf = ctx.prove_chips_air(list(zip(dev, lns, ws)), progs, pub, Params(*prm))
        assert pf.tobytes() == O.prove_chips_air(traces, progs, pub, O.default_params(*prm)).tobytes(), ("machine", lns, ws, prm)
        assert verify_chips_air(pf, lns, ws, progs, pub, Params(*prm)) == (0, 0)
        for d in dev: d.free()
        n_machine += 1
    elif r_kind < 0.25:
        # a pseudo-random AIR supplied as a constraint program (degree 2..5: two or four quotient chunks)
        log_n, width, maxdeg = int(rng.integers(5, 11)), 4 * int(rng.integers(1, 6)), int(rng.choice([2, 3, 4, 5]))
        prog, trace, pub = airs.random_program_and_trace(int(rng.integers(0, 2**31)), log_n, width, maxdeg)
        lqd = O.air_log_quotient_degree(prog)
        K = int(rng.integers(1, 4))
        fs = [f for f in range(0, min(log_n, 6) + 1) if (log_n - f) % K == 0]
        if not fs: continue
        shape = (int(rng.integers(lqd, 4)), int(rng.integers(1, 12)), int(rng.integers(0, 6)), 0, K, int(rng.choice(fs)), int(rng.choice([16, 24])))
        pf = ctx.prove_shard_air(prog, ctx.from_numpy(trace), log_n, width, pub, Params(*shape))
        assert pf.tobytes() == O.prove_shard_air(prog, trace, pub, O.default_params(*shape)).tobytes(), ("air", log_n, width, maxdeg, shape)
        assert verify_shard_air(prog, pf, log_n, width, pub, Params(*shape)) == (0, 0)
        n_air += 1
    elif r_kind < 0.68:
        log_n = int(rng.integers(5, 13)); width = 4 * int(rng.integers(1, 25)); b = int(rng.integers(1, 4)); K = int(rng.integers(1, 5))
        fs = [f for f in range(0, min(log_n, 9) + 1) if (log_n - f) % K == 0]
        if not fs: continue
        F = int(rng.choice(fs)); hw = int(rng.choice([16, 24])); pairs = int(rng.integers(0, width // 8 + 1)) if rng.random() < 0.4 else 0
        cw = 4 * int(rng.integers(1, width // 4)) if width > 4 and rng.random() < 0.35 else 0     # code / data group split
        shape = (b, int(rng.integers(1, 15)), int(rng.integers(0, 8)), pairs, K, F, hw, cw)
        shard = int(rng.integers(0, 1000)); pub = [int(x) for x in rng.integers(0, 2013265921, int(rng.integers(0, 5)))]
        d = ctx.gen_trace_logup(SEED, shard, log_n, width, pairs) if pairs else ctx.gen_trace(SEED, shard, log_n, width)
        h = O.gen_trace_logup(SEED, shard, log_n, width, pairs) if pairs else O.gen_trace(SEED, shard, log_n, width)
        pf = ctx.prove_shard(d, log_n, width, pub, Params(*shape))
        assert pf.tobytes() == O.prove_shard(h, pub, O.default_params(*shape)).tobytes(), ("single", log_n, width, shape)
        assert verify_shard(pf, log_n, width, pub, Params(*shape)) == (0, 0)
        d.free(); n_single += 1
    else:
        n = int(rng.integers(1, 9))
        hs = sorted((int(x) for x in rng.integers(5, 12, n)), reverse=True)
        if max(hs.count(x) for x in hs) > 4: continue
        chips = []
        for hgt in hs:
            w = 4 * int(rng.integers(1, 14)); chips.append([hgt, w, int(rng.integers(0, w // 8 + 1)) if rng.random() < 0.4 else 0, -1])
        # make partnerships among equal-height chips with equal pair counts
        for i in range(n):
            for j in range(i + 1, n):
                if chips[i][3] < 0 and chips[j][3] < 0 and chips[i][0] == chips[j][0] and chips[i][2] and rng.random() < 0.5:
                    q = min(chips[i][2], chips[j][2], chips[i][1] // 8, chips[j][1] // 8)
                    if q and chips[j][2]:
                        chips[i][2] = chips[j][2] = q; chips[i][3] = j; chips[j][3] = i
        prm = (int(rng.integers(1, 4)), int(rng.integers(1, 12)), int(rng.integers(0, 7)))
        dev, host = [], []
        for i, (ln, w, pr, pa) in enumerate(chips):
            if pa >= 0:
                dev.append(ctx.gen_trace_logup_cross(SEED, i, pa, ln, w, chips[pa][1], pr)); host.append(O.gen_trace_logup_cross(SEED, i, pa, ln, w, chips[pa][1], pr))
            elif pr:
                dev.append(ctx.gen_trace_logup(SEED, i, ln, w, pr)); host.append(O.gen_trace_logup(SEED, i, ln, w, pr))
            else:
                dev.append(ctx.gen_trace(SEED, i, ln, w)); host.append(O.gen_trace(SEED, i, ln, w))
        prs, pas = [c[2] for c in chips], [c[3] for c in chips]
        cross = any(p >= 0 for p in pas)
        pf = ctx.prove_chips([(d, c[0], c[1], c[2], c[3]) for d, c in zip(dev, chips)], [7], Params(*prm))
        assert pf.tobytes() == O.prove_chips(host, [7], O.default_params(*prm), prs, pas if cross else None).tobytes(), ("chips", chips, prm)
        assert verify_chips(pf, [c[0] for c in chips], [c[1] for c in chips], [7], Params(*prm), prs, pas if cross else None) == (0, 0)
        for d in dev: d.free()
        n_chips += 1
print("ok: %d single-matrix, %d multi-chip, %d constraint-program, %d chips-with-programs, %d lookup-machine, %d keyed-machine, %d Poseidon2-chip, %d shard-verifier (join), %d air-mode and %d machine-mode shard-verifier, %d host-mirror machine-plan configurations in %.0f s"
      % (n_single, n_chips, n_air, n_machine, n_lookup, n_keyed, n_p2, n_rec, n_rec_air, n_rec_machine, globals().get("n_mirror", 0), time.time() - t0))
