"""MACHINE MODE of the shard verifier machine on the GPU (csrc/machine_verifier.inl through the C ABI): version-11 keyed-machine proofs verified
in-circuit -- key and outer proof bytes against the oracle's generic keyed-machine prover run on the Python restatement's arrays
(tests/recursion_machine.py) -- and THE TREE: shard proofs -> joins (zkhip_prove_shard_verifier) -> ONE proof that verifies the joins
(zkhip_prove_machine_verifier), checked from (the join machine's description, the shard proofs' public values, the key)."""
import numpy as np
import pytest

import machines as M
import recursion_air as R
import recursion_machine as RM
from zktls_amd._lib import Params, ZkHipError
from zktls_amd.device import InnerMachine, machine_verifier_key_host, shard_verifier_describe, verify_machine_recursive, verify_shard_recursive

pytestmark = pytest.mark.gpu
SEED = 0x5A4B544C53


def inner_of(O, mains, pres, progs, tabs, pub, q, pb):
    lns = [m.shape[0].bit_length() - 1 for m in mains]
    prm = O.default_params(1, q, pb)
    vk = [int(x) for x in O.machine_setup(pres, lns, prm)]
    chips = [dict(ln=lns[c], W=mains[c].shape[1], Pw=0 if pres[c] is None else pres[c].shape[1], prog=progs[c], tab=tabs[c]) for c in range(len(mains))]
    return chips, vk, O.prove_machine_keyed(mains, pres, progs, tabs, pub, prm)


@pytest.mark.parametrize("which", ["byte-6-3", "byte-7-3", "byte-5-3", "random-4", "random-6"])
def test_key_and_proof_bytes_equal_the_oracles(ctx, oracle, which):
    O = oracle
    kind, *args = which.split("-")
    q, pb = 3, 1
    made = [M.byte_machine(int(args[0]), int(args[1]), seed) for seed in (1, 2)] if kind == "byte" else [M.random_keyed_machine(int(args[0]))]
    chips, vk, p0 = inner_of(O, *made[0], q, pb)
    proofs, pubs = [p0], [made[0][4]]
    if len(made) > 1:                                                            # a join of two proofs of one machine
        proofs.append(inner_of(O, *made[1], q, pb)[2]), pubs.append(made[1][4])
    n = len(proofs)
    prm, oprm = Params(1, 20, 8), O.default_params(1, 20, 8)
    im = InnerMachine(chips, vk, q, pb, len(pubs[0]))
    key = ctx.machine_verifier_setup(im, prm, n)
    sh, mains, pres, progs, tabs, pv = RM.machine(chips, vk, [p.tobytes() for p in proofs], pubs, q, pb)
    lns = [m.shape[0].bit_length() - 1 for m in mains]
    assert key.root.tolist() == O.machine_setup(pres, lns, oprm).tolist(), "the key differs from the oracle's commitment to the restatement's preprocessed traces"
    assert machine_verifier_key_host(im, prm, n).tolist() == key.root.tolist()
    outer = ctx.prove_machine_verifier(key, im, proofs, pubs, prm)
    assert outer.tobytes() == O.prove_machine_keyed(mains, pres, progs, tabs, pv, oprm).tobytes(), "outer proof bytes differ from the oracle's"
    assert verify_machine_recursive(im, outer, pv, key.root, prm, n) == (0, 0)
    if which == "byte-7-3":                                                     # the host walks the Poseidon2 rows sixteen queries at a time (AVX-512) or one by one: same rows
        from zktls_amd import _lib
        prev = _lib.load().zkhip_host_simd(0)
        try:
            assert ctx.prove_machine_verifier(key, im, proofs, pubs, prm).tobytes() == outer.tobytes()
        finally:
            _lib.load().zkhip_host_simd(prev)
    other = list(pv)
    other[-1] = (other[-1] + 1) % R.P
    assert verify_machine_recursive(im, outer, other, key.root, prm, n)[0] != 0
    assert verify_machine_recursive(im, outer, pv, (key.root + 1) % R.P, prm, n)[0] != 0
    bad = proofs[0].copy()
    bad[bad.size // 2] ^= 1
    with pytest.raises(ZkHipError):
        ctx.prove_machine_verifier(key, im, [bad] + proofs[1:], pubs, prm)
    key.close()


def join_machine(log_n, width, q, pb, npub, n_proofs, key_root, oq, opb):
    """the shard verifier machine for n_proofs shard proofs of a shape, as the inner machine of machine mode: the library's own description of it"""
    chips = []
    for i in range(8):
        p, ln, mw, pw = shard_verifier_describe(log_n, width, q, pb, npub, i, 0, n_proofs)
        t, _, _, _ = shard_verifier_describe(log_n, width, q, pb, npub, i, 1, n_proofs)
        chips.append(dict(ln=ln, W=mw, Pw=pw, prog=p, tab=t))
    return chips, InnerMachine(chips, key_root, oq, opb, npub * n_proofs)


def test_the_tree_four_shard_proofs_two_joins_one_proof(ctx, oracle):
    """level 0: four shard proofs; level 1: two joins of two (zkhip_prove_shard_verifier: eight chips, version 11); level 2: ONE proof that verifies both
    joins in-circuit.  Bytes against the oracle on the restatement's arrays; the verifier takes the join machine's description, the four
    shards' public values and the key"""
    O = oracle
    log_n, width, q, pb = 6, 16, 3, 1
    iprm, jprm, tprm, oprm = Params(1, q, pb), Params(1, 4, 2), Params(1, 20, 8), O.default_params(1, 20, 8)
    pubs = [[5, 6, 30 + s] for s in range(4)]
    shards = [ctx.prove_shard(ctx.gen_trace(SEED, 40 + s, log_n, width), log_n, width, pubs[s], iprm) for s in range(4)]
    jkey = ctx.shard_verifier_setup(log_n, width, q, pb, 3, jprm, n_proofs=2)
    joins = [ctx.prove_shard_verifier(jkey, shards[2 * j:2 * j + 2], log_n, width, pubs[2 * j:2 * j + 2], iprm, jprm) for j in range(2)]
    jpubs = [pubs[0] + pubs[1], pubs[2] + pubs[3]]
    for j in range(2):
        assert verify_shard_recursive(joins[j], log_n, width, q, pb, jpubs[j], jkey.root, jprm, n_proofs=2) == (0, 0)
    chips, im = join_machine(log_n, width, q, pb, 3, 2, jkey.root, 4, 2)
    tkey = ctx.machine_verifier_setup(im, tprm, 2)
    top = ctx.prove_machine_verifier(tkey, im, joins, jpubs, tprm)
    flat = [v for p_ in jpubs for v in p_]
    assert verify_machine_recursive(im, top, flat, tkey.root, tprm, 2) == (0, 0)
    assert machine_verifier_key_host(im, tprm, 2).tolist() == tkey.root.tolist()
    sh, mains, pres, progs, tabs, pv = RM.machine(chips, [int(x) for x in jkey.root], [j.tobytes() for j in joins], jpubs, 4, 2)
    lns = [m.shape[0].bit_length() - 1 for m in mains]
    assert tkey.root.tolist() == O.machine_setup(pres, lns, oprm).tolist()
    assert top.tobytes() == O.prove_machine_keyed(mains, pres, progs, tabs, pv, oprm).tobytes(), "the tree's top differs from the oracle's proof"
    swapped = pubs[2] + pubs[3] + pubs[0] + pubs[1]
    assert verify_machine_recursive(im, top, swapped, tkey.root, tprm, 2)[0] != 0
    print("tree: 4 shard proofs %d B -> 2 joins %d B -> top %d B" % (sum(s.size for s in shards), sum(j.size for j in joins), top.size))
    jkey.close(), tkey.close()


def test_a_two_level_tree_over_sixty_four_headline_shard_proofs(ctx):
    """VERDICT r4 item 1, done-criterion b: BASELINE configs[1] x 64 -- sixty-four 2^20 x 256 shard proofs (61 MB) -> four joins of sixteen (1.6 MB each)
    -> ONE proof; its verifier takes the join machine's description (a function of the shard shape), the 64 x 9 public values and the key
    (derived on the host)"""
    log_n, width, q, pb, n_join, n_top = 20, 256, 100, 16, 16, 4
    prm = Params(1, 100, 16)
    pubs = [[1, 2, 3, 4, 5, 6, 7, 8, 500 + s] for s in range(n_join * n_top)]
    shards = []
    for s in range(n_join * n_top):
        tr = ctx.gen_trace(SEED, 700 + s, log_n, width)
        shards.append(ctx.prove_shard(tr, log_n, width, pubs[s], prm))
        tr.free()
    jkey = ctx.shard_verifier_setup(log_n, width, q, pb, 9, prm, n_proofs=n_join)
    joins = [ctx.prove_shard_verifier(jkey, shards[n_join * j:n_join * (j + 1)], log_n, width, pubs[n_join * j:n_join * (j + 1)], prm, prm) for j in range(n_top)]
    jpubs = [[v for p_ in pubs[n_join * j:n_join * (j + 1)] for v in p_] for j in range(n_top)]
    chips, im = join_machine(log_n, width, q, pb, 9, n_join, jkey.root, 100, 16)
    tkey = ctx.machine_verifier_setup(im, prm, n_top)
    top = ctx.prove_machine_verifier(tkey, im, joins, jpubs, prm)
    flat = [v for p_ in jpubs for v in p_]
    assert verify_machine_recursive(im, top, flat, machine_verifier_key_host(im, prm, n_top), prm, n_top) == (0, 0)
    bad = list(flat)
    bad[9 * 37 + 8] += 1                                                        # shard 37's last public value
    assert verify_machine_recursive(im, top, bad, tkey.root, prm, n_top)[0] != 0
    total = sum(s.size for s in shards)
    assert top.size * 20 < total
    print("tree: 64 shard proofs %d B -> 4 joins %d B -> top %d B (1 / %.1f)" % (total, sum(j.size for j in joins), top.size, total / top.size))
    jkey.close(), tkey.close()


def sha256_machine_desc(log_n_sha, key_root, q, pb):
    """the keyed SHA-256 machine of zkhip_prove_transcripts (the compression chip + the preprocessed 2^16-row range table) as an inner machine:
    programs and tables as tests/machines.py's sha256_machine builds them (equal to the library's: tests/test_gpu_keyed_machine.py)"""
    import oracle_lib as O
    import sha256_air as S
    sent = [S.OUT + 6, S.OUT + 7, S.OUT + 14, S.OUT + 15]
    sha = dict(ln=log_n_sha, W=S.WIDTH, Pw=0, prog=S.program(), tab=O.interaction_table([(O.SEND, None, 16, [c]) for c in sent]))
    table = dict(ln=16, W=4, Pw=4, prog=O.air_program(8, S.N_PUBLIC, [(O.SEL_FIRST, [(1, [O.air_var(0)])])]), tab=O.interaction_table([(O.RECEIVE, 5, 16, [0])]))
    chips = [sha, table] if log_n_sha > 16 else [table, sha]
    return chips, InnerMachine(chips, key_root, q, pb, S.N_PUBLIC)


def test_sixty_four_keyed_transcript_proofs_become_one_proof(ctx):
    """VERDICT r4 item 1, done-criterion a, to the letter: 64 transcript proofs of zkhip_prove_transcripts (the KEYED SHA-256 machine: version 11, two chips
    of heights 2^16 and 2^14, a preprocessed table, a bus) -> ONE proof, verified with (the machine's description, 64 x (digest, length), the key)"""
    import hashlib
    from zktls_amd.device import prove_transcripts, sha256_padding_publics
    n, nbytes = 64, 13221
    prm = Params(1, 100, 16)
    msgs = [bytes((5 * i + 11 * p + 3) & 0xff for i in range(nbytes)) for p in range(n)]
    vk, res = prove_transcripts(msgs, prm, devices=[0])
    pubs = []
    for m, (d, _) in zip(msgs, res):
        assert d == hashlib.sha256(m).digest()
        limbs = []
        for i in range(8):
            w = int.from_bytes(d[4 * i:4 * i + 4], "big")
            limbs += [w & 0xffff, w >> 16]
        pubs.append(limbs + sha256_padding_publics(len(m)).tolist())
    from zktls_amd.device import sha256_inner_machine
    im = sha256_inner_machine(nbytes, vk, prm)                                  # (zkhip_sha256_machine_describe: the library's own description; sha256_machine_desc above is the test-side one)
    assert [k[0].tolist() for k in im.keep] == [np.asarray(c["prog"]).tolist() for c in sha256_machine_desc(14, vk, 100, 16)[0]]
    tkey = ctx.machine_verifier_setup(im, prm, n)
    top = ctx.prove_machine_verifier(tkey, im, [p for _, p in res], pubs, prm)
    flat = [v for p_ in pubs for v in p_]
    assert verify_machine_recursive(im, top, flat, tkey.root, prm, n) == (0, 0)
    bad = list(flat)
    bad[91 * 23 + 5] ^= 1                                                       # transcript 23's digest
    assert verify_machine_recursive(im, top, bad, tkey.root, prm, n)[0] != 0
    total = sum(p.size for _, p in res)
    assert top.size * 10 < total
    print("64 keyed transcript proofs: %d B -> %d B (1 / %.1f)" % (total, top.size, total / top.size))
    tkey.close()


def described_machine(im, n_proofs, key_root, oq, opb, npub_total):
    """the machine-mode machine over n_proofs proofs of `im` as an inner machine itself: the library's description of its ten chips"""
    from zktls_amd.device import machine_verifier_describe
    chips = []
    for i in range(10):
        p, ln, mw, pw = machine_verifier_describe(im, i, 0, n_proofs)
        t, _, _, _ = machine_verifier_describe(im, i, 1, n_proofs)
        chips.append(dict(ln=ln, W=mw, Pw=pw, prog=p, tab=t))
    return chips, InnerMachine(chips, key_root, oq, opb, npub_total)


def test_the_machine_verifies_its_own_proofs_three_levels(ctx, oracle):
    """the recursion closes: a machine-mode proof is a version-11 proof of a ten-chip keyed machine, which machine mode takes like any other.
    shard proofs -> joins (level 1) -> tops over the joins (level 2) -> ONE proof over the tops (level 3); its bytes against the oracle on the
    restatement's arrays, its verifier handed the level-2 machine's description, the shards' public values and the key"""
    O = oracle
    log_n, width, q, pb = 5, 8, 2, 0
    iprm, p1, p2, p3, o3 = Params(1, q, pb), Params(1, 2, 0), Params(1, 2, 1), Params(1, 20, 8), O.default_params(1, 20, 8)
    pubs = [[9, 40 + s] for s in range(2)]
    shards = [ctx.prove_shard(ctx.gen_trace(SEED, 90 + s, log_n, width), log_n, width, pubs[s], iprm) for s in range(2)]
    jkey = ctx.shard_verifier_setup(log_n, width, q, pb, 2, p1, n_proofs=1)
    joins = [ctx.prove_shard_verifier(jkey, shards[s], log_n, width, pubs[s], iprm, p1) for s in range(2)]
    chips1, im1 = join_machine(log_n, width, q, pb, 2, 1, jkey.root, 2, 0)
    k2 = ctx.machine_verifier_setup(im1, p2, 1)
    tops = [ctx.prove_machine_verifier(k2, im1, [joins[s]], [pubs[s]], p2) for s in range(2)]
    for s in range(2):
        assert verify_machine_recursive(im1, tops[s], pubs[s], k2.root, p2, 1) == (0, 0)
    chips2, im2 = described_machine(im1, 1, k2.root, 2, 1, 2)
    k3 = ctx.machine_verifier_setup(im2, p3, 2)
    top3 = ctx.prove_machine_verifier(k3, im2, tops, pubs, p3)
    flat = pubs[0] + pubs[1]
    assert verify_machine_recursive(im2, top3, flat, k3.root, p3, 2) == (0, 0)
    assert verify_machine_recursive(im2, top3, pubs[1] + pubs[0], k3.root, p3, 2)[0] != 0
    assert machine_verifier_key_host(im2, p3, 2).tolist() == k3.root.tolist()
    sh, mains, pres, progs, tabs, pv = RM.machine(chips2, [int(x) for x in k2.root], [t.tobytes() for t in tops], pubs, 2, 1)
    lns = [m.shape[0].bit_length() - 1 for m in mains]
    assert k3.root.tolist() == O.machine_setup(pres, lns, o3).tolist()
    assert top3.tobytes() == O.prove_machine_keyed(mains, pres, progs, tabs, pv, o3).tobytes(), "the level-3 proof differs from the oracle's"
    print("three levels: shard proofs %d B -> joins %d B -> tops %d B -> one proof %d B" % (sum(x.size for x in shards), sum(x.size for x in joins), sum(x.size for x in tops), top3.size))
    jkey.close(), k2.close(), k3.close()


def test_a_tree_over_statements_about_real_data(ctx):
    """four SHA-256 proofs ("digest = SHA-256 of a message of L bytes") -> two air-mode joins of two (nine chips: the EVAL chip over the SHA-256 program)
    -> ONE proof over the joins (machine mode takes the nine-chip machine like any other); its verifier is handed the join machine's
    description, the four (digest, length) statements and the key"""
    import hashlib
    from zktls_amd.device import sha256_air, sha256_padding_publics
    q, pb = 4, 2
    iprm, jprm, tprm = Params(1, q, pb), Params(1, 3, 1), Params(1, 20, 8)
    prog = sha256_air()
    msgs = [bytes((3 * i + 17 * p + 1) & 0xff for i in range(100)) for p in range(4)]
    inner, pubs = [], []
    for m in msgs:
        d, pf = ctx.prove_sha256(m, iprm)
        assert d == hashlib.sha256(m).digest()
        limbs = []
        for i in range(8):
            w = int.from_bytes(d[4 * i:4 * i + 4], "big")
            limbs += [w & 0xffff, w >> 16]
        inner.append(pf), pubs.append(limbs + sha256_padding_publics(len(m)).tolist())
    jkey = ctx.shard_verifier_setup(7, 640, q, pb, 91, jprm, n_proofs=2, program=prog)
    joins = [ctx.prove_shard_verifier(jkey, inner[2 * j:2 * j + 2], 7, 640, pubs[2 * j:2 * j + 2], iprm, jprm, program=prog) for j in range(2)]
    chips = []
    for i in range(9):
        p_, ln, mw, pw = shard_verifier_describe(7, 640, q, pb, 91, i, 0, 2, program=prog)
        t_, _, _, _ = shard_verifier_describe(7, 640, q, pb, 91, i, 1, 2, program=prog)
        chips.append(dict(ln=ln, W=mw, Pw=pw, prog=p_, tab=t_))
    im = InnerMachine(chips, jkey.root, 3, 1, 2 * 91)
    tkey = ctx.machine_verifier_setup(im, tprm, 2)
    jpubs = [pubs[0] + pubs[1], pubs[2] + pubs[3]]
    top = ctx.prove_machine_verifier(tkey, im, joins, jpubs, tprm)
    flat = [v for p_ in jpubs for v in p_]
    assert verify_machine_recursive(im, top, flat, machine_verifier_key_host(im, tprm, 2), tprm, 2) == (0, 0)
    bad = list(flat)
    bad[91 * 3 + 1] ^= 1                                                        # the fourth message's digest
    assert verify_machine_recursive(im, top, bad, tkey.root, tprm, 2)[0] != 0
    bad = flat[:91 * 2] + pubs[2][:16] + sha256_padding_publics(99).tolist() + flat[91 * 3:]      # the third message: another length
    assert verify_machine_recursive(im, top, bad, tkey.root, tprm, 2)[0] != 0
    jkey.close(), tkey.close()


@pytest.mark.parametrize("in_flight", [1, 3])
def test_the_tree_in_one_call_is_the_joins_and_the_top_made_one_by_one(ctx, in_flight):
    """zkhip_prove_shard_tree: six shard proofs -> three joins of two (in flight on pooled contexts; each one's tables for the top filled on its worker's
    thread the moment it exists) -> ONE proof; the joins and the top are the bytes of zkhip_prove_shard_verifier / zkhip_prove_machine_verifier called one
    by one; a tampered shard proof, a join machine with another key and a count that is no multiple of the join size are refused"""
    from zktls_amd.device import prove_shard_tree
    log_n, width, q, pb = 6, 16, 3, 1
    iprm, jprm, tprm = Params(1, q, pb), Params(1, 4, 2), Params(1, 20, 8)
    pubs = [[5, 6, 30 + s] for s in range(6)]
    shards = [ctx.prove_shard(ctx.gen_trace(SEED, 140 + s, log_n, width), log_n, width, pubs[s], iprm) for s in range(6)]
    jkey = ctx.shard_verifier_setup(log_n, width, q, pb, 3, jprm, n_proofs=2)
    joins = [ctx.prove_shard_verifier(jkey, shards[2 * j:2 * j + 2], log_n, width, pubs[2 * j:2 * j + 2], iprm, jprm) for j in range(3)]
    jpubs = [pubs[2 * j] + pubs[2 * j + 1] for j in range(3)]
    chips, im = join_machine(log_n, width, q, pb, 3, 2, jkey.root, 4, 2)
    tkey = ctx.machine_verifier_setup(im, tprm, 3)
    top = ctx.prove_machine_verifier(tkey, im, joins, jpubs, tprm)
    for rep in range(2):
        top1, joins1, jvk = prove_shard_tree(ctx, tkey, im, shards, 2, log_n, width, pubs, iprm, jprm, tprm, devices=[0], in_flight=in_flight)
        assert jvk.tolist() == jkey.root.tolist()
        assert [j.tobytes() for j in joins1] == [j.tobytes() for j in joins]
        assert top1.tobytes() == top.tobytes()
    assert verify_machine_recursive(im, top1, [v for p in jpubs for v in p], tkey.root, tprm, 3) == (0, 0)
    bad = [p.copy() for p in shards]
    bad[4][bad[4].size // 2] ^= 1
    with pytest.raises(ZkHipError) as e:
        prove_shard_tree(ctx, tkey, im, bad, 2, log_n, width, pubs, iprm, jprm, tprm, devices=[0], in_flight=in_flight)
    assert e.value.code == -6
    with pytest.raises(ZkHipError):
        prove_shard_tree(ctx, tkey, im, shards[:5], 2, log_n, width, pubs[:5], iprm, jprm, tprm, devices=[0])
    _, other = join_machine(log_n, width, q, pb, 3, 2, [(int(x) + 1) % 2013265921 for x in jkey.root], 4, 2)
    with pytest.raises(ZkHipError):
        prove_shard_tree(ctx, tkey, other, shards, 2, log_n, width, pubs, iprm, jprm, tprm, devices=[0])          # (a join machine with another key: its tables reject the joins)
    jkey.close(), tkey.close()


# ---------------------------------------------------------------------------------------------------------------- SP1's shard structure, core -> compress
SP1_SMALL = [(8, 24, 3, 1), (8, 32, 3, 0), (7, 16, 2, -1), (6, 32, 2, -1), (5, 8, 1, -1)]


def sp1_shards(ctx, shape, seed, shards, prm):
    """-> (key, inner machine, [proof], [public values]) for the listed shards of one SP1-shaped machine, proven on the device"""
    key, keep = shape.setup(ctx, seed, prm)
    proofs, pubs = [], []
    for s in shards:
        traces = shape.gen_traces(ctx, seed, s)
        pv = [(11 * (i + 1)) % R.P for i in range(shape.n_public - 1)] + [s]
        proofs.append(ctx.prove_machine_keyed(key, shape.main_chips(traces), shape.programs, shape.tables, pv, prm))
        pubs.append(pv)
        for t in traces:
            t.free()
    return key, keep, shape.inner_machine(key.root, prm), proofs, pubs


def test_the_sp1_shaped_shard_core_to_compress_bytes_equal_the_oracles(ctx, oracle):
    """VERDICT r5 item 3 at a small shape: three shards of SP1's structure (five chips of four heights, in-table LogUp pairs, a cross-table bus, preprocessed
    columns) proven on the device as version-11 proofs of ONE keyed machine -- key and bytes == the oracle's on tests/machines.py's traces --, joined in machine
    mode: the join's key and bytes == the oracle's on the restatement's arrays; accepted from (description, public values, key); the shards swapped: refused"""
    from zktls_amd.device import Sp1ShapedShard
    O = oracle
    q, pb, seed = 3, 1, 5
    iprm, oiprm = Params(1, q, pb), O.default_params(1, q, pb)
    shape = Sp1ShapedShard(SP1_SMALL, ((3, 8),), 3)
    key, keep, im, proofs, pubs = sp1_shards(ctx, shape, seed, [0, 1, 2], iprm)
    for s in range(3):
        mains, pres, progs, tabs, pub = M.sp1_shaped_machine(SP1_SMALL, seed=seed, shard=s, pre=((3, 8),))
        assert pub == pubs[s]
        lns = [m.shape[0].bit_length() - 1 for m in mains]
        assert key.root.tolist() == O.machine_setup(pres, lns, oiprm).tolist(), "the shard machine's key differs from the oracle's"
        assert proofs[s].tobytes() == O.prove_machine_keyed(mains, pres, progs, tabs, pub, oiprm).tobytes(), "shard %d: proof bytes differ from the oracle's" % s
    prm, oprm = Params(1, 20, 8), O.default_params(1, 20, 8)
    tkey = ctx.machine_verifier_setup(im, prm, 3)
    top = ctx.prove_machine_verifier(tkey, im, proofs, pubs, prm)
    chips = [dict(ln=ln, W=w, Pw=pw, prog=g, tab=t) for ln, w, pw, g, t in zip(shape.log_ns, shape.widths, shape.pre_widths, shape.programs, shape.tables)]
    sh, mains, pres, progs, tabs, pv = RM.machine(chips, [int(x) for x in key.root], [p.tobytes() for p in proofs], pubs, q, pb)
    lns = [m.shape[0].bit_length() - 1 for m in mains]
    assert tkey.root.tolist() == O.machine_setup(pres, lns, oprm).tolist(), "the join's key differs from the oracle's commitment to the restatement's preprocessed traces"
    assert machine_verifier_key_host(im, prm, 3).tolist() == tkey.root.tolist()
    assert top.tobytes() == O.prove_machine_keyed(mains, pres, progs, tabs, pv, oprm).tobytes(), "the join's bytes differ from the oracle's"
    flat = pubs[0] + pubs[1] + pubs[2]
    assert pv == flat and verify_machine_recursive(im, top, flat, tkey.root, prm, 3) == (0, 0)
    assert verify_machine_recursive(im, top, pubs[1] + pubs[0] + pubs[2], tkey.root, prm, 3)[0] != 0
    with pytest.raises(ZkHipError):                                             # the shard PROOFS swapped under the stated order
        ctx.prove_machine_verifier(tkey, im, [proofs[1], proofs[0], proofs[2]], pubs, prm)
    key.close(), tkey.close()


def test_sp1_shaped_shards_at_the_bench_size_join_and_top(ctx, oracle):
    """the bench's six-chip shard (2^20 x 96, 2^20 x 32 looking each other up, 2^19 x 64, 2^18 x 128, 2^16 x 256 with 32 preprocessed columns, 2^14 x 40;
    LogUp pairs 3, 3, 2, 4, 8, 1) as a keyed machine: four shard proofs accepted by the oracle's verifier and the host's, joined two by two in machine
    mode, the two joins topped -- the top checked from (the join machine's description, the four shards' public values, the key derived on the host)"""
    from zktls_amd.device import Sp1ShapedShard, verify_machine_keyed
    O = oracle
    prm, oprm = Params(1, 100, 16), O.default_params(1, 100, 16)
    shape = Sp1ShapedShard()
    key, keep, im, proofs, pubs = sp1_shards(ctx, shape, SEED, [0, 1, 2, 3], prm)
    widths = shape.widths
    for s in (0, 3):
        assert O.verify_machine_keyed(proofs[s], shape.log_ns, widths, shape.pre_widths, key.root, shape.programs, shape.tables, pubs[s], oprm) == 0, "the oracle refuses shard %d" % s
    for s in range(4):
        assert verify_machine_keyed(proofs[s], shape.log_ns, widths, shape.pre_widths, key.root, shape.programs, shape.tables, pubs[s], prm) == (0, 0)
    assert verify_machine_keyed(proofs[1], shape.log_ns, widths, shape.pre_widths, key.root, shape.programs, shape.tables, pubs[2], prm)[0] != 0
    jkey = ctx.machine_verifier_setup(im, prm, 2)
    joins = [ctx.prove_machine_verifier(jkey, im, proofs[2 * j:2 * j + 2], pubs[2 * j:2 * j + 2], prm) for j in range(2)]
    jpubs = [pubs[0] + pubs[1], pubs[2] + pubs[3]]
    for j in range(2):
        assert verify_machine_recursive(im, joins[j], jpubs[j], jkey.root, prm, 2) == (0, 0)
    assert verify_machine_recursive(im, joins[0], jpubs[1], jkey.root, prm, 2)[0] != 0
    chips2, im2 = described_machine(im, 2, jkey.root, 100, 16, 2 * shape.n_public)
    tkey = ctx.machine_verifier_setup(im2, prm, 2)
    top = ctx.prove_machine_verifier(tkey, im2, joins, jpubs, prm)
    flat = jpubs[0] + jpubs[1]
    assert verify_machine_recursive(im2, top, flat, machine_verifier_key_host(im2, prm, 2), prm, 2) == (0, 0)
    assert verify_machine_recursive(im2, top, jpubs[1] + jpubs[0], tkey.root, prm, 2)[0] != 0
    print("SP1-shaped shards: 4 x %d B -> 2 joins %d B -> top %d B" % (proofs[0].size, sum(j.size for j in joins), top.size))
    key.close(), jkey.close(), tkey.close()


def test_device_witnesses_and_the_hosts_walk_give_one_proof(ctx, oracle):
    """round 6: the per-query tables of the recursion machines (ROWSUM, QUERY, FOLD, the queries' Poseidon2 rows) are filled by device kernels from the inner
    proofs' words; zkhip_recursion_witnesses_on_host(1) keeps the host's walk.  Same machine, same inner proofs -> the same outer proof bytes: machine mode (a join of
    two keyed-machine proofs) and the shard verifier (a join of two shard proofs, plain and air mode)"""
    from zktls_amd.device import air_synthetic
    O = oracle
    made = [M.byte_machine(7, 3, seed) for seed in (1, 2)]
    q, pb = 3, 1
    chips, vk, p0 = inner_of(O, *made[0], q, pb)
    proofs, pubs = [p0, inner_of(O, *made[1], q, pb)[2]], [made[0][4], made[1][4]]
    im = InnerMachine(chips, vk, q, pb, len(pubs[0]))
    prm = Params(1, 20, 8)
    key = ctx.machine_verifier_setup(im, prm, 2)
    log_n, width = 6, 16
    iprm, jprm = Params(1, q, pb), Params(1, 4, 2)
    spubs = [[5, 6, 30 + s] for s in range(2)]
    shards = [ctx.prove_shard(ctx.gen_trace(SEED, 40 + s, log_n, width), log_n, width, spubs[s], iprm) for s in range(2)]
    jkey = ctx.shard_verifier_setup(log_n, width, q, pb, 3, jprm, n_proofs=2)
    prog = air_synthetic(width, 3)
    ashards = [ctx.prove_shard_air(prog, ctx.gen_trace(SEED, 40 + s, log_n, width), log_n, width, spubs[s], iprm) for s in range(2)]
    akey = ctx.shard_verifier_setup(log_n, width, q, pb, 3, jprm, n_proofs=2, program=prog)

    def all_three():
        return (ctx.prove_machine_verifier(key, im, proofs, pubs, prm).tobytes(), ctx.prove_shard_verifier(jkey, shards, log_n, width, spubs, iprm, jprm).tobytes(),
                ctx.prove_shard_verifier(akey, ashards, log_n, width, spubs, iprm, jprm, program=prog).tobytes())
    from zktls_amd.device import recursion_witnesses_on_host
    prev = recursion_witnesses_on_host(0)
    try:
        dev = all_three()
        recursion_witnesses_on_host(1)
        host = all_three()
    finally:
        recursion_witnesses_on_host(prev)
    assert dev == host
    key.close(), jkey.close(), akey.close()
