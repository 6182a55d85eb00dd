import sys, os, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from zktls_amd._lib import Params, ZkHipError
from zktls_amd.device import Context, sha256_air, sha256_padding_publics, verify_sha256, verify_shard_recursive, shard_verifier_key_host, shard_verifier_describe
ctx = Context(0)
prog = sha256_air()
W, NPUB = 640, 91
def statement(digest, L):
    limbs = []
    for i in range(8):
        w = int.from_bytes(digest[4 * i:4 * i + 4], "big")
        limbs += [w & 0xffff, w >> 16]
    return limbs + sha256_padding_publics(L).tolist()
for nproofs, nbytes, q, pb in ((1, 100, 6, 3), (3, 150, 5, 2), (8, 13221, 20, 8)):
    iprm, prm = Params(1, q, pb), Params(1, 20, 8)
    msgs = [bytes((7 * i + 3 * p + 1) & 0xff for i in range(nbytes)) for p in range(nproofs)]
    inners, pubs = [], []
    for m in msgs:
        d, pf = ctx.prove_sha256(m, iprm)
        assert d == hashlib.sha256(m).digest() and verify_sha256(pf, d, iprm, len(m)) == (0, 0)
        inners.append(pf); pubs.append(statement(d, len(m)))
    log_n = int(np.frombuffer(inners[0][8:12].tobytes(), dtype=np.uint32)[0])
    t0 = time.perf_counter()
    key = ctx.shard_verifier_setup(log_n, W, q, pb, NPUB, prm, n_proofs=nproofs, program=prog)
    t1 = time.perf_counter()
    for rep in range(2):
        t2 = time.perf_counter()
        outer = ctx.prove_shard_verifier(key, inners, log_n, W, pubs, iprm, prm, program=prog)
        t3 = time.perf_counter()
    flat = [v for p in pubs for v in p]
    ok = verify_shard_recursive(outer, log_n, W, q, pb, flat, key.root, prm, n_proofs=nproofs, program=prog)
    bad = list(flat); bad[3] ^= 1
    bad2 = list(flat); bad2[16] += 1          # another block count = another length
    print("%d SHA-256 proofs of %d bytes (2^%d x %d, %d queries): inner %d B each -> outer %d B; setup %.1f ms, join %.1f ms; verify %s; wrong digest %s; wrong length %s"
          % (nproofs, nbytes, log_n, W, q, inners[0].size, outer.size, (t1 - t0) * 1e3, (t3 - t2) * 1e3, ok,
             verify_shard_recursive(outer, log_n, W, q, pb, bad, key.root, prm, n_proofs=nproofs, program=prog)[0],
             verify_shard_recursive(outer, log_n, W, q, pb, bad2, key.root, prm, n_proofs=nproofs, program=prog)[0]), flush=True)
    hk = shard_verifier_key_host(log_n, W, q, pb, NPUB, prm, n_proofs=nproofs, program=prog)
    print("  host key equal:", hk.tolist() == key.root.tolist(), " heights:", [shard_verifier_describe(log_n, W, q, pb, NPUB, i, 0, nproofs, program=prog)[1] for i in range(9)], flush=True)
    key.close()
