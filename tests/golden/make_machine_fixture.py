#!/usr/bin/env python3
"""Makes tests/golden/proofs/machine_blob_small_x3.bin, machine_tree_blob_small_x5.bin and machine_blob_small.vk ON A GPU BOX: the host mirror's compress stage
(zktls_guest_prove_machine, zktls_amd/host: setup -> core -> compress, sp1.rs:113-116) for shards in SP1's shard structure at a small shape -- three shards -> ONE
machine-mode proof; five shards in joins of at most two -> three joins -> ONE proof above them (flag TREE).  tests/test_host_mirror_machine.py checks them with NO device.
    gpurun -- python3 tests/golden/make_machine_fixture.py gpurun_out/fixtures      (then copy the three files to tests/golden/proofs/)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))

if __name__ == "__main__":
    import test_host_mirror_machine as T
    out_dir = sys.argv[1]
    os.makedirs(out_dir, exist_ok=True)
    L = T.lib.__wrapped__() if hasattr(T.lib, "__wrapped__") else None
    assert L is not None
    rc, err, _, blob, vk = T.prove(L, 2, T.mplan(T.SP1_SMALL, T.PRE, 3, 3, 1), compress=1)
    assert rc == 0, err
    L.zktls_set_compress_join_size(2)
    rc, err, _, tblob, tvk = T.prove(L, 2, T.mplan(T.SP1_SMALL, T.PRE, 5, 3, 1), compress=1)
    assert rc == 0 and tvk == vk, err
    for name, data in (("machine_blob_small_x3.bin", blob), ("machine_tree_blob_small_x5.bin", tblob), ("machine_blob_small.vk", vk)):
        open(os.path.join(out_dir, name), "wb").write(data)
        print("wrote %d bytes to %s" % (len(data), name))
