"""N > 1 path on CPU: two gloo ranks run the shard-parallel scheduling used by bench.py
(partition, seed broadcast, max-over-ranks timing, digest gather).  libzkhip has no CPU path,
so the per-shard prover is either a byte-producing stub (distribution logic alone) or -- in
test_two_ranks_real_proofs -- the CPU oracle proving a tiny shard, with every proof checked by
the PRODUCT's host verifier (zkhip_verify_shard) and the dealing taken from the library's own
zkhip_shard_device, the function zkhip_prove_shards_multi deals shards with."""
import hashlib
import os
import socket

import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    import torch.distributed as dist
    from zktls_amd import shards
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seed = shards.broadcast_seed(dist, [11 * (rank + 1) + i for i in range(8)])
    local = shards.prove_batch(lambda s: hashlib.sha256(bytes(seed) + bytes([s])).digest() * 4, total, rank, world)
    elapsed = shards.max_over_ranks(dist, 1.0 + rank)
    merged = shards.gather_proof_digests(dist, local)
    q.put((rank, seed, sorted(local), elapsed, merged))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 8])
def test_two_rank_shard_parallel(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, seed0, mine0, t0, m0), (r1, seed1, mine1, t1, m1) = res
    assert seed0 == seed1 == [11 + i for i in range(8)]          # rank 0's seed everywhere
    assert sorted(mine0 + mine1) == list(range(total))            # every shard exactly once
    assert abs(len(mine0) - len(mine1)) <= 1
    assert t0 == t1 == 2.0                                        # max over ranks
    assert m0 == m1 and sorted(m0) == list(range(total))


def _worker_real(rank, world, port, total, q):
    """the payload is a real proof: oracle-proven tiny shards, verified by libzkhip's host verifier"""
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import oracle_lib as O
    from zktls_amd import _lib, shards
    from zktls_amd.device import shard_device, verify_shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O.set_threads(1)
    seed = shards.broadcast_seed(dist, [3 + i for i in range(8)] if rank == 0 else [0] * 8)
    log_n, width = 6, 8
    prm, oprm = _lib.Params(1, 8, 4), O.default_params(1, 8, 4)
    mine = [s for s in range(total) if shard_device(s, None, world) == rank]      # the library's dealing function
    assert mine == shards.shard_indices(total, rank, world)

    def prove_one(s):
        proof = O.prove_shard(O.gen_trace(0x5A4B544C53, s, log_n, width), seed + [s], oprm)
        assert verify_shard(proof, log_n, width, seed + [s], prm) == (0, 0)          # product verifier accepts it
        assert verify_shard(proof, log_n, width, seed + [s + 1], prm)[0] == -6        # ... and binds the shard index
        return proof.tobytes()
    local = shards.prove_batch(prove_one, total, rank, world)
    merged = shards.gather_proof_digests(dist, local)
    q.put((rank, sorted(local), merged))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_real_proofs():
    total = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, mine0, m0), (_, mine1, m1) = res
    assert mine0 == [0, 2, 4] and mine1 == [1, 3]
    assert m0 == m1 and sorted(m0) == list(range(total))
    assert len(set(m0.values())) == total                          # five different proofs


def test_partition_properties():
    from zktls_amd.shards import shard_indices
    for world in (1, 2, 4, 8):
        for total in (0, 1, 7, 64):
            parts = [shard_indices(total, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(total))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)
