"""N > 1 path on CPU: two gloo ranks run the shard-parallel scheduling used by bench.py
(partition, seed broadcast, max-over-ranks timing, digest gather).  The per-shard prover is
replaced by a byte-producing stub here ON PURPOSE: libzkhip has no CPU path, and what this
test covers is the distribution logic, not arithmetic."""
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


def test_partition_properties():
    from zktls_amd.shards import shard_indices
    for world in (1, 2, 4, 8):
        for total in (0, 1, 7, 64):
            parts = [shard_indices(total, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(total))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)
