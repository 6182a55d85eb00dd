"""Shard-parallel scheduling across the GPUs of one node.

Shards are independent units: rank r proves shards r, r + world, r + 2*world, ... with no
data-path collective.  The only exchange is one broadcast of the batch transcript seed from
rank 0 (RCCL on GPUs; the same code runs over gloo in the CPU tests) and, for reporting,
one MAX all-reduce of the elapsed time and a gather of per-shard proof digests."""
import hashlib


def shard_indices(total_shards, rank, world):
    """round-robin partition: every shard exactly once, balanced within one shard"""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, total_shards, world))


def broadcast_seed(dist, seed_words, device=None):
    """rank 0's 8 seed words reach every rank (the Fiat-Shamir batch seed)"""
    import torch
    t = torch.tensor(list(seed_words), dtype=torch.int32, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=0)
    return [int(x) for x in t.tolist()]


def max_over_ranks(dist, seconds, device=None):
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_proof_digests(dist, local):
    """local: {shard_index: proof bytes}; returns {shard_index: sha256 hex} on every rank"""
    mine = {int(k): hashlib.sha256(bytes(v)).hexdigest() for k, v in local.items()}
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return mine
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, mine)
    merged = {}
    for d in out:
        for k, v in d.items():
            if k in merged:
                raise RuntimeError("shard %d proven twice" % k)
            merged[k] = v
    return merged


def prove_batch(prove_one, total_shards, rank, world):
    """run prove_one(shard_index) -> bytes for this rank's shards"""
    return {s: prove_one(s) for s in shard_indices(total_shards, rank, world)}
