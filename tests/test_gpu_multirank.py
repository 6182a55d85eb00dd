"""The N > 1 path of bench.py and RCCL itself on whatever GPUs the box has.
Both tests start bench.py as a FRESH child process (the way tests/test_server.py starts its server): bench.py's parent spawns the
ranks before anything in it touches the GPU, and nothing here replaces a process that has initialised HIP.
  * two ranks on one GPU (--share-gpu: gloo collectives, because RCCL refuses two ranks on one device): the shard dealing, the seed
    broadcast, the max-over-ranks timing and the digest gather run end to end, every shard is proven on exactly one rank;
  * one rank with --force-collective: the 'nccl' (= RCCL) process group is initialised on the GPU and a broadcast, a MAX
    all-reduce and a barrier go through it, so RCCL is loaded and called on hardware in every driver run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(*flags, timeout=900):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), env=env, cwd=ROOT, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_two_ranks_share_one_gpu_and_prove_every_shard_once():
    r = run_bench("--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--log-n", "18", "--width", "64")
    assert r["n_gpus"] == 2 and r["rccl_world_size"] == 2 and r["share_gpu_test_mode"] is True
    assert r["collective_backend"] == "gloo"
    assert r["shards_proven"] == 4 and r["distinct_shards_proven"] == 4 and r["shard_digests_gathered"] == 4
    assert r["verified"] is True and r["scaling"] == "weak"
    assert r["value"] > 0 and r["metric"] == "trace-cells/s"


def test_rccl_is_initialised_and_called_at_world_size_one():
    r = run_bench("--gpus", "1", "--force-collective", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--log-n", "18", "--width", "64")
    assert r["collective_backend"] == "nccl" and r["rccl_selfcheck_calls"] == 3
    assert r["rccl_world_size"] == 1 and r["shards_proven"] == 2 and r["verified"] is True
