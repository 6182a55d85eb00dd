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


def test_one_process_deals_the_shards_through_the_librarys_device_list():
    """bench.py --one-process: zkhip_prove_shards_multi(NULL, 0, ...) is the measuring path (the entry a ZkProver::prove binds on a multi-GPU node)"""
    r = run_bench("--gpus", "1", "--one-process", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-batch64", "--no-recursion16",
                  "--log-n", "18", "--width", "128")
    assert r["one_process_mode"] is True and r["n_gpus"] == 1 and r["logical_devices_test_mode"] == 0
    assert r["shards_proven"] == 3 and r["distinct_shards_proven"] == 3
    assert r["config"]["parallelism"].startswith("one process, device list")
    assert r["verified"] is True and r["value"] > 0


def test_one_process_over_two_logical_devices():
    """the same with a device list of TWO: the A/B build's logical devices (own pools, own workers, traces generated where their shard is dealt)"""
    r = run_bench("--gpus", "2", "--one-process", "--logical-devices", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-batch64",
                  "--no-recursion16", "--log-n", "18", "--width", "128")
    assert r["one_process_mode"] is True and r["n_gpus"] == 2 and r["logical_devices_test_mode"] == 2
    assert r["shards_proven"] == 6 and r["distinct_shards_proven"] == 6
    assert r["verified"] is True


def test_the_rank_per_gpu_line_carries_the_one_process_entry_too():
    r = run_bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-batch64", "--no-recursion16", "--log-n", "18", "--width", "128")
    assert r["one_process"]["shards_proven"] == 2 and r["one_process"]["distinct_proofs"] == 2 and r["one_process"]["same_bytes_as_the_timed_step"] is True
    assert r["one_process_value"] > 0
    # ... and the plug point itself: one call of the host mirror of ZkProver::prove for 22 shards, core and core + compress
    e = r["execution22"]
    assert e["compressed_blob_verified_on_host"] is True and e["compressed_blob_bytes"] * 4 < e["core_blob_bytes"] and e["core_plus_compress_ms"] > e["core_ms"] * 0.5


def test_multi_device_entries_on_two_logical_devices():
    """tests/checks/multi_device_logical.py: device traces dealt where they live (and refused where they do not), the NULL device list, the
    lock-step dealer, host traces, a transcript batch and a batch of joins (the compress stage) -- all over a device list of two, bytes against the oracle"""
    env = dict(os.environ)
    env["ZKHIP_LOGICAL_DEVICES"] = "2"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", "multi_device_logical.py")], env=env, cwd=ROOT, timeout=900,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    r = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert r["logical_devices"] == 2 and r["device_traces_dealt_where_they_live"] == 6 and r["misplaced_trace_refused"] is True
    assert r["null_device_list_same_bytes"] is True and r["lockstep_small_shards"] == 16 and r["host_traces"] == 5 and r["transcripts_over_the_device_list"] == 8 and r["joins_over_the_device_list"] == 5 and r["tree_in_one_call_over_the_device_list"] is True


def test_dry_run_of_the_eight_gpu_configuration_on_logical_devices():
    """the 8-GPU configuration dry-run LOGICALLY on the one GPU -- a device list of eight (A/B build), four shards in flight per
    device = 32 prover threads under the container's CPU quota, so `--wait auto` must pick blocking waits; every shard proven exactly once.  No
    scaling number comes out of this; it removes first-contact failures from the day an 8-GPU node exists."""
    r = run_bench("--gpus", "8", "--one-process", "--logical-devices", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-batch64",
                  "--no-recursion16", "--log-n", "18", "--width", "64", timeout=1500)
    assert r["one_process_mode"] is True and r["n_gpus"] == 8 and r["logical_devices_test_mode"] == 8
    assert r["shards_proven"] == 16 and r["distinct_shards_proven"] == 16
    assert r["host_wait"] == "block" and r["host_cores_busy_per_rank"] is not None
    assert r["verified"] is True and r["value"] > 0


def test_dry_run_of_eight_ranks_sharing_the_one_gpu():
    """the rank-per-GPU path at world size 8 (gloo collectives: RCCL refuses several ranks on one device): dealing, seed broadcast, max-over-ranks
    timing and the digest gather with eight processes; blocking waits by default for N >= 4"""
    r = run_bench("--gpus", "8", "--share-gpu", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--log-n", "18", "--width", "64", timeout=1500)
    assert r["n_gpus"] == 8 and r["rccl_world_size"] == 8 and r["share_gpu_test_mode"] is True and r["collective_backend"] == "gloo"
    assert r["shards_proven"] == 16 and r["distinct_shards_proven"] == 16 and r["shard_digests_gathered"] == 16
    assert r["host_wait"] == "block" and r["verified"] is True and r["value"] > 0
