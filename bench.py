#!/usr/bin/env python3
"""Contract benchmark: whole-shard STARK proofs on MI355X through libzkhip's C ABI.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU over RCCL.  Under torch.distributed.run the ranks are the launcher's; started
   plainly, bench.py spawns its N rank processes itself BEFORE anything in the parent touches the GPU)

One "step" = one complete shard proof (commit trace -> quotient -> openings -> FRI ->
PoW -> queries) of a synthetic SP1-core-like shard: 2^20 rows x 256 columns, log_blowup 1,
100 queries, 16 PoW bits (BASELINE.json configs[1] / SURVEY.md 8d).  The trace is resident
in HBM before the timed region.  Shards are independent: the K*N shards of the job are dealt round-robin
(zktls_amd.shards.shard_indices == the library's zkhip_shard_device): rank r proves shards r, r+N, ...
(weak scaling, no data-path collective); RCCL broadcasts the 8-word batch transcript seed and, after
the timed region, gathers the proof digests (every timed step proves a different shard, up to 48 per rank;
beyond that shards repeat and `distinct_shards_proven` says so).
Up to --streams shards are in flight per GPU, each on its own context + HIP stream + host
thread, so the latency-bound stretches of one proof hide under the kernels of the others.

Prints ONE JSON line (rank 0): metric trace-cells/s (+ proofs/s), `roofline` for the NTT
pass kernel (HIP events on the launch stream; the LDE launches of a proof -- first inverse pass, the fused
middle launch, second forward pass per coset -- on the proving context's OWN workspaces, i.e. the in-proof
buffer placement), `valu_roofline` for the Poseidon2 leaf kernel
(the largest share of a proof, integer-multiply bound) and `cpu_baseline` (the CPU oracle
timed on the host cores, bounded sample, N = 1 only).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x5A4B544C53
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def _spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes.  Nothing in this (parent)
    process has touched torch.cuda or HIP, and it never does: it only waits.  Rank 0 inherits stdout and prints the JSON line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for pr in procs:
        code = pr.wait()
        rc = rc or code
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--logup-pairs", type=int, default=0, help="LogUp lookup pairs (SURVEY 8a row a8); 0 = main AIR only")
    ap.add_argument("--shape", choices=("sp1", "r0"), default="sp1",
                    help="proof-system shape: sp1 = the headline (blowup 2, 100 queries, 16 PoW bits, fold by 2, Poseidon2-16); "
                         "r0 = RISC-Zero-like (blowup 4, 50 queries, fold by 16, 256 final coefficients, Poseidon2-24)")
    ap.add_argument("--chips", default="", help="prove a shard of several chips instead of one matrix: 'LOGNxW,LOGNxW,...' tallest first, "
                                                "or 'sp1like' = 20x96,20x32,19x64,18x128,16x256,14x40 (not the headline workload)")
    ap.add_argument("--host-traces", action="store_true", help="traces start in (pinned) HOST memory: every step includes the 1 GiB H2D copy "
                                                               "(the PCIe-inclusive rate quoted in DESIGN.md; never the headline value)")
    ap.add_argument("--streams", type=int, default=4, help="shards in flight per GPU (at most --steps): each on its own context + HIP stream")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST MODE for boxes with fewer GPUs than ranks: rank r uses device r mod (visible devices) and the collectives run "
                         "over gloo (RCCL refuses two ranks on one GPU).  Exercises the N > 1 code path end to end; the line says so and is not a scaling result")
    ap.add_argument("--force-collective", action="store_true",
                    help="with --gpus 1: initialise the RCCL ('nccl') process group at world size 1 and run the seed broadcast, the MAX "
                         "all-reduce and a barrier through it, so that RCCL is loaded and called on hardware even on a one-GPU box")
    ap.add_argument("--one-process", action="store_true",
                    help="ONE process deals the N x K shards over the N devices through the library's own multi-GPU entry "
                         "(zkhip_prove_shards_multi(NULL, 0, ...): shard s on device s mod N, traces generated where zkhip_shard_device puts them, "
                         "--streams pooled contexts per device) instead of one rank per GPU; same JSON line, parallelism 'one process, device list'")
    ap.add_argument("--logical-devices", type=int, default=0,
                    help="TEST MODE with --one-process on a box with fewer GPUs than --gpus: load the A/B build (libzkhip_ab.so) with "
                         "ZKHIP_LOGICAL_DEVICES=K, whose device ordinals 0..K-1 are logical devices on the physical ones (own pools, own workers, "
                         "traces checked to live where their shard is dealt); the line says so and is not a scaling result")
    ap.add_argument("--no-execution", action="store_true", help="skip the reference-sized execution (22 shards: core, then core + compress) through the host mirror of ZkProver::prove")
    ap.add_argument("--no-multichip", action="store_true", help="skip the multi-chip shard with LogUp pairs (SP1's shard structure, rows a7 mixed heights + a8) measured beside the headline")
    ap.add_argument("--no-fri-graph", action="store_true",
                    help="zkhip_set_fri_graph(0): the FRI commit phase as plain launches instead of one hipGraphLaunch per proof (a debugging switch; "
                         "same proof bytes; profiles/r04_segv.md)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--wait", choices=("auto", "poll", "block"), default="auto",
                    help="how the prover's host threads wait for their streams (zkhip_set_wait_mode): poll = hipStreamSynchronize, block = sleep "
                         "on an event; auto = block when the ranks' waiting threads outnumber the cores this container may use")
    ap.add_argument("--no-recursion16", action="store_true", help="skip the 16 recursion proofs (the FRI check of 16 shard proofs of the headline shape proven in-circuit, one call) measured beside the headline")
    ap.add_argument("--no-batch64", action="store_true", help="skip the 64-transcript batch (BASELINE configs[2]) measured beside the headline")
    ap.add_argument("--cpu-log-n", type=int, default=20, help="rows of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the CPU baseline (0: min(cores, 64))")
    args = ap.parse_args()

    if args.logical_devices:
        if not args.one_process:
            raise SystemExit("--logical-devices goes with --one-process")
        os.environ["ZKHIP_LOGICAL_DEVICES"] = str(args.logical_devices)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import _ab  # noqa: F401  (points the binding at libzkhip_ab.so; nothing has loaded the library yet)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.one_process:
        _spawn_ranks(args.gpus)             # never returns; no GPU call has happened in this process

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    one_proc = args.one_process
    if one_proc:
        if world != 1:
            raise SystemExit("--one-process is ONE process: start it plainly, not under a launcher")
        n_dev = args.gpus                   # devices of the job (the library's device list); `world` stays 1: no ranks, no collective
    elif world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))

    def usable_cores():
        """cores this process may keep busy: the cgroup CPU quota when there is one, else the affinity mask"""
        n = len(os.sched_getaffinity(0))
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                n = min(n, max(1, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
        return n
    # How the prover's host threads wait for the GPU.  The runtime's default polls: ~1.2 busy cores per shard in flight (measured: 4.7 per
    # rank at four in flight).  When all ranks of this node together would need more cores than the container may use, polling threads
    # get throttled together with the ones that prepare launches: then the waits sleep (hipDeviceScheduleBlockingSync, set through the
    # library BEFORE anything initialises the device -- torch.cuda.is_available() below does).
    cores_ok = usable_cores()
    # Host-thread budget: a rank with four shards in flight has 4 proving threads + the Python main thread; polling, each
    # proving thread keeps ~1.2 cores busy.  Eight ranks x four in flight = 32 proving threads = ~38 cores' worth of polling under the
    # 16-core quota of the boxes of this pool: from four GPUs on the waits sleep by default (1.0 core per rank, same ms per step at N = 1).
    gpus_of_node = args.gpus if one_proc else world
    wait_block = args.wait == "block" or (args.wait == "auto" and (gpus_of_node >= 4 or gpus_of_node * (min(args.streams, args.steps) * 1.2 + 1.0) > cores_ok))
    from zktls_amd import _lib as zk_lib
    n_vis = torch.cuda.device_count()                    # (counting devices does not initialise them)
    wait_rc = zk_lib.load().zkhip_set_wait_mode(1 if wait_block else 0, -1 if one_proc else ((local_rank % max(n_vis, 1)) if args.share_gpu else local_rank))
    if wait_rc != 0 and wait_block:
        sys.stderr.write("bench.py: blocking waits could not be set (%d): polling\n" % wait_rc)
        wait_block = False
    if args.no_fri_graph:
        zk_lib.load().zkhip_set_fri_graph(0)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libzkhip has no CPU path")
    if args.share_gpu:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    elif args.force_collective:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    coll_dev = "cpu" if args.share_gpu else "cuda"
    rccl_calls = 0
    if dist is not None and not args.share_gpu:
        # RCCL on hardware, whatever the world size: one broadcast, one MAX all-reduce, one barrier through the 'nccl' backend
        t_ = torch.tensor([rank + 7], dtype=torch.int32, device="cuda")
        dist.broadcast(t_, src=0)
        a_ = torch.tensor([float(rank)], dtype=torch.float64, device="cuda")
        dist.all_reduce(a_, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()
        if int(t_.item()) != 7 or float(a_.item()) != float(world - 1):
            raise SystemExit("RCCL self-check failed: broadcast %d, max %f" % (int(t_.item()), float(a_.item())))
        rccl_calls = 3

    from zktls_amd._lib import Params
    from zktls_amd.device import Context, verify_shard

    log_n, width = args.log_n, args.width
    chip_list = None
    if args.chips:
        spec = "20x96,20x32,19x64,18x128,16x256,14x40" if args.chips == "sp1like" else args.chips
        chip_list = [tuple(int(v) for v in c.split("x")) for c in spec.split(",")]
    n = 1 << log_n
    cells = n * width if chip_list is None else sum(w << ln for ln, w in chip_list)
    LQ = args.logup_pairs
    prm = Params(1, 100, 16, LQ) if args.shape == "sp1" else Params(2, 50, 0, LQ, 4, 8, 24)
    # one context (= one HIP stream + its workspaces) per shard in flight: while one shard sits in a
    # latency-bound stretch (small FRI layers, host round trips) the other keeps the CUs busy
    S = max(1, min(args.streams, max(args.steps, 1)))
    in_flight = S                          # shards in flight per GPU (one-process mode: the library's pooled contexts per device)
    if one_proc:
        S = 1                              # this process's own context serves the roofline sections only
    streams = [torch.cuda.Stream(device=local_rank) for _ in range(S)]
    ctxs = [Context(local_rank, stream=st.cuda_stream) for st in streams]
    stream, ctx = streams[0], ctxs[0]

    # batch transcript seed: rank 0 draws it, RCCL broadcasts it (the only collective)
    from zktls_amd import shards
    public = shards.broadcast_seed(dist, [(SEED >> (8 * i)) & 0xFF for i in range(8)], device=coll_dev)

    K, W = args.steps, args.warmup
    nbuf = min(max(K, 1), 48)      # every timed step proves a DIFFERENT shard up to 48 steps (48 GiB of traces at the headline shape)
    # this rank's shards of the K * world-shard job, dealt by the tested scheduling function (round-robin)
    my = shards.shard_indices(max(K, 1) * world, rank, world)
    assert len(my) == max(K, 1)
    op = None
    if one_proc:
        # ---- ONE process, the library's device list (what a ZkProver::prove binds on a multi-GPU node).  Shard s of the
        # N x K-shard job is proven on device zkhip_shard_device(s) = s mod N; its trace is generated THERE, through a context of that device.
        if chip_list is not None or args.host_traces or LQ:
            raise SystemExit("--one-process measures the headline shard (no --chips / --host-traces / --logup-pairs)")
        from zktls_amd.device import prove_shards_multi, shard_device
        vis = zk_lib.device_count()
        if n_dev > vis:
            raise SystemExit("--gpus %d but %d device(s) visible%s" % (n_dev, vis, "" if args.logical_devices else " (--logical-devices K runs the path on fewer GPUs, test mode)"))
        dev_list = None if n_dev == vis else list(range(n_dev))     # NULL, 0 = every visible device
        dev_ctx = {0: ctx}
        for d in range(1, n_dev):
            dev_ctx[d] = Context(d)
        job_shards = list(range(max(K, 1) * n_dev))
        op_traces = {}
        for s_ in job_shards:
            d = shard_device(s_, dev_list, n_dev)
            slot = (s_ // n_dev) % nbuf
            if (d, slot) not in op_traces:
                op_traces[(d, slot)] = (dev_ctx[d].gen_trace(SEED, s_, log_n, width), s_)
        for c in dev_ctx.values():
            c.sync()
        bufs = [op_traces[(0, k)][0] for k in range(nbuf)]
        my = [op_traces[(0, k)][1] for k in range(nbuf)]
        with torch.cuda.stream(stream):
            roof_scratch = [torch.empty(n * width, dtype=torch.int32, device="cuda") for _k in range(4)]
        op = {"dev_list": dev_list, "shards": job_shards}

        def op_run(count_per_dev):
            ids = list(range(count_per_dev * n_dev))
            trs = [op_traces[(shard_device(s_, dev_list, n_dev), (s_ // n_dev) % nbuf)] for s_ in ids]
            return trs, prove_shards_multi([t for t, _ in trs], log_n, width, [public + [sid] for _, sid in trs], prm, devices=dev_list, in_flight=in_flight)
    elif chip_list is None:
        with torch.cuda.stream(stream):
            traces = [torch.empty(cells, dtype=torch.int32, device="cuda") for _ in range(nbuf)]
        bufs = [ctx.wrap(t) for t in traces]
        with torch.cuda.stream(stream):
            # destinations of the stand-alone strided-pass scan of the roofline section (context only: the roofline itself is
            # measured on the proving context's own workspaces).  The strided pass runs in one of two modes depending on where
            # source and destination lie; the scan reports the spread over 16 pairs.
            roof_scratch, spacers = [], []
            for _k in range(4):          # 2, 6, 10 GiB apart: buffer classes come in runs of several GiB, neighbours share one
                roof_scratch.append(torch.empty(n * width, dtype=torch.int32, device="cuda"))
                if _k < 3:
                    spacers.append(torch.empty((2 + 4 * _k) * n * width, dtype=torch.int32, device="cuda"))
            del spacers
        for i, b in enumerate(bufs):
            if LQ:
                ctx.gen_trace_logup(SEED, my[i], log_n, width, LQ, out=b)
            else:
                ctx.gen_trace(SEED, my[i], log_n, width, out=b)
    else:
        nbuf = 1
        chip_bufs = [(ctx.gen_trace(SEED, 100 * rank + j, ln, w), ln, w) for j, (ln, w) in enumerate(chip_list)]
        with torch.cuda.stream(stream):
            traces = [torch.empty(n * width, dtype=torch.int32, device="cuda")]      # source of the roofline section only
            roof_scratch, spacers = [], []
            for _k in range(4):          # 2, 6, 10 GiB apart: buffer classes come in runs of several GiB, neighbours share one
                roof_scratch.append(torch.empty(n * width, dtype=torch.int32, device="cuda"))
                if _k < 3:
                    spacers.append(torch.empty((2 + 4 * _k) * n * width, dtype=torch.int32, device="cuda"))
            del spacers
        bufs = [ctx.wrap(traces[0])]
        ctx.fill_uniform(SEED, log_n, width, out=bufs[0])
    ctx.sync()

    host_traces = None
    if args.host_traces and chip_list is None:
        host_traces = []
        for b in bufs:
            h = torch.empty(cells, dtype=torch.int32).pin_memory()
            ctx.lib.zkhip_from_monty(ctx.handle, __import__("ctypes").c_void_p(b.ptr), __import__("ctypes").c_void_p(b.ptr), cells)     # canonical words on the host side
            ctx.lib.zkhip_memcpy_d2h(ctx.handle, __import__("ctypes").c_void_p(h.data_ptr()), __import__("ctypes").c_void_p(b.ptr), cells * 4)
            host_traces.append(h)
        ctx.sync()

    def step(i, c=None):
        if host_traces is not None:
            return (c or ctx).prove_shard_host(None, public + [my[i % nbuf]], prm, host_ptr=host_traces[i % nbuf].data_ptr(), log_n=log_n, width=width)
        if chip_list is not None:
            return (c or ctx).prove_chips(chip_bufs, public + [my[i % len(my)]], prm)
        return (c or ctx).prove_shard(bufs[i % nbuf], log_n, width, public + [my[i % nbuf]], prm)

    def run_steps(count, static=False):
        """`count` shard proofs, S at a time (one host thread per context; ctypes drops the GIL).
        static: worker w takes items w, w + S, ... (warm-up: every context gets work); otherwise first come first served"""
        if S == 1:
            return [step(i) for i in range(count)]
        import threading
        results = [None] * count
        errors = []

        nxt = [0]
        lock = threading.Lock()

        def worker(w):
            try:
                if static:
                    for i in range(w, count, S):
                        results[i] = step(i, ctxs[w])
                    return
                while True:
                    with lock:              # shards are handed out as contexts become free
                        i = nxt[0]
                        nxt[0] += 1
                    if i >= count:
                        return
                    results[i] = step(i, ctxs[w])
            except Exception as e:          # surfaced after the join
                errors.append(e)
        ts = [threading.Thread(target=worker, args=(w,)) for w in range(min(S, count))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errors:
            raise errors[0]
        return results

    if one_proc:
        op_run(max(W, in_flight if W else 0))            # warm every pooled context of every device
    else:
        run_steps(max(W, S if W else 0), static=True)       # warm every context (plans, workspaces)
    for c in ctxs:
        c.sync()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The interpreter's cyclic collector stays out of every timed region of this script (as timeit keeps it out): a full collection over
    # torch's heap holds the GIL for tens of milliseconds, and a prover thread that finishes a shard meanwhile cannot pick up the next one --
    # measured on one box, same library: 14.0 - 14.3 ms per step when no collection fell into the ten steps, 15 - 18 ms when one did (which
    # of the two a run got depended on the allocation count at start-up: bytecode cache or not, one more ctypes binding or not).
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    cpu0 = time.process_time()                          # CPU time of all threads of this rank
    if one_proc:
        op_trs, proofs = op_run(K)                         # ONE call: N x K shards over the device list
        for c in dev_ctx.values():
            c.sync()
        elapsed = time.perf_counter() - t0
        host_cores_busy = (time.process_time() - cpu0) / elapsed
        import hashlib
        digests = {sid: hashlib.sha256(p_.tobytes()).hexdigest() for (_, sid), p_ in zip(op_trs, proofs)}
        if len(set(digests.values())) != len(digests) or len(digests) != min(K, nbuf) * n_dev:
            raise SystemExit("shard coverage broken: %d distinct proofs for %d distinct shards" % (len(set(digests.values())), min(K, nbuf) * n_dev))
        last = proofs[-1]
        last_public = public + [op_trs[-1][1]]
        world_eff = n_dev
        del proofs
    else:
        proofs = run_steps(K)
        barrier()
        elapsed = time.perf_counter() - t0
        host_cores_busy = (time.process_time() - cpu0) / elapsed
        elapsed = shards.max_over_ranks(dist, elapsed, device=coll_dev)
        last = proofs[K - 1]
        last_public = public + [my[(K - 1) % nbuf]] if chip_list is None else public + [my[(K - 1) % len(my)]]
        world_eff = world
        # after the timed region: every distinct shard of the job was proven on exactly one rank (digest gather over RCCL)
        # (keyed by the shard actually proven: step i proves shard my[i % nbuf], so beyond nbuf steps shards repeat and only the distinct ones count)
        distinct = min(K, nbuf) if chip_list is None else K
        digests = shards.gather_proof_digests(dist, {(my[i % nbuf] if chip_list is None else my[i]): proofs[i] for i in range(K)})
        expect = sorted(s_ for r_ in range(world) for s_ in shards.shard_indices(max(K, 1) * world, r_, world)[:distinct])
        if sorted(digests) != expect:
            raise SystemExit("shard coverage broken: %d digests for %d distinct shards" % (len(digests), len(expect)))
        del proofs

    # the library's own multi-GPU entry on the same shards (one process, device list [this GPU]): the path a ZkProver::prove binds
    # (INTEGRATION.md 2), measured beside the rank-per-GPU path whenever this rank is alone on the node
    one_process = None
    if not one_proc and world == 1 and chip_list is None and host_traces is None and LQ == 0:
        from zktls_amd.device import prove_shards_multi
        ids = [i % nbuf for i in range(K)]
        prove_shards_multi([bufs[i] for i in ids[:max(S, 1)]], log_n, width, [public + [my[i]] for i in ids[:max(S, 1)]], prm, devices=None if zk_lib.device_count() == 1 else [local_rank], in_flight=S)
        to0 = time.perf_counter()
        pm = prove_shards_multi([bufs[i] for i in ids], log_n, width, [public + [my[i]] for i in ids], prm, devices=None if zk_lib.device_count() == 1 else [local_rank], in_flight=S)
        dto = time.perf_counter() - to0
        one_process = {"entry": "zkhip_prove_shards_multi(NULL, 0, ...), %d pooled contexts per device" % S, "ms_per_step": round(dto / K * 1e3, 3),
                       "value": round(cells * K / dto, 1), "shards_proven": K, "distinct_proofs": len({p_.tobytes() for p_ in pm}),
                       "same_bytes_as_the_timed_step": bool(pm[K - 1].tobytes() == last.tobytes())}
        del pm
        zk_lib.load().zkhip_release_cached_contexts()

    # single-shard latency (one shard in flight, nothing else on the GPU): NOT the metric -- `value` is throughput with S shards in flight
    latency_ms = None
    if rank == 0 and chip_list is None and host_traces is None:
        ctx.sync()
        step(0)
        tl = time.perf_counter()
        for i in range(2):
            step(i)
        ctx.sync()
        latency_ms = (time.perf_counter() - tl) / 2 * 1e3

    # the last proof of the timed region must verify (host verifier of the product); its second run is timed: the reference verifies
    # every proof on the CPU inside `prove` (sp1.rs:120), outside the span it times
    tv = time.perf_counter()
    if chip_list is None:
        rc, reason = verify_shard(last, log_n, width, last_public, prm)
        tv = time.perf_counter()
        rc, reason = verify_shard(last, log_n, width, last_public, prm)
    else:
        from zktls_amd.device import verify_chips
        rc, reason = verify_chips(last, [c[0] for c in chip_list], [c[1] for c in chip_list], last_public, prm)
    verified = rc == 0
    host_verify_ms = (time.perf_counter() - tv) * 1e3

    # ---- roofline of the NTT pass kernel: one launch = 8 B/element (read 4 + write 4).
    # Measured on the IN-PROOF placement: the four launches of the trace LDE (inverse strided / inverse contiguous / forward
    # block-in strided-out / forward contiguous) exactly as zkhip_prove_shard enqueues them, on context 0's own coefficient and
    # LDE workspaces (zkhip_ntt_pass which = 2..5).  `achieved` = algorithmic bytes / mean launch time over the six launches of one
    # LDE (I1, I2, 2 x F1, 2 x F2).  Nothing is selected: the placement is whatever the proving context got.  A scan of the
    # stand-alone strided pass over 16 (source, destination) buffer pairs is reported beside it (min / median / max) for context.
    roof = None
    if rank == 0 and log_n > 20:
        # 2^21 / 2^22 rows: the 2^20-row machinery on R = 2 / 4 row classes plus one streaming radix-R pass either way
        # (ntt_combine_kernel).  No single-launch hook at this size: the whole trace LDE is timed (HIP events, own output buffer)
        # and priced against the bytes it must move.  With the fused middle launch (width a multiple of 32):
        # 8 (inverse radix-R pass) + 8 (I1) + 12 (fused) + 2 x 8 (F2) + 2 x 8 (forward radix-R pass) = 60 B per trace cell.
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            big_out = torch.empty(2 * n * width, dtype=torch.int32, device="cuda")
        bo = ctx.wrap(big_out)
        for _ in range(3):
            ctx.coset_lde(bufs[0], log_n, width, out=bo)
        e0.record(stream)
        for _ in range(20):
            ctx.coset_lde(bufs[0], log_n, width, out=bo)
        e1.record(stream)
        e1.synchronize()
        lde_ms = e0.elapsed_time(e1) / 20
        bpc = 60.0 if width % 32 == 0 else 72.0
        ach = bpc * n * width / lde_ms / 1e6
        roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": None, "kernel": "the whole 2^%d x %d trace LDE (zk::ntt_combine_kernel + zk::ntt_pass_kernel + zk::lde_fused_kernel launches), %g B per trace cell" % (log_n, width, bpc),
                "algorithmic_bytes_per_launch": bpc * n * width, "avg_launch_ms": round(lde_ms, 4),
                "lde": {"ms": round(lde_ms, 4), "bytes_per_trace_cell": bpc, "GB/s": round(ach, 1)}}
        del big_out
    if rank == 0 and log_n <= 20:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timed(fn, reps, warm=3):
            for _ in range(warm):
                fn()
            e0.record(stream)
            for _ in range(reps):
                fn()
            e1.record(stream)
            e1.synchronize()
            return e0.elapsed_time(e1) / reps
        for _ in range(300):                      # settle clocks (with the measured pass itself: its launches share one kernel name in a profile)
            ctx.ntt_pass(bufs[0], None, log_n, width, 6 if log_n == 20 and width % 32 == 0 else 2)
        reps = 1000
        # kernel names as a rocprofv3 summary of this command lists them: the launches of this section run under their own
        # template tag, so the profile keeps the isolated launches apart from the in-proof ones, which overlap with the other
        # shards in flight
        fused_lde = log_n == 20 and width % 32 == 0
        alg_bytes = 8.0 * n * width

        def i1_over_traces(which):
            """the LDE's first pass reads the shard's trace: its duration depends on where THAT trace lies relative to the context's
            workspace (0.46 ... 0.52 ms over the pairs of one process, tools/archive/i1_sources.py) -- timed over the traces of all timed shards,
            the same number of launches each, mean reported (and the spread beside it); nothing is selected"""
            per = max(10, reps // len(bufs))
            ms = [timed(lambda b=b: ctx.ntt_pass(b, None, log_n, width, which), per) for b in bufs]
            srt = sorted(ms)
            return sum(ms) / len(ms), {"traces": len(ms), "launches_each": per, "min_ms": round(srt[0], 4), "median_ms": round(srt[len(srt) // 2], 4), "max_ms": round(srt[-1], 4)}
        fused_info = None
        # the LDE of THIS proof shape: 2^log_blowup cosets (sp1: 2, r0: 4).  Fused (2^20 rows x a multiple of 32 columns): I1 once, one
        # fused middle launch per PAIR of cosets (read 4 B + write 8 B per cell), F2 per coset:
        #     bytes per trace cell = 8 + (cosets / 2) x 12 + cosets x 8     (sp1: 36, r0: 64)
        # unfused: I1 + I2 once, F1 + F2 per coset = 16 + cosets x 16      (sp1: 48, r0: 80)
        cosets = 1 << prm.log_blowup
        if fused_lde:
            # The roofline kernel is the pass kernel: its 1 + cosets launches of the LDE (I1, F2 per coset); the fused launch is bound by
            # its butterflies, not by HBM, and is reported beside it together with the whole LDE (`lde`, `lde_frac`).
            names = {6: "zk::ntt_pass_kernel<4,true,2,5,4>, LDE pass I1 (inverse, strided in -> one contiguous block per tile; mean over the traces of all timed shards)",
                     5: "zk::ntt_pass_kernel<4,false,2,5,3>, LDE pass F2 (forward, contiguous, in place)"}
            in_proof = {w: timed(lambda w=w: ctx.ntt_pass(bufs[0], None, log_n, width, w), reps) for w in (7, 5)}
            in_proof[6], i1_spread = i1_over_traces(6)
            avg_ms = (in_proof[6] + cosets * in_proof[5]) / (1.0 + cosets)
            launches_note = "mean over the %d pass-kernel launches of one 2^%d x %d trace LDE (I1, %d x F2)" % (1 + cosets, log_n, width, cosets)
            fb = 12.0 * n * width
            n_fused = cosets // 2
            lde_bpc = 8 + n_fused * 12 + cosets * 8
            lde_ms = in_proof[6] + n_fused * in_proof[7] + cosets * in_proof[5]
            # (the unfused launches for comparison: few of them -- the strided -> strided first pass runs under the same kernel name as I1)
            unfused = {w: timed(lambda w=w: ctx.ntt_pass(bufs[0], None, log_n, width, w), 20) for w in (2, 3, 4)}
            unfused_ms = unfused[2] + unfused[3] + cosets * unfused[4] + cosets * in_proof[5]
            fused_info = {"kernel": "zk::lde_fused_kernel<1>, second inverse pass + first forward pass of two cosets in one launch (%d per LDE)" % n_fused,
                          "ms": round(in_proof[7], 4), "algorithmic_bytes_per_launch": fb, "GB/s": round(fb / in_proof[7] / 1e6, 1),
                          "frac_of_hbm_peak": round(fb / in_proof[7] / 1e6 / HBM_PEAK_GBS, 4),
                          "bound": "integer VALU: three 1024-point tile transforms per 12 B (DESIGN.md 4.1); it replaces three pass launches of 8 B per cell each",
                          "replaces_ms": round(unfused[3] + 2 * unfused[4], 4)}
            lde_info = {"launches": "I1 + %s fused + %d x F2" % ("" if n_fused == 1 else "%d x" % n_fused, cosets), "cosets": cosets, "ms": round(lde_ms, 4),
                        "bytes_per_trace_cell": lde_bpc, "bytes_formula": "8 + (cosets / 2) x 12 + cosets x 8",
                        "GB/s": round(float(lde_bpc) * n * width / lde_ms / 1e6, 1), "unfused_launches_ms": round(unfused_ms, 4),
                        "unfused_bytes_per_trace_cell": 16 + 16 * cosets,
                        "information_minimum_bytes_per_trace_cell": 4 + 4 * cosets}
            detail = (6, 5)
        else:
            names = {2: "zk::ntt_pass_kernel<4,true,2,5,4>, LDE pass I1 (inverse, strided in -> strided out)",
                     3: "zk::ntt_pass_kernel<4,true,2,5,3>, LDE pass I2 (inverse, contiguous, in place)",
                     4: "zk::ntt_pass_kernel<4,false,2,5,4>, LDE pass F1 (forward, block in -> strided bit-reversed out)",
                     5: "zk::ntt_pass_kernel<4,false,2,5,3>, LDE pass F2 (forward, contiguous, in place)"}
            in_proof = {w: timed(lambda w=w: ctx.ntt_pass(bufs[0], None, log_n, width, w), reps) for w in (3, 4, 5)}
            in_proof[2], i1_spread = i1_over_traces(2)
            avg_ms = (in_proof[2] + in_proof[3] + cosets * in_proof[4] + cosets * in_proof[5]) / (2.0 + 2 * cosets)
            launches_note = "mean over the %d launches of one 2^%d x %d trace LDE" % (2 + 2 * cosets, log_n, width)
            lde_ms = in_proof[2] + in_proof[3] + cosets * in_proof[4] + cosets * in_proof[5]
            lde_bpc = 16 + 16 * cosets
            lde_info = {"launches": "I1 + I2 + %d x (F1 + F2)" % cosets, "cosets": cosets, "ms": round(lde_ms, 4), "bytes_per_trace_cell": lde_bpc,
                        "bytes_formula": "16 + cosets x 16", "GB/s": round(float(lde_bpc) * n * width / lde_ms / 1e6, 1),
                        "information_minimum_bytes_per_trace_cell": 4 + 4 * cosets}
            detail = (2, 3, 4, 5)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        # stand-alone passes on caller buffers, every (source, destination) pair of up to four traces and four scratch buffers
        cands = [ctx.wrap(t) for t in roof_scratch]
        srcs = bufs[:4]
        placements = sorted(timed(lambda sb=sb, cb=cb: ctx.ntt_pass(sb, cb, log_n, width, 0), 100) for sb in srcs for cb in cands)
        contiguous_ms = timed(lambda: ctx.ntt_pass(srcs[0], cands[0], log_n, width, 1), 500)
        med = placements[len(placements) // 2]
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_ntt_pass.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic = pj.get("hbm_bytes_per_launch")
                traffic_src = "from profiles/pmc_ntt_pass.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, %s), not measured in this run" % pj.get("measured_at", "round 1")
            except Exception:
                traffic = None

        def gbs(ms):
            return round(alg_bytes / ms / 1e6, 1)
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                # also against the measured float4-copy ceiling of this part (MI355X_MICROARCH.md: 6.29 TB/s)
                "frac_of_measured_copy_ceiling": round(achieved / 6290.0, 4),
                # the WHOLE LDE against the peak: every byte its launches move (lde.bytes_per_trace_cell) over the sum of their durations --
                # `frac` above covers the pass kernel only (the fused middle launch is VALU-bound and excluded from it)
                "lde_frac": round(lde_info["GB/s"] / HBM_PEAK_GBS, 4),
                "lde_frac_of_information_minimum": round(lde_info["information_minimum_bytes_per_trace_cell"] * n * width / lde_ms / 1e6 / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "zk::ntt_pass_kernel, %s on the proving context's own workspaces (in-proof placement, nothing selected)" % launches_note,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(avg_ms, 4),
                "kernels": {names[w]: {"ms": round(in_proof[w], 4), "GB/s": gbs(in_proof[w]), "frac": round(gbs(in_proof[w]) / HBM_PEAK_GBS, 4)} for w in detail},
                "first_pass_over_the_timed_shards_traces": i1_spread,
                "fused_middle_launch": fused_info, "lde": lde_info,
                "standalone_strided_pass_by_placement": {"pairs": len(placements), "min_ms": round(placements[0], 4), "median_ms": round(med, 4),
                                                         "max_ms": round(placements[-1], 4), "median_frac": round(gbs(med) / HBM_PEAK_GBS, 4),
                                                         "best_placement_frac": round(gbs(placements[0]) / HBM_PEAK_GBS, 4)},
                "standalone_contiguous_pass": {"ms": round(contiguous_ms, 4), "frac": round(gbs(contiguous_ms) / HBM_PEAK_GBS, 4)}}

    # ---- the kernel that takes most of a proof's time is not HBM-shaped: Poseidon2 leaf hashing, priced against the
    # measured integer-multiply roof (v_mul_lo / v_mad_u64_u32 issue at 4.2 clk per wave64 per SIMD at the
    # nominal 2.4 GHz -> 256 CUs x 4 SIMDs x 64 / 4.2 x 2.4e9 = 37.4 T int-mul/s; tools/microbench.hip)
    valu = None
    if rank == 0 and args.shape == "sp1":
        rows = 2 * n
        with torch.cuda.stream(stream):
            lde_t = torch.empty(rows * width, dtype=torch.int32, device="cuda")
            dig_t = torch.empty(rows * 8, dtype=torch.int32, device="cuda")
        lde_b, dig_b = ctx.wrap(lde_t), ctx.wrap(dig_t)
        ctx.fill_uniform(SEED + 99, log_n + 1, width, out=lde_b)          # uniform field elements, like a real LDE
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ctx.hash_rows([(lde_b, width)], rows, out=dig_b)
        reps_h = 5
        e0.record(stream)
        for _ in range(reps_h):
            ctx.hash_rows([(lde_b, width)], rows, out=dig_b)
        e1.record(stream)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps_h
        perms = rows * ((width + 7) // 8)
        # integer multiplies of one permutation as implemented (poseidon2.cuh): 141 S-boxes x 4 Montgomery products x 3,
        # 13 internal layers x (16 for the row sum + 2 for its reduction + 1 + 16 x 3)
        imul_per_perm = 141 * 4 * 3 + 13 * (16 + 2 + 1 + 16 * 3)
        ach = perms * imul_per_perm / (ms * 1e-3) / 1e12
        valu = {"bound": "int-mul (not in the roofline schema: reported beside it)", "kernel": "zk::hash_rows_vec_kernel, 2^%d x %d leaves" % (log_n + 1, width),
                "ms": round(ms, 3), "perm_per_s": round(perms / (ms * 1e-3), 1), "int_mul_per_perm": imul_per_perm,
                "achieved": round(ach, 2), "peak": 37.4, "unit": "T int-mul/s", "frac": round(ach / 37.4, 4),
                "hbm_GB/s": round((4.0 * rows * width + 32.0 * rows) / (ms * 1e-3) / 1e9, 1)}
        del lde_t, dig_t

    # ---- BASELINE configs[2]: sixty-four 13 KB transcripts in ONE call, each a keyed SHA-256 machine proof (padding, trace, range table,
    # proof; host bytes in, proofs out), in lock-step batches (csrc/batch.h) and, beside it, one context + stream per worker
    batch64 = None
    if rank == 0 and not args.no_batch64 and chip_list is None:
        import hashlib
        from zktls_amd.device import lockstep_stats, prove_transcripts, set_lockstep, verify_sha256_machine
        base = open(os.path.join(ROOT, "tests", "golden", "reference", "guest_input0.cbor"), "rb").read()
        msgs = [base + i.to_bytes(4, "little") for i in range(64)]
        tprm = Params(1, 100, 16)

        def best_of(reps, **kw):
            best, res = 1e9, None
            for _ in range(reps):
                tb0 = time.perf_counter()
                res = prove_transcripts(msgs, tprm, devices=[local_rank], **kw)
                best = min(best, time.perf_counter() - tb0)
            return best, res

        set_lockstep(0)
        best_of(1, in_flight=16)
        t_plain, (vk_p, res_p) = best_of(3, in_flight=16)
        set_lockstep(16, 6)
        best_of(1)
        st0 = lockstep_stats()
        t_lock, (vk_l, res_l) = best_of(3)
        st1 = lockstep_stats()
        t_lock_v, _ = best_of(2, verify=True)
        same = vk_p.tolist() == vk_l.tolist() and all(a[0] == b[0] and a[1].tobytes() == b[1].tobytes() for a, b in zip(res_p, res_l))
        ok64 = all(d == hashlib.sha256(m).digest() for m, (d, _) in zip(msgs, res_l)) and verify_sha256_machine(res_l[63][1], res_l[63][0], vk_l, tprm, len(msgs[63])) == (0, 0)
        batch64 = {"workload": "64 transcripts of %d bytes, one zkhip_prove_transcripts call, keyed SHA-256 machine (chip 2^14 x 640 + range table 2^16), log_blowup 1, 100 queries, 16 PoW bits" % len(msgs[0]),
                   "ms": round(t_lock * 1e3, 2), "transcripts_per_s": round(64 / t_lock, 1), "mode": "lock-step: 6 lanes x up to 16 members, launches merged",
                   "ms_with_verify_inside": round(t_lock_v * 1e3, 2),
                   "merged_launches_per_call": (st1[0] - st0[0]) // 3, "member_launch_requests_per_call": (st1[1] - st0[1]) // 3,
                   "one_stream_per_worker_ms": round(t_plain * 1e3, 2), "one_stream_per_worker_in_flight": 16,
                   "same_bytes_both_ways": bool(same), "digests_and_last_proof_verified": bool(ok64),
                   "proof_bytes": int(res_l[0][1].size)}
        # ... and the 64 statements as ONE proof: each message proven as zkhip_prove_sha256 does (zkhip_prove_transcripts_air: a version-7 proof of the chip's constraint program, the
        # same 100 queries / 16 PoW bits), the 64 proofs verified in-circuit by one zkhip_prove_shard_verifier_air call (nine chips: the EVAL
        # chip evaluates the SHA-256 program's 5 192 terms per proof).  The outer proof's verifier takes the program, 64 x (digest, length), the key.
        from zktls_amd.device import sha256_air, sha256_padding_publics, shard_verifier_max_proofs, verify_shard_recursive as _vsr
        sprog = sha256_air()
        s_log_n = 14
        if shard_verifier_max_proofs(s_log_n, 640, tprm.num_queries, tprm.pow_bits, 91, tprm, program=sprog) >= 64:
            def statement(d, L):
                limbs = []
                for i in range(8):
                    w_ = int.from_bytes(d[4 * i:4 * i + 4], "big")
                    limbs += [w_ & 0xffff, w_ >> 16]
                return limbs + sha256_padding_publics(L).tolist()
            t_inner, t_cmp, joined64, spubs, sinner = 1e9, 1e9, None, None, None
            ckey64 = ctx.shard_verifier_setup(s_log_n, 640, tprm.num_queries, tprm.pow_bits, 91, tprm, n_proofs=64, program=sprog)
            for _ in range(3):
                tb0 = time.perf_counter()
                made = prove_transcripts(msgs, tprm, devices=[local_rank], keyed=False)[1]
                tb1 = time.perf_counter()
                sinner, spubs = [pf for _, pf in made], [statement(d, len(m)) for (d, _), m in zip(made, msgs)]
                tb2 = time.perf_counter()
                joined64 = ctx.prove_shard_verifier(ckey64, sinner, s_log_n, 640, spubs, tprm, tprm, program=sprog)
                tb3 = time.perf_counter()
                t_inner, t_cmp = min(t_inner, tb1 - tb0), min(t_cmp, tb3 - tb2)
            tb0 = time.perf_counter()
            ok_c = _vsr(joined64, s_log_n, 640, tprm.num_queries, tprm.pow_bits, [v for pv_ in spubs for v in pv_], ckey64.root, tprm, n_proofs=64, program=sprog) == (0, 0)
            t_cv = time.perf_counter() - tb0
            ckey64.close()
            batch64["compressed"] = {"workload": "the 64 transcripts -> ONE proof: zkhip_prove_transcripts_air (64 version-7 proofs of the chip alone, 2^14 x 640, 100 queries, lock-step lanes), then zkhip_prove_shard_verifier_air with n_proofs = 64",
                                     "inner_ms": round(t_inner * 1e3, 2), "compress_ms": round(t_cmp * 1e3, 2), "ms": round((t_inner + t_cmp) * 1e3, 2),
                                     "inner_bytes_total": int(sum(x.size for x in sinner)), "bytes": int(joined64.size), "compression": round(sum(x.size for x in sinner) / joined64.size, 2),
                                     "host_verify_ms": round(t_cv * 1e3, 2), "verified": bool(ok_c),
                                     "verifier_inputs": "the chip's program, 64 x (digest, message length), the key of (shape, program) -- derivable on the host; no byte of an inner proof"}
            batch64["compressed_ms"], batch64["compressed_bytes"] = batch64["compressed"]["ms"], batch64["compressed"]["bytes"]
            # ... and the KEYED proofs of the batch above themselves (version 11: two chips, a preprocessed table, a bus) as ONE proof: machine mode
            from zktls_amd.device import sha256_inner_machine, verify_machine_recursive as _vmr
            kim = sha256_inner_machine(len(msgs[0]), vk_l, tprm)
            kkey = ctx.machine_verifier_setup(kim, tprm, 64)
            kproofs = [pf for _, pf in res_l]
            kpubs = [statement(d, len(m)) for (d, _), m in zip(res_l, msgs)]
            t_k, ktop = 1e9, None
            kcpu0, kwall0 = time.process_time(), time.perf_counter()
            for _ in range(3):
                tb0 = time.perf_counter()
                ktop = ctx.prove_machine_verifier(kkey, kim, kproofs, kpubs, tprm)
                t_k = min(t_k, time.perf_counter() - tb0)
            k_cores = (time.process_time() - kcpu0) / (time.perf_counter() - kwall0)
            ok_k = _vmr(kim, ktop, [v for pv_ in kpubs for v in pv_], kkey.root, tprm, 64) == (0, 0)
            kkey.close()
            batch64["compressed_keyed"] = {"workload": "the 64 KEYED proofs of the zkhip_prove_transcripts call above -> ONE proof: zkhip_prove_machine_verifier (machine mode: lookups, two heights, the preprocessed table's openings against the machine's key, in-circuit)",
                                           "compress_ms": round(t_k * 1e3, 2), "host_cores_busy": round(k_cores, 2), "ms": round((t_lock + t_k) * 1e3, 2), "inner_bytes_total": int(sum(x.size for x in kproofs)),
                                           "bytes": int(ktop.size), "compression": round(sum(x.size for x in kproofs) / ktop.size, 2), "verified": bool(ok_k)}

    # ---- the compress-like step (sp1.rs:116: core -> compress verifies the shard proofs): the FRI check of sixteen shard proofs of the headline
    # shape proven in-circuit (Merkle paths, folds, challenges, proof of work, query indices) by ONE call of zkhip_prove_fri_indices_batch
    recursion16 = None
    if rank == 0 and not args.no_recursion16 and chip_list is None and host_traces is None and args.shape == "sp1" and LQ == 0 and 6 <= log_n <= 20:
        from zktls_amd.device import prove_fri_indices_batch, set_lockstep, verify_fri_indices
        sps, spv = [], []
        for i in range(16):
            spv.append(public + [1000 + i])
            sps.append(ctx.prove_shard(bufs[i % nbuf], log_n, width, spv[-1], prm))
        set_lockstep(16, 6)
        prove_fri_indices_batch(sps, log_n, width, spv, prm, prm, devices=[local_rank])
        t_rec, rec = 1e9, None
        for _ in range(3):
            tb0 = time.perf_counter()
            rec = prove_fri_indices_batch(sps, log_n, width, spv, prm, prm, devices=[local_rank])
            t_rec = min(t_rec, time.perf_counter() - tb0)
        ok_rec = all(verify_fri_indices(p, fin, cap, log_n, prm.num_queries, prm.pow_bits, vk, prm) == (0, 0) for p, vk, fin, cap in rec)
        # ... and THE JOIN: the same sixteen shard proofs verified WHOLE (transcript, AIR identity at zeta, every opening, reduced openings, FRI, proof
        # of work) by ONE outer proof whose verifier takes the 16 x n_public public values and the shape's key -- no byte of an inner proof
        from zktls_amd.device import verify_shard_recursive
        set_lockstep(16, 6)
        jkey = ctx.shard_verifier_setup(log_n, width, prm.num_queries, prm.pow_bits, len(spv[0]), prm, n_proofs=16)
        ctx.prove_shard_verifier(jkey, sps, log_n, width, spv, prm, prm)
        t_join, joined = 1e9, None
        for _ in range(3):
            tb0 = time.perf_counter()
            joined = ctx.prove_shard_verifier(jkey, sps, log_n, width, spv, prm, prm)
            t_join = min(t_join, time.perf_counter() - tb0)
        tb0 = time.perf_counter()
        ok_join = verify_shard_recursive(joined, log_n, width, prm.num_queries, prm.pow_bits, [v for pv_ in spv for v in pv_], jkey.root, prm, n_proofs=16) == (0, 0)
        t_join_verify = time.perf_counter() - tb0
        join16 = {"workload": "16 shard proofs (2^%d x %d, 100 queries) -> 1 proof: zkhip_prove_shard_verifier with n_proofs = 16 (eight chips; the Poseidon2 chip has %s rows), shard proofs in as bytes" % (log_n, width, "2^19" if log_n == 20 and width == 256 else "fewer"),
                  "ms": round(t_join * 1e3, 2), "ms_per_inner_proof": round(t_join * 1e3 / 16, 3), "inner_bytes_total": int(sum(p_.size for p_ in sps)), "outer_bytes": int(joined.size),
                  "compression": round(sum(p_.size for p_ in sps) / joined.size, 2), "host_verify_ms": round(t_join_verify * 1e3, 2), "verified": bool(ok_join),
                  "verifier_inputs": "shape, 16 x %d public values, the shape's key (8 words); no byte of an inner proof" % len(spv[0])}
        jkey.close()
        # the same join with the OUTER proof in SP1's compress shape (blowup 4, 50 queries, 16 proof-of-work bits: ZKHIP_PARAMS_SP1_COMPRESS, [RECALLED]):
        # half the queries over twice the domain -- a smaller proof for more transform and hashing work
        from zktls_amd._lib import Params as _P
        cprm = _P(2, 50, 16)
        ckey = ctx.shard_verifier_setup(log_n, width, prm.num_queries, prm.pow_bits, len(spv[0]), cprm, n_proofs=16)
        ctx.prove_shard_verifier(ckey, sps, log_n, width, spv, prm, cprm)
        t_cj, cjoined = 1e9, None
        for _ in range(2):
            tb0 = time.perf_counter()
            cjoined = ctx.prove_shard_verifier(ckey, sps, log_n, width, spv, prm, cprm)
            t_cj = min(t_cj, time.perf_counter() - tb0)
        ok_cj = verify_shard_recursive(cjoined, log_n, width, prm.num_queries, prm.pow_bits, [v for pv_ in spv for v in pv_], ckey.root, cprm, n_proofs=16) == (0, 0)
        ckey.close()
        join16["outer_in_sp1_compress_shape"] = {"outer_params": "log_blowup 2, 50 queries, 16 PoW bits", "ms": round(t_cj * 1e3, 2), "outer_bytes": int(cjoined.size),
                                                 "compression": round(sum(p_.size for p_ in sps) / cjoined.size, 2), "verified": bool(ok_cj)}
        # ... and THE TREE (machine mode: csrc/machine_verifier.inl): 64 shard proofs -> four joins of 16 -> ONE proof that verifies the four joins
        # in-circuit (zkhip_prove_machine_verifier: ten chips; the join machine's own description is the inner machine)
        tree = None
        if log_n == 20 and width == 256:
            from zktls_amd.device import InnerMachine, machine_verifier_key_host, prove_shard_verifier_batch, shard_verifier_describe, verify_machine_recursive
            tkey1 = ctx.shard_verifier_setup(log_n, width, prm.num_queries, prm.pow_bits, len(spv[0]), prm, n_proofs=16)
            tsps, tspv = list(sps), list(spv)
            for i in range(16, 64):
                tspv.append(public + [1000 + i])
                tsps.append(ctx.prove_shard(bufs[i % nbuf], log_n, width, tspv[-1], prm))
            chips_ = []
            for i in range(8):
                p_, ln_, mw_, pw_ = shard_verifier_describe(log_n, width, prm.num_queries, prm.pow_bits, len(spv[0]), i, 0, 16)
                t_, _, _, _ = shard_verifier_describe(log_n, width, prm.num_queries, prm.pow_bits, len(spv[0]), i, 1, 16)
                chips_.append(dict(ln=ln_, W=mw_, Pw=pw_, prog=p_, tab=t_))
            im_ = InnerMachine(chips_, tkey1.root, prm.num_queries, prm.pow_bits, 16 * len(spv[0]))
            tkey2 = ctx.machine_verifier_setup(im_, prm, 4)
            t_joins, t_top, top, joins4 = 1e9, 1e9, None, None
            jp = [[v for pv_ in tspv[16 * j:16 * j + 16] for v in pv_] for j in range(4)]
            for _ in range(3):
                tb0 = time.perf_counter()
                joins4, jvk = prove_shard_verifier_batch(tsps, 16, log_n, width, tspv, prm, prm, devices=[local_rank], in_flight=4)
                tb1 = time.perf_counter()
                top = ctx.prove_machine_verifier(tkey2, im_, joins4, jp, prm)
                tb2 = time.perf_counter()
                t_joins, t_top = min(t_joins, tb1 - tb0), min(t_top, tb2 - tb1)
            # ... and the same tree in ONE call (zkhip_prove_shard_tree): a join's tables for the top are filled on its worker's thread the moment the join
            # exists, beside the joins still being proven -- after the last join only the uploads and the machine's proof remain.  Same bytes.
            from zktls_amd.device import prove_shard_tree
            t_one = 1e9
            tcpu0, twall0 = time.process_time(), time.perf_counter()
            for _ in range(3):
                tb0 = time.perf_counter()
                top1, joins1, jvk1 = prove_shard_tree(ctx, tkey2, im_, tsps, 16, log_n, width, tspv, prm, prm, prm, devices=[local_rank], in_flight=4)
                t_one = min(t_one, time.perf_counter() - tb0)
            tree_cores = (time.process_time() - tcpu0) / (time.perf_counter() - twall0)      # CPU time of ALL threads of the process per second of the calls
            assert top1.tobytes() == top.tobytes() and all(a_.tobytes() == b_.tobytes() for a_, b_ in zip(joins1, joins4))
            tb0 = time.perf_counter()
            hkey_ = machine_verifier_key_host(im_, prm, 4)            # the top's key on the host's cores: a function of the shape, derived ONCE by a verifier and kept
            t_hk = time.perf_counter() - tb0
            tb0 = time.perf_counter()
            ok_tree = verify_machine_recursive(im_, top, [v for p_ in jp for v in p_], hkey_, prm, 4) == (0, 0)
            t_tv1 = time.perf_counter() - tb0
            t_tv = t_hk + t_tv1
            assert jvk.tolist() == tkey1.root.tolist()
            tree = {"workload": "64 shard proofs (2^20 x 256, 100 queries) -> 4 joins of 16 (zkhip_prove_shard_verifier_batch: the four in flight on pooled contexts, as the shards below them are) -> ONE proof (zkhip_prove_machine_verifier: the four joins' version-11 proofs verified in-circuit, ten chips)",
                    "ms": round(t_one * 1e3, 2), "entry": "zkhip_prove_shard_tree (one call: the four joins in flight, each one's tables for the top filled the moment it exists, then the machine's proof)",
                    "host_cores_busy": round(tree_cores, 2), "host_wait": "block" if wait_block else "poll",
                    "two_calls_ms": round((t_joins + t_top) * 1e3, 2), "joins_ms": round(t_joins * 1e3, 2), "top_ms": round(t_top * 1e3, 2), "inner_bytes_total": int(sum(x.size for x in tsps)),
                    "join_bytes_total": int(sum(x.size for x in joins4)), "bytes": int(top.size), "compression": round(sum(x.size for x in tsps) / top.size, 2),
                    "host_verify_ms_with_the_key_derived_on_the_host": round(t_tv * 1e3, 2), "host_verify_ms": round(t_tv1 * 1e3, 2), "host_key_derivation_ms_once_per_shape": round(t_hk * 1e3, 2),
                    "verified": bool(ok_tree),
                    "verifier_inputs": "the join machine's description (a function of the shard shape), 64 x %d public values, the key; no byte of a shard proof or of a join" % len(spv[0])}
            # the same top with ITS proof in SP1's compress shape (blowup 4, 50 queries): what leaves the tree is half the size
            from zktls_amd._lib import Params as _P2
            cprm2 = _P2(2, 50, 16)
            tkey3 = ctx.machine_verifier_setup(im_, cprm2, 4)
            ctx.prove_machine_verifier(tkey3, im_, joins4, jp, cprm2)
            tb0 = time.perf_counter()
            ctop = ctx.prove_machine_verifier(tkey3, im_, joins4, jp, cprm2)
            t_ctop = time.perf_counter() - tb0
            ok_ctop = verify_machine_recursive(im_, ctop, [v for p_ in jp for v in p_], tkey3.root, cprm2, 4) == (0, 0)
            tree["top_in_sp1_compress_shape"] = {"outer_params": "log_blowup 2, 50 queries, 16 PoW bits", "top_ms": round(t_ctop * 1e3, 2), "bytes": int(ctop.size),
                                                 "compression": round(sum(x.size for x in tsps) / ctop.size, 2), "verified": bool(ok_ctop)}
            tkey3.close()
            tkey1.close(), tkey2.close()
        recursion16 = {"join16": join16, "tree": tree, "tree_ms": tree["ms"] if tree else None, "tree_bytes": tree["bytes"] if tree else None, "fri_only_workload": "the FRI check of 16 shard proofs (2^%d x %d, 100 queries x %d layers each) proven in-circuit: Poseidon2 chip (Merkle paths + transcript) + FRI-fold chip + SAMPLES chip + two tables per proof, one zkhip_prove_fri_indices_batch call, shard proofs in as bytes (host view included)" % (log_n, width, log_n),
                       "fri_only_ms": round(t_rec * 1e3, 2), "fri_only_proof_bytes": int(rec[0][0].size), "fri_only_all_verified": bool(ok_rec),
                       "ms": round(t_join * 1e3, 2), "recursion_proofs_per_s": round(16 / t_join, 1), "proof_bytes": int(joined.size), "all_verified": bool(ok_join)}

    # ---- SP1's real shard structure beside the headline: six chips of different heights in ONE proof (mixed-height commitments, row a7)
    # with in-table LogUp pairs (permutation traces + their commitment, row a8), proven with the same number of shards in flight
    multichip = None
    if rank == 0 and not args.no_multichip and chip_list is None and host_traces is None and args.shape == "sp1" and LQ == 0 and log_n == 20 and not one_proc:
        from zktls_amd.device import verify_chips
        mc_spec = [(20, 96), (20, 32), (19, 64), (18, 128), (16, 256), (14, 40)]
        mc_pairs = [max(1, w_ // 32) for _, w_ in mc_spec]
        mc_cells = sum(w_ << ln_ for ln_, w_ in mc_spec)
        mc_bufs = [(ctx.gen_trace_logup(SEED, 7000 + j_, ln_, w_, q_), ln_, w_, q_) for j_, ((ln_, w_), q_) in enumerate(zip(mc_spec, mc_pairs))]
        ctx.sync()
        import threading
        mc_n = 2 * S
        mc_out = [None] * (mc_n + S)

        def mc_worker(w_, lo_, cnt_):
            for i_ in range(lo_ + w_, lo_ + cnt_, S):
                mc_out[i_] = ctxs[w_].prove_chips(mc_bufs, public + [i_], prm)

        def mc_round(lo_, cnt_):
            ts_ = [threading.Thread(target=mc_worker, args=(w_, lo_, cnt_)) for w_ in range(S)]
            for t_ in ts_:
                t_.start()
            for t_ in ts_:
                t_.join()
        mc_round(0, S)                                        # warm every context at these shapes
        for c_ in ctxs:
            c_.sync()
        tm0 = time.perf_counter()
        mc_round(S, mc_n)
        for c_ in ctxs:
            c_.sync()
        mc_dt = (time.perf_counter() - tm0) / mc_n
        mc_rc, _ = verify_chips(mc_out[S + mc_n - 1], [c_[0] for c_ in mc_spec], [c_[1] for c_ in mc_spec], public + [S + mc_n - 1], prm, pairs=mc_pairs)
        multichip = {"workload": "one shard of six chips %s with in-table LogUp pairs %s (three commitments: main, permutation, quotient; mixed heights), %d trace cells + the permutation traces, log_blowup 1, 100 queries, 16 PoW bits, full zkhip_prove_chips, %d in flight" % (
                         ",".join("%dx%d" % c_ for c_ in mc_spec), mc_pairs, mc_cells, S),
                     "ms_per_shard": round(mc_dt * 1e3, 3), "trace_cells_per_s": round(mc_cells / mc_dt, 1), "shards_timed": mc_n,
                     "proof_bytes": int(mc_out[S + mc_n - 1].size), "verified": bool(mc_rc == 0)}
        for b_ in mc_bufs:
            b_[0].free()
        del mc_out
        # ---- the same shard structure through core -> compress (sp1.rs:116: core, then compress, in one client.prove): the six chips
        # as ONE keyed machine (proof version 11: constraint programs, the LogUp pairs as interaction tables -- two of them ACROSS the two 2^20-row
        # tables --, 32 preprocessed columns committed once by setup), mc_k shards proven with S in flight, then joined into ONE proof in machine mode
        # (zkhip_prove_machine_verifier: lookups, mixed heights, the preprocessed openings against the machine's key, in-circuit)
        try:
            from zktls_amd.device import Sp1ShapedShard, machine_verifier_key_host, verify_machine_recursive
            shape_ = Sp1ShapedShard()
            mc_keys = [shape_.setup(c_, SEED, prm) for c_ in ctxs]                  # a key belongs to the context it was made with
            mc_k = 2 * S
            mc_pubs = [public + [9000 + i_] for i_ in range(mc_k)]
            mc_tr = [shape_.gen_traces(ctx, SEED, 9000 + i_) for i_ in range(mc_k)]
            ctx.sync()
            mc_proofs = [None] * mc_k

            def mk_worker(w_, lo_, hi_):
                for i_ in range(lo_ + w_, hi_, S):
                    mc_proofs[i_] = ctxs[w_].prove_machine_keyed(mc_keys[w_][0], shape_.main_chips(mc_tr[i_]), shape_.programs, shape_.tables, mc_pubs[i_], prm)

            def mk_round(lo_, hi_):
                ts_ = [threading.Thread(target=mk_worker, args=(w_, lo_, hi_)) for w_ in range(S)]
                for t_ in ts_:
                    t_.start()
                for t_ in ts_:
                    t_.join()
            mk_round(0, S)                                                        # warm every context at these shapes
            tk0 = time.perf_counter()
            mk_round(0, mc_k)
            for c_ in ctxs:
                c_.sync()
            mk_dt = (time.perf_counter() - tk0) / mc_k
            for tr_ in mc_tr:
                for b_ in tr_:
                    b_.free()
            im_ = shape_.inner_machine(mc_keys[0][0].root, prm)
            jk_ = ctx.machine_verifier_setup(im_, prm, mc_k)
            ctx.prove_machine_verifier(jk_, im_, mc_proofs, mc_pubs, prm)            # warm: workspaces, host tables
            tj0 = time.perf_counter()
            mc_top = ctx.prove_machine_verifier(jk_, im_, mc_proofs, mc_pubs, prm)
            mj_dt = time.perf_counter() - tj0
            flat_ = [v_ for p_ in mc_pubs for v_ in p_]
            th0 = time.perf_counter()
            mc_ok = verify_machine_recursive(im_, mc_top, flat_, jk_.root, prm, mc_k) == (0, 0)
            mh_dt = time.perf_counter() - th0
            swapped_ = mc_pubs[1] + mc_pubs[0] + [v_ for p_ in mc_pubs[2:] for v_ in p_]
            mc_ok = mc_ok and verify_machine_recursive(im_, mc_top, swapped_, jk_.root, prm, mc_k)[0] != 0
            inner_total = sum(int(p_.size) for p_ in mc_proofs)
            multichip["compressed"] = {
                "workload": "the same six chips as ONE keyed machine (version 11: constraint programs, pairs %s as interaction tables -- chips 0 and 1 look each other up ACROSS tables --, 32 preprocessed columns on the 2^16-row chip), %d shards proven with %d in flight (zkhip_prove_machine_keyed), then ONE zkhip_prove_machine_verifier call over them (machine mode: n_proofs = %d)" % (
                    [c_[2] for c_ in shape_.spec], mc_k, S, mc_k),
                "keyed_ms_per_shard": round(mk_dt * 1e3, 3), "keyed_trace_cells_per_s": round(shape_.cells / mk_dt, 1), "keyed_proof_bytes": int(mc_proofs[0].size),
                "compress_ms": round(mj_dt * 1e3, 2), "inner_bytes_total": inner_total, "bytes": int(mc_top.size), "compression": round(inner_total / mc_top.size, 2),
                "host_verify_ms": round(mh_dt * 1e3, 2), "verified_and_a_swap_refused": bool(mc_ok),
                "verifier_inputs": "the machine's description (programs, tables, heights, key root), %d x %d public values, the join's key; no byte of a shard proof" % (mc_k, shape_.n_public),
                "bound": "machine mode takes inner machines of at most 16 chips and a Poseidon2 chip of at most 2^22 rows per join"}
            multichip["compressed_ms"] = multichip["compressed"]["compress_ms"]
            multichip["compressed_bytes"] = multichip["compressed"]["bytes"]
            jk_.close()
            for k_, keep_ in mc_keys:
                k_.close()
                for b_ in keep_:
                    if b_ is not None:
                        b_.free()
            del mc_proofs, mc_top
        except Exception as e_:                                               # the line says so instead of dropping the section silently
            multichip["compressed"] = {"error": repr(e_)}

    # ---- the plug point itself on the measuring path: ONE call of the host mirror of ZkProver::prove
    # (zktls_amd/host, what crates/guest-prover-sp1/src/sp1.rs:102-133 would bind) for an execution of the size the reference benchmarks
    # -- 22.1 M cycles = 22 shards of 2^20 rows (/root/reference/benchmark.md:9) -- first core only (every shard proven and checked,
    # sp1.rs:116-120), then core + COMPRESS (the shard proofs joined into one proof, checked on the host without them)
    execution = None
    mirror_so = os.path.join(ROOT, "zktls_amd", "libzktls_guest_prover.so")
    if (rank == 0 and world == 1 and not args.no_execution and chip_list is None and host_traces is None and args.shape == "sp1" and LQ == 0 and log_n <= 20 and width % 8 == 0
            and not one_proc and os.path.exists(mirror_so)):
        import ctypes as C_

        class _Plan(C_.Structure):
            _fields_ = [("log_n", C_.c_int32), ("width", C_.c_uint32), ("shards", C_.c_uint32), ("num_queries", C_.c_int32), ("pow_bits", C_.c_int32)]
        ML = C_.CDLL(mirror_so)
        u8pp_, szp_ = C_.POINTER(C_.POINTER(C_.c_uint8)), C_.POINTER(C_.c_size_t)
        sig_ = [C_.c_int, C_.c_int, C_.POINTER(_Plan), C_.c_char_p, C_.c_size_t, C_.c_char_p, C_.c_size_t, u8pp_, szp_, u8pp_, szp_, C_.c_char_p, C_.c_size_t]
        ML.zktls_guest_prove.argtypes = sig_
        ML.zktls_guest_prove_compressed.argtypes = sig_
        ML.zktls_compress_key.argtypes = [C_.c_int, C_.POINTER(_Plan), C_.POINTER(C_.c_uint32), C_.c_char_p, C_.c_size_t]
        ML.zktls_verify_compressed_blob.argtypes = [C_.c_char_p, C_.c_size_t, C_.POINTER(_Plan), C_.c_char_p, C_.c_size_t, C_.c_char_p, C_.c_size_t, C_.POINTER(C_.c_uint32), C_.POINTER(C_.c_int)]
        ML.zktls_free.argtypes = [C_.c_void_p]
        eplan = _Plan(log_n, width, 22, prm.num_queries, prm.pow_bits)
        ecbor, eelf = b"\xa1bench execution", b"\x7fELF bench guest"

        def _run(fn):
            out_, outn_, pr_, prn_ = C_.POINTER(C_.c_uint8)(), C_.c_size_t(), C_.POINTER(C_.c_uint8)(), C_.c_size_t()
            err_ = C_.create_string_buffer(512)
            t0_ = time.perf_counter()
            rc_ = fn(local_rank, 2, C_.byref(eplan), ecbor, len(ecbor), eelf, len(eelf), C_.byref(out_), C_.byref(outn_), C_.byref(pr_), C_.byref(prn_), err_, 512)
            dt_ = time.perf_counter() - t0_
            if rc_ != 0:
                raise RuntimeError("host mirror: %s" % err_.value.decode("utf-8", "replace"))
            blob_ = C_.string_at(pr_, prn_.value)
            ML.zktls_free(out_)
            ML.zktls_free(pr_)
            return dt_, blob_
        _run(ML.zktls_guest_prove)                                   # warm the mirror's own contexts
        t_core, b_core = min((_run(ML.zktls_guest_prove) for _ in range(2)), key=lambda x_: x_[0])
        _run(ML.zktls_guest_prove_compressed)
        t_cc, b_cc = min((_run(ML.zktls_guest_prove_compressed) for _ in range(2)), key=lambda x_: x_[0])
        ekey = (C_.c_uint32 * 8)()
        eerr = C_.create_string_buffer(512)
        ok_key = ML.zktls_compress_key(local_rank, C_.byref(eplan), ekey, eerr, 512) == 0
        t0_ = time.perf_counter()
        ok_blob = ok_key and ML.zktls_verify_compressed_blob(b_cc, len(b_cc), C_.byref(eplan), ecbor, len(ecbor), eelf, len(eelf), ekey, None) == 0
        t_ev = time.perf_counter() - t0_
        ML.zktls_release_cached()
        execution = {"workload": "ONE call of the host mirror of ZkProver::prove (libzktls_guest_prover.so) for 22 synthetic shards of 2^%d x %d -- the size of the reference's 22.1 M-cycle benchmark execution; every proof checked by the prover as sp1.rs:120 does" % (log_n, width),
                     "core_ms": round(t_core * 1e3, 1), "core_ms_per_shard": round(t_core * 1e3 / 22, 2), "core_blob_bytes": len(b_core),
                     "core_plus_compress_ms": round(t_cc * 1e3, 1), "compressed_blob_bytes": len(b_cc), "compressed_blob_verified_on_host": bool(ok_blob),
                     "compressed_blob_host_verify_ms": round(t_ev * 1e3, 2),
                     "note": "synthetic AIR in SP1's shard shape; no executor, no shrink / wrap / Groth16 (out of scope): not comparable with the reference's end-to-end seconds"}
        # ... and the same execution in SP1's shard STRUCTURE: 22 shards of the six-chip keyed machine of the `multichip` section through the same plug
        # point -- setup (the preprocessed columns committed once), every shard ONE version-11 proof checked against the key (sp1.rs:113-120), then the compress stage in
        # machine mode behind the same call; the compressed blob checked on the host from (plan, input, ELF, vk) alone
        try:
            from zktls_amd.device import SP1_SHAPED_SPEC, SP1_SHAPED_PRE

            class _MPlan(C_.Structure):
                _fields_ = [("n_chips", C_.c_int32), ("log_ns", C_.POINTER(C_.c_int32)), ("widths", C_.POINTER(C_.c_uint32)), ("pairs", C_.POINTER(C_.c_uint32)),
                            ("partners", C_.POINTER(C_.c_int32)), ("pre_widths", C_.POINTER(C_.c_uint32)), ("shards", C_.c_uint32), ("num_queries", C_.c_int32),
                            ("pow_bits", C_.c_int32), ("in_flight", C_.c_uint32)]
            nch_, prew_ = len(SP1_SHAPED_SPEC), dict(SP1_SHAPED_PRE)
            keep_ = ((C_.c_int32 * nch_)(*[c_[0] for c_ in SP1_SHAPED_SPEC]), (C_.c_uint32 * nch_)(*[c_[1] for c_ in SP1_SHAPED_SPEC]), (C_.c_uint32 * nch_)(*[c_[2] for c_ in SP1_SHAPED_SPEC]),
                     (C_.c_int32 * nch_)(*[c_[3] for c_ in SP1_SHAPED_SPEC]), (C_.c_uint32 * nch_)(*[prew_.get(i_, 0) for i_ in range(nch_)]))
            mplan_ = _MPlan(nch_, keep_[0], keep_[1], keep_[2], keep_[3], keep_[4], 22, prm.num_queries, prm.pow_bits, 0)
            ML.zktls_guest_prove_machine.argtypes = [C_.c_int, C_.c_int, C_.POINTER(_MPlan), C_.c_int, C_.c_int, C_.c_char_p, C_.c_size_t, C_.c_char_p, C_.c_size_t, u8pp_, szp_, u8pp_, szp_,
                                                     C_.c_char_p, C_.c_char_p, C_.c_size_t]
            ML.zktls_verify_machine_blob.argtypes = [C_.c_char_p, C_.c_size_t, C_.POINTER(_MPlan), C_.c_char_p, C_.c_size_t, C_.c_char_p, C_.c_size_t, C_.c_char_p, C_.POINTER(C_.c_int)]

            def _mrun(compress_):
                out_, outn_, pr_, prn_ = C_.POINTER(C_.c_uint8)(), C_.c_size_t(), C_.POINTER(C_.c_uint8)(), C_.c_size_t()
                err_, vk_ = C_.create_string_buffer(512), C_.create_string_buffer(64)
                t0_ = time.perf_counter()
                rc_ = ML.zktls_guest_prove_machine(local_rank, 2, C_.byref(mplan_), compress_, 1, ecbor, len(ecbor), eelf, len(eelf), C_.byref(out_), C_.byref(outn_), C_.byref(pr_),
                                                   C_.byref(prn_), vk_, err_, 512)
                dt_ = time.perf_counter() - t0_
                if rc_ != 0:
                    raise RuntimeError("host mirror, machine shards: %s" % err_.value.decode("utf-8", "replace"))
                blob_ = C_.string_at(pr_, prn_.value)
                ML.zktls_free(out_)
                ML.zktls_free(pr_)
                return dt_, blob_, vk_.raw
            _mrun(0)
            tm_core, bm_core, mvk_ = min((_mrun(0) for _ in range(2)), key=lambda x_: x_[0])
            _mrun(1)
            tm_cc, bm_cc, mvk2_ = min((_mrun(1) for _ in range(2)), key=lambda x_: x_[0])
            t0_ = time.perf_counter()
            okm_ = mvk_ == mvk2_ and ML.zktls_verify_machine_blob(bm_cc, len(bm_cc), C_.byref(mplan_), ecbor, len(ecbor), eelf, len(eelf), mvk_, None) == 0
            tm_ev = time.perf_counter() - t0_
            t0_ = time.perf_counter()
            okm_ = okm_ and ML.zktls_verify_machine_blob(bm_cc, len(bm_cc), C_.byref(mplan_), ecbor, len(ecbor), eelf, len(eelf), mvk_, None) == 0
            tm_ev2 = time.perf_counter() - t0_
            okm_ = okm_ and ML.zktls_verify_machine_blob(bm_cc, len(bm_cc), C_.byref(mplan_), ecbor + b"!", len(ecbor) + 1, eelf, len(eelf), mvk_, None) == -2
            mcells_ = sum(c_[1] << c_[0] for c_ in SP1_SHAPED_SPEC)
            execution["multichip"] = {
                "workload": "the same ONE call for 22 shards in SP1's shard structure: six chips %s (LogUp pairs in-table and ACROSS the two 2^20-row tables, 32 preprocessed columns), setup + every shard one keyed-machine proof (version 11) checked against the key, then the compress stage in machine mode (22 shard proofs -> ONE proof)" % (
                    ",".join("%dx%d" % (c_[0], c_[1]) for c_ in SP1_SHAPED_SPEC)),
                "core_ms": round(tm_core * 1e3, 1), "core_ms_per_shard": round(tm_core * 1e3 / 22, 2), "core_trace_cells_per_s": round(22 * mcells_ / tm_core, 1), "core_blob_bytes": len(bm_core),
                "core_plus_compress_ms": round(tm_cc * 1e3, 1), "compressed_blob_bytes": len(bm_cc), "compressed_blob_verified_on_host_and_another_request_refused": bool(okm_),
                "compressed_blob_host_verify_ms_with_the_join_key_derived": round(tm_ev * 1e3, 2), "compressed_blob_host_verify_ms": round(tm_ev2 * 1e3, 2),
                "note": "the join's key is a function of (plan, vk): the verifier derives it on its host cores once per plan (the first figure) and keeps it (the second)"}
            ML.zktls_release_cached()
        except Exception as e_:
            execution["multichip"] = {"error": repr(e_)}

    # ---- CPU baseline: the oracle on the host cores, bounded sample of the same workload
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        cores = os.cpu_count() or 1
        # the scalar oracle's loops stop scaling long before 256 threads, and a container may be allowed fewer cores than it sees (cgroup
        # quota: 16 of 256 on the MI355X boxes of this pool; a team of 64 threads on 16 cores is 12 % slower than one of 16): cap, and say so
        quota = int(cores_ok) if cores_ok and cores_ok >= 1 else cores
        used = O.set_threads(args.cpu_threads if args.cpu_threads > 0 else max(1, min(cores, 64, quota)))
        cl = args.cpu_log_n
        tr = O.gen_trace_logup(SEED, 0, cl, width, LQ) if LQ else O.gen_trace(SEED, 0, cl, width)
        oprm = O.default_params(1, 100, 16, LQ) if args.shape == "sp1" else O.default_params(2, 50, 0, LQ, 4, 8, 24)
        tc0 = time.perf_counter()
        simd_proof = O.prove_shard(tr, public + [0], oprm)
        dt_simd = time.perf_counter() - tc0
        simd_on = bool(O.lib().orc_simd_enabled())
        # the SCALAR form of the same oracle (ORC_NO_SIMD=1 is read once per process: a child), the baseline of rounds 1 - 5; the AVX-512 form beside it
        dt = dt_simd
        same = None
        if simd_on and not LQ and args.shape == "sp1":
            import hashlib
            import subprocess
            code = ("import sys, time, hashlib; sys.path.insert(0, %r); import oracle_lib as O; O.set_threads(%d); tr = O.gen_trace(%d, 0, %d, %d); t0 = time.perf_counter(); "
                    "p = O.prove_shard(tr, %r, O.default_params(1, 100, 16)); print(time.perf_counter() - t0, hashlib.sha256(p.tobytes()).hexdigest())") % (
                        os.path.join(ROOT, "tests"), used, SEED, cl, width, public + [0])
            try:
                r_ = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ORC_NO_SIMD="1"), capture_output=True, text=True, timeout=600)
                dt, dg_ = float(r_.stdout.split()[0]), r_.stdout.split()[1]
                same = dg_ == hashlib.sha256(simd_proof.tobytes()).hexdigest()
            except Exception:
                dt = dt_simd
        cpu = {"value": round((width << cl) / dt, 1), "unit": "trace-cells/s", "cores": used, "kind": "port",
               "sample": "one 2^%d x %d shard proof (same AIR, %s), %.1f s, C oracle with its scalar Poseidon2 (ORC_NO_SIMD=1, a child process) + OpenMP on %d threads (%d host CPUs visible, CPU quota of the container %s cores)" % (
                   cl, width, "log_blowup 1, 100 queries, 16 PoW bits" if args.shape == "sp1" else "RISC-Zero-like shape", dt, used, cores,
                   ("%g" % cores_ok) if cores_ok else "none")}
        if simd_on:
            cpu["simd"] = {"value": round((width << cl) / dt_simd, 1), "unit": "trace-cells/s", "cores": used, "seconds": round(dt_simd, 2),
                           "what": "the same oracle proof with its AVX-512 forms on: eight Poseidon2 sponges / compressions per register (oracle/poseidon2_x8.c, self-checked against the scalar "
                                   "permutation at start-up) and division-free radix-4 sweeps in the transforms (oracle/ntt.c)", "same_bytes_as_the_scalar_run": same}
            if same is None:
                cpu["sample"] = cpu["sample"].replace("scalar C oracle", "C oracle (AVX-512 forms on)")

    if rank == 0:
        world = world_eff                               # (one-process mode: the devices of the job)
        total_cells = cells * K * world
        out = {
            "metric": "trace-cells/s",
            "value": round(total_cells / elapsed, 1),
            "unit": "trace-cells/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "streams_per_gpu": in_flight, "host_wait": "block" if wait_block else "poll", "host_cores_usable": cores_ok,
            "host_cores_busy_per_rank": round(host_cores_busy, 2),
            "fri_commit_phase_as_hip_graph": (not args.no_fri_graph), "one_process_mode": bool(one_proc), "logical_devices_test_mode": int(args.logical_devices),
            "rccl_world_size": (dist.get_world_size() if dist is not None else 1),
            "collective_backend": (dist.get_backend() if dist is not None else None), "rccl_selfcheck_calls": rccl_calls,
            "share_gpu_test_mode": bool(args.share_gpu),
            "shards_proven": K * world, "distinct_shards_proven": len(digests), "shard_digests_gathered": len(digests),
            "shard_assignment": "round-robin (zkhip_shard_device: shard s on device s mod N, inside zkhip_prove_shards_multi)" if one_proc else "round-robin (zktls_amd.shards.shard_indices)",
            "timing_note": "ms_per_step is amortised throughput with %d shards in flight per GPU, not latency; the Python harness's cyclic garbage collector is off from the first timed region on (gc.disable, as timeit does)" % in_flight,
            "single_shard_latency_ms": (round(latency_ms, 3) if latency_ms is not None else None),
            "inputs": "host memory, H2D copy inside every step" if host_traces is not None else "resident in HBM",
            "config": {"workload": ("multi-chip shard (SP1's shard structure): chips %s, one commitment per phase, %d trace cells, log_blowup 1, 100 queries, 16 PoW bits, full prove_chips" % (args.chips, cells))
                                   if chip_list is not None else
                                   (("SP1-core-like synthetic shard: 2^%d rows x %d cols BabyBear, log_blowup 1, 100 queries, 16 PoW bits, %s, full prove_shard" if args.shape == "sp1" else
                                     "RISC-Zero-like synthetic segment: 2^%d rows x %d cols BabyBear, blowup 4, 50 queries, FRI fold 16, 256 final coefficients, Poseidon2 width 24, %s, full prove_shard")
                                    % (log_n, width, ("LogUp x%d" % LQ) if LQ else "no lookups")),
                       "parallelism": ("one process, device list (zkhip_prove_shards_multi), shard-parallel x%d" % world) if one_proc else "shard-parallel x%d" % world,
                       "shards_per_step": world},
            "proofs_per_s": round(K * world / elapsed, 3),
            "proof_bytes": int(last.size),
            "verified": bool(verified), "host_verify_ms": round(host_verify_ms, 2),
            "roofline": roof,
            "valu_roofline": valu,
            "transcripts_per_s": (batch64["transcripts_per_s"] if batch64 else None),
            "batch64": batch64,
            "recursion_proofs_per_s": (recursion16["recursion_proofs_per_s"] if recursion16 else None),
            "recursion16": recursion16,
            "recursion": ({"tree_ms": recursion16["tree_ms"], "tree_bytes": recursion16["tree_bytes"], "join16_ms": recursion16["join16"]["ms"], "join16_bytes": recursion16["join16"]["outer_bytes"]} if recursion16 else None),
            "multichip": multichip,
            "execution22": execution,
            "one_process": one_process,
            "one_process_value": (one_process["value"] if one_process else None),
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if one_proc:
        for t_, _ in op_traces.values():
            t_.free()
        for d_, c_ in dev_ctx.items():
            if d_ != 0:
                c_.close()
        zk_lib.load().zkhip_release_cached_contexts()
    if dist is not None:
        dist.barrier()                     # rank 0 is still measuring its roofline section: leave together
        dist.destroy_process_group()
    if not verified:
        raise SystemExit("last proof failed verification (reason %d)" % reason)


if __name__ == "__main__":
    main()
