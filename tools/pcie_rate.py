"""exploration: PCIe-inclusive shard rate through zkhip_prove_shard_host (pageable and pinned host memory)"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from zktls_amd.device import Context
from zktls_amd._lib import Params
log_n, w = 20, 256
ctx = Context(0)
d = ctx.gen_trace(1, 0, log_n, w)
host = d.download().reshape(-1, w)                      # canonical words, pageable numpy memory
pinned = torch.from_numpy(host.astype(np.int32)).pin_memory()
prm = Params(1, 100, 16)
ref = ctx.prove_shard(d, log_n, w, [1], prm)
for name, kw in (("pageable", dict(host_trace=host)), ("pinned", dict(host_trace=None, host_ptr=pinned.data_ptr(), log_n=log_n, width=w))):
    ctx.prove_shard_host(public_values=[1], params=prm, **kw)
    t0 = time.perf_counter(); n = 5
    for _ in range(n): pf = ctx.prove_shard_host(public_values=[1], params=prm, **kw)
    dt = (time.perf_counter() - t0) / n
    assert pf.tobytes() == ref.tobytes()
    print("%s host trace: %.1f ms per shard incl. H2D (1 GiB) -> %.2f G cells/s" % (name, dt * 1e3, (w << log_n) / dt / 1e9))
t0 = time.perf_counter()
for _ in range(5): ctx.prove_shard(d, log_n, w, [1], prm)
print("device-resident: %.1f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
