"""zkhip_set_wait_mode: sleeping instead of polling while a host thread waits for the GPU (hipDeviceScheduleBlockingSync).  The mode is
fixed when the device is first used, so every case runs in a fresh child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, sys, time
sys.path.insert(0, %r)
from zktls_amd import _lib
from zktls_amd._lib import Params
L = _lib.load()
mode = int(sys.argv[1])
rc = L.zkhip_set_wait_mode(mode, 0)
from zktls_amd.device import Context
ctx = Context(0)
late = L.zkhip_set_wait_mode(mode, -1)        # a context exists: refused
tr = ctx.gen_trace(11, 3, 16, 64)
prm = Params(1, 20, 8)
p = ctx.prove_shard(tr, 16, 64, [5, 6], prm)
c0, w0 = time.process_time(), time.perf_counter()
for _ in range(20):
    ctx.prove_shard(tr, 16, 64, [5, 6], prm)
busy = (time.process_time() - c0) / (time.perf_counter() - w0)
print(rc, late, hashlib.sha256(p.tobytes()).hexdigest(), "%%.3f" %% busy)
""" % ROOT


def run(mode):
    out = subprocess.run([sys.executable, "-c", CHILD, str(mode)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rc, late, digest, busy = out.stdout.split()[-4:]
    return int(rc), int(late), digest, float(busy)


def test_blocking_waits_change_the_cpu_use_not_the_proof():
    rc0, late0, d0, busy0 = run(0)
    rc1, late1, d1, busy1 = run(1)
    assert rc0 == 0 and rc1 == 0
    assert late0 == -1 and late1 == -1                      # ZKHIP_ERR_INVALID once a context exists
    assert d0 == d1                                         # same proof bytes
    print("cores busy while proving one shard at a time: polling %.2f, blocking %.2f" % (busy0, busy1))
    assert busy1 < busy0 - 0.2                              # the proving thread sleeps through its waits (one runtime thread keeps polling)
