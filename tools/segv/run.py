#!/usr/bin/env python3
"""root-cause the SIGSEGV inside hipLaunchKernel seen under `rocprofv3 --kernel-trace`.

On the GPU box:  python3 tools/segv/run.py [processes per configuration] [calls per process]
Every configuration is started P times as its own process, with and (two of them) without the profiler; the exit statuses are counted
(-11 / 139 = SIGSEGV, -6 / 134 = abort).  Output: gpurun_out/segv/summary.txt and the stderr tail of every process that crashed.
  nolib_t / nolib_p / nolib_f / nolib_g : tools/segv/repro_nolib (NO library code): short-lived threads / pooled threads / lanes of fibers / HIP graphs
  lib_a .. lib_d               : tools/segv/repro_lib (zkhip_prove_transcripts): lock-step off 1 worker / off 16 workers / 1 lane / 6 lanes
This process never touches the GPU; it only starts children."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out", "segv")


def run(name, prof, cmd, procs):
    ok = crashed = other = 0
    last_out = [""]
    for i in range(procs):
        pdir = "/tmp/segv_prof_%s_%d" % (name, i)
        full = (["rocprofv3", "--kernel-trace", "-d", pdir, "--"] + cmd) if prof else cmd
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            p = subprocess.run(full, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            rc, err = p.returncode, p.stderr.decode("utf-8", "replace")
        except subprocess.TimeoutExpired as e:
            rc, err = 124, (e.stderr or b"").decode("utf-8", "replace")
        shutil.rmtree(pdir, ignore_errors=True)
        if rc == 0:
            ok += 1
            last_out = p.stdout.decode("utf-8", "replace").strip().splitlines()[-1:] or [""]
        elif rc in (-11, 139, -6, 134):
            crashed += 1
            open(os.path.join(OUT, "%s.%d.rc%d.tail" % (name, i, rc)), "w").write(err[-4000:])
        else:
            other += 1
            open(os.path.join(OUT, "%s.%d.rc%d.tail" % (name, i, rc)), "w").write(err[-4000:])
    line = "%s profiler=%d processes=%d ok=%d crashed=%d other=%d   [%s]" % (name, prof, procs, ok, crashed, other, last_out[0])
    print(line, flush=True)
    open(os.path.join(OUT, "summary.txt"), "a").write(line + "\n")


def main():
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    os.makedirs(OUT, exist_ok=True)
    open(os.path.join(OUT, "summary.txt"), "w").write("processes per configuration %d, calls (rounds) per process %d\n" % (procs, calls))
    nolib, lib = os.path.join(ROOT, "tools", "segv", "repro_nolib"), os.path.join(ROOT, "tools", "segv", "repro_lib")
    only = sys.argv[3] if len(sys.argv) > 3 else ""          # e.g. "nolib_g,lib_a": just these configurations
    if only:
        for name in only.split(","):
            kind, m = name.split("_")[:2]
            run(name, 0 if name.endswith("noprofiler") else 1, [nolib if kind == "nolib" else lib, m, str(calls)], procs)
        return
    for m in "tpfg":
        run("nolib_" + m, 1, [nolib, m, str(calls)], procs)
    for m in "abcd":
        run("lib_" + m, 1, [lib, m, str(calls)], procs)
    run("nolib_t_noprofiler", 0, [nolib, "t", str(calls)], procs)
    run("lib_d_noprofiler", 0, [lib, "d", str(calls)], procs)


if __name__ == "__main__":
    main()
