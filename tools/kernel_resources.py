#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of the built objects (zktls_amd/csrc/build/*.hip.o), read from the code objects' metadata.
usage: tools/kernel_resources.py > table"""
import glob, os, re, shutil, subprocess, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zktls_amd", "csrc", "build")
LLVM = "/opt/rocm/lib/llvm/bin/"
tmp = tempfile.mkdtemp()
rows = []
for o in sorted(glob.glob(ROOT + "/*.hip.o")):
    dst = os.path.join(tmp, os.path.basename(o))
    shutil.copy(o, dst)
    subprocess.run([LLVM + "llvm-objdump", "--offloading", dst], capture_output=True, cwd=tmp)
    for co in glob.glob(dst + ".*gfx950"):
        txt = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
            rows.append((name.split("(")[0][-100:], g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("vgpr_spill_count")))
shutil.rmtree(tmp)
for r in sorted(set(rows)):
    print("%-100s vgpr %4s sgpr %4s scratch %5s lds %6s spill %s" % r)
