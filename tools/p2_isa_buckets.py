#!/usr/bin/env python3
"""the instructions of ONE Poseidon2 permutation as the compiler emits it for gfx950, bucketed.  Disassembles
zk::compress_level_kernel (one permutation per lane, nothing else) from the built hash.hip.o, finds the three round loops (full rounds 0-3,
the 13 partial rounds, full rounds 4-7) by their backward branches and multiplies every segment's static counts by its trip count.
Runs anywhere (no GPU).    python3 tools/p2_isa_buckets.py > profiles/r05_leaf_hash_isa.md"""
import collections
import os
import re
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
tmp = tempfile.mkdtemp()
obj = os.path.join(tmp, "hash.hip.o")
shutil.copy(os.path.join(ROOT, "zktls_amd", "csrc", "build", "hash.hip.o"), obj)
subprocess.run([LLVM + "llvm-objdump", "--offloading", obj], capture_output=True, cwd=tmp)
co = [f for f in os.listdir(tmp) if f.endswith("gfx950")][0]
lines = subprocess.run([LLVM + "llvm-objdump", "-d", os.path.join(tmp, co)], capture_output=True, text=True).stdout.split("\n")
shutil.rmtree(tmp)
start = [i for i, l in enumerate(lines) if "<_ZN2zk21compress_level_kernelEPKjPjm>:" in l][0]
end = [i for i, l in enumerate(lines) if "<_ZN2zk27compress_level_kernel_batch" in l][0]
ins = []
for l in lines[start + 1:end]:
    m = re.match(r"\s+(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):", l)
    if m:
        ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
addr = {a: i for i, (a, _, _) in enumerate(ins)}
loops = []
for i, (a, op, args) in enumerate(ins):
    if op == "s_cbranch_scc1":
        simm = int(args)
        simm -= 65536 if simm >= 32768 else 0
        loops.append((addr[a + 4 + 4 * simm], i))
assert len(loops) == 3, loops


def classify(op):
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo", "v_mul_hi")):
        return "multiply (v_mad_i64_i32 / v_mad_u64_u32 / v_mul_lo_u32)"
    if op.startswith("v_cndmask"):
        return "v_cndmask (second half of a conditional subtraction)"
    if op.startswith(("v_subrev_co", "v_sub_co", "v_add_co", "v_addc", "v_subb")):
        return "v_subrev_co / v_add_co (first half of a conditional subtraction)"
    if op.startswith(("v_add", "v_sub", "v_lshl", "v_ashr", "v_lshr")):
        return "plain add / sub / shift"
    if op.startswith(("v_min", "v_max")):
        return "v_min_u32 (dcanon)"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "v_mov"
    if op.startswith("v_"):
        return "other VALU"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "s_nop / s_waitcnt"
    if op.startswith(("s_load", "s_buffer")):
        return "s_load (round constants)"
    if op.startswith("s_"):
        return "SALU"
    return "memory"


segs = [("load + first M_E + constants", 0, loops[0][0], 1), ("full rounds 0-3, per round", loops[0][0], loops[0][1] + 1, 4),
        ("hand-over", loops[0][1] + 1, loops[1][0], 1), ("partial rounds, per round", loops[1][0], loops[1][1] + 1, 13),
        ("dcanon + constants", loops[1][1] + 1, loops[2][0], 1), ("full rounds 4-7, per round", loops[2][0], loops[2][1] + 1, 4),
        ("dcanon + store", loops[2][1] + 1, len(ins), 1)]
classes, tot, table = [], collections.Counter(), []
for name, s, e, trips in segs:
    c = collections.Counter(classify(op) for _, op, _ in ins[s:e])
    table.append((name, e - s, trips, c))
    for k, v in c.items():
        tot[k] += v * trips
        if k not in classes:
            classes.append(k)
print("# One Poseidon2 permutation (width 16) as compiled for gfx950: instructions by segment and class (`tools/p2_isa_buckets.py`)\n")
print("| segment | static | trips | " + " | ".join(classes) + " |")
print("|---|---|---|" + "---|" * len(classes))
for name, n, trips, c in table:
    print("| %s | %d | %d | " % (name, n, trips) + " | ".join(str(c.get(k, 0)) for k in classes) + " |")
print("| **dynamic, per permutation** | | | " + " | ".join("**%d**" % tot[k] for k in classes) + " |")
valu = sum(v for k, v in tot.items() if k.startswith(("multiply", "v_", "plain", "other VALU")))
print("\nVALU instructions per permutation: %d, of them multiplies %d; scalar / wait %d." % (valu, tot[classes[[k.startswith("multiply") for k in classes].index(True)]],
                                                                                            sum(tot.values()) - valu))
