#!/usr/bin/env python3
"""One-off source transformation (kept for the record): give every listed __global__ kernel of a .hip file a batched twin.

    __global__ void ATTRS NAME(PARAMS) { BODY }
becomes
    __device__ __forceinline__ void NAME_body(PARAMS) { BODY }
    __global__ void ATTRS NAME(PARAMS) { NAME_body(names...); }
    struct NAME_bargs { PARAMS as fields; static NAME_bargs make(PARAMS); };
    __global__ void ATTRS NAME_batch(const NAME_bargs* zk_arr) { const NAME_bargs zk_b = zk_arr[blockIdx.z]; NAME_body(zk_b.names...); }
and every  hipLaunchKernelGGL(NAME..., grid, block, lds, stream, args...)  becomes  ZK_LAUNCH(NAME..., NAME_batch..., NAME_bargs, ...).
blockIdx.z is the batch index: B proofs of one shape advance through ONE launch per stage (batch.h).
usage: tools/batchify.py file.hip name1 name2 ..."""
import re
import sys


def split_params(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def param_name(p):
    m = re.search(r"(\w+)\s*$", p)
    return m.group(1)


def match_brace(text, i):
    """index just past the brace that closes text[i] == '{' (comments and strings of this code base hold no unbalanced braces)"""
    depth = 0
    while i < len(text):
        c = text[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1
    raise ValueError("unbalanced")


def transform(text, name):
    # definition: optional template line(s) directly above
    pat = re.compile(r"(?P<tmpl>(?:template\s*<[^>]*>\s*\n(?:#[^\n]*\n)*)?)__global__ void (?P<attrs>(?:__launch_bounds__\([^)]*\)\s*|__attribute__\(\([^\n]*?\)\)\)\s*)*)" + re.escape(name) + r"\((?P<params>[^)]*)\)\s*\{", re.S)
    m = pat.search(text)
    if not m:
        raise SystemExit("definition of %s not found" % name)
    end = match_brace(text, m.end() - 1)
    body = text[m.end() - 1:end]
    tmpl, attrs, params = m.group("tmpl"), m.group("attrs").strip(), " ".join(m.group("params").split())
    plist = split_params(params)
    names = [param_name(p) for p in plist]
    targs = ""
    if tmpl.strip():
        inner = re.search(r"template\s*<(.*)>", tmpl, re.S).group(1)
        tn = []
        for t in split_params(inner):
            t = t.split("=")[0].strip()
            tn.append(param_name(t))
        targs = "<" + ", ".join(tn) + ">"
    fields = "; ".join(re.sub(r"\s*__restrict__\s*", " ", p) for p in plist) + ";"
    attrs_sp = (attrs + " ") if attrs else ""
    fwd = re.search(r"^__global__ void " + re.escape(name) + r"\([^;{]*\);\n", text[:m.start()], re.M)
    # struct parameters reach the body by const reference: the batched twin then indexes their arrays in the argument ring itself
    # (a by-value copy of a struct whose arrays are indexed at run time would live in scratch)
    def by_ref(p):
        ty = p[:p.rfind(param_name(p))].strip()
        prim = ty.endswith("*") or ty.endswith("__restrict__") or re.sub(r"\bconst\b", "", ty).strip() in (
            "uint32_t", "uint64_t", "int", "bool", "size_t", "unsigned", "uint8_t", "int64_t", "uint16_t")
        return p if prim or ty.endswith("&") else "const " + ty + "& " + param_name(p)
    body_params = ", ".join(by_ref(p) for p in plist)
    new = (tmpl + "__device__ __forceinline__ void %s_body(%s) %s\n" % (name, body_params, body) +
           tmpl + "__global__ void %s%s(%s) { %s_body%s(%s); }\n" % (attrs_sp, name, params, name, targs, ", ".join(names)))
    plain = ", ".join(re.sub(r"\s*__restrict__\s*", " ", p) for p in plist)
    struct = "struct %s_bargs { %s static %s_bargs make(%s) { return %s_bargs{%s}; } };\n" % (name, fields, name, plain, name, ", ".join(names))
    batch = (tmpl + "__global__ void %s%s_batch(const %s_bargs* __restrict__ zk_arr) { const %s_bargs& zk_b = zk_arr[blockIdx.z]; %s_body%s(%s); }\n"
             % (attrs_sp, name, name, name, name, targs, ", ".join("zk_b." + n for n in names)))
    if fwd:
        new += batch
        text = text[:m.start()] + new + text[end:]
        # complete struct + declaration of the batch kernel next to the forward declaration
        decl = struct + "__global__ void %s_batch(const %s_bargs* __restrict__ zk_arr);\n" % (name, name)
        text = text[:fwd.end()] + decl + text[fwd.end():]
    else:
        new += struct + batch
        text = text[:m.start()] + new + text[end:]
    # launch sites
    def repl(mm):
        k = mm.group(1)
        paren = k.startswith("(")
        inner = k[1:-1] if paren else k
        base = inner.split("<")[0]                                     # possibly ns::name
        kb = base + "_batch" + inner[len(base):]
        if paren:
            kb = "(" + kb + ")"
        return "ZK_LAUNCH(%s, %s, %s_bargs, " % (k, kb, base)
    text = re.sub(r"hipLaunchKernelGGL\((\(?(?:\w+::)*" + re.escape(name) + r"(?:<[^()]*>)?\)?), ", repl, text)
    return text


if __name__ == "__main__":
    path = sys.argv[1]
    src = open(path).read()
    for nm in sys.argv[2:]:
        src = transform(src, nm)
    if '#include "batch.h"' not in src:
        src = src.replace('#include "kernels.h"', '#include "kernels.h"\n#include "batch.h"', 1)
    open(path, "w").write(src)
