#!/usr/bin/env python3
"""every operator of csrc/hal.hip (the RISC Zero `Hal` set) timed at RISC Zero's sizes --
2^20-row segments of 128 columns, 2^22 x 4-word extension vectors -- with HIP events on the context's stream, GB/s against the
bytes each operator must move, fraction of the 8 TB/s peak.  Writes a markdown table (stdout, and --out).
    python3 tools/hal_ops_time.py --out gpurun_out/r04_hal_ops.md"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from time_ops import Timer  # noqa: E402
from zktls_amd.device import Context  # noqa: E402

P = 2013265921


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--log-rows", type=int, default=20)
    ap.add_argument("--cols", type=int, default=128)
    args = ap.parse_args()
    ctx = Context(0)
    t = Timer(ctx.stream)
    rows, cols = 1 << args.log_rows, args.cols
    rng = np.random.default_rng(1)
    lines = []

    def row(name, kernel, ms, bytes_moved, note=""):
        gbs = bytes_moved / ms / 1e6
        lines.append("| `%s` | %s | %.3f | %.0f | %.1f | %.3f | %s |" % (name, kernel, ms, bytes_moved / 2**20, gbs, gbs / 8000.0, note))

    # element-wise
    n = 1 << 26
    a, b, o = ctx.fill_uniform(1, 20, 64), ctx.fill_uniform(2, 20, 64), ctx.alloc(n)
    ms, _ = t.time(lambda: ctx.eltwise_add(a, b, out=o), reps=10, warm=2)
    row("eltwise_add", "hal_add_kernel", ms, 12.0 * n, "2^26 elements, read 8 + write 4 B")
    ms, _ = t.time(lambda: ctx.eltwise_zeroize(a), reps=10, warm=2)
    row("eltwise_zeroize", "hal_zeroize_kernel", ms, 4.0 * n, "2^26 elements, read 4 B (nothing to rewrite)")
    cnt, to_add = 1 << 22, 4
    ms, _ = t.time(lambda: ctx.eltwise_sum_ext(a, cnt, out=o), reps=10, warm=2)
    row("eltwise_sum_ext", "hal_sum_ext_kernel", ms, 16.0 * cnt * (to_add + 1), "2^22 extension elements x 4 summands")
    # zk_shift: 128 polynomials of 2^20
    polys = ctx.fill_uniform(3, args.log_rows, cols)
    ms, _ = t.time(lambda: ctx.zk_shift(polys, cols, args.log_rows, 3), reps=10, warm=2)
    row("zk_shift", "hal_zk_shift_kernel", ms, 8.0 * rows * cols, "%d polynomials of 2^%d, in place" % (cols, args.log_rows))
    # mix_poly_coeffs: 128 inputs of 2^20 into 3 combos
    ncombo = 3
    combos = ctx.from_raw(rng.integers(0, ncombo, cols, dtype=np.uint32))
    mo = ctx.alloc(4 * ncombo * rows)
    start, mix = rng.integers(0, P, 4, dtype=np.uint32), rng.integers(0, P, 4, dtype=np.uint32)
    for ef in (0, 1):
        ms, _ = t.time(lambda: ctx.mix_poly_coeffs(mo, start, mix, polys, combos, cols, rows, ef), reps=10, warm=2)
        row("mix_poly_coeffs (ext %d)" % ef, "hal_mix_powers + hal_mix_plan + hal_mix_sorted_kernel", ms, 4.0 * rows * cols + 32.0 * ncombo * rows,
            "%d inputs of 2^%d into %d combos: read 4 B per input element + one read-modify-write per combo" % (cols, args.log_rows, ncombo))
    # batch_evaluate_any: every polynomial at two points
    nev = 2 * cols
    which = ctx.from_raw(np.arange(nev, dtype=np.uint32) % cols)
    xs = ctx.from_numpy(rng.integers(0, P, (nev, 4), dtype=np.uint32))
    eo = ctx.alloc(4 * nev)
    for ef in (0, 1):
        ms, _ = t.time(lambda: ctx.batch_evaluate_any(polys, args.log_rows, which, xs, ef, out=eo), reps=5, warm=1)
        row("batch_evaluate_any (ext %d)" % ef, "hal_eval_tables + hal_batch_evaluate_any_kernel", ms, 4.0 * rows * nev, "%d evaluations of 2^%d-coefficient polynomials" % (nev, args.log_rows))
    # gather_sample: one row of a [size][stride] matrix
    ms, _ = t.time(lambda: ctx.gather_sample(polys, 5, rows, cols), reps=10, warm=2)
    row("gather_sample", "hal_gather_sample_kernel", ms, 8.0 * rows, "2^%d samples at stride %d words (a 4-byte gather: 64 B lines fetched for 4 B used)" % (args.log_rows, cols))
    # prefix products of 2^22 extension elements
    pn = 1 << 22
    pv = ctx.fill_uniform(5, 22, 4)
    for ef in (0, 1):
        ms, _ = t.time(lambda: ctx.prefix_products_ext(pv, ef), reps=5, warm=1)
        row("prefix_products_ext (ext %d)" % ef, "hal_scan_blocks / _totals / _apply", ms, 16.0 * pn * 3, "2^22 extension elements: read twice, written once")
    # SHA-256 rows and fold
    dig = ctx.alloc(8 * rows)
    ms, _ = t.time(lambda: ctx.hash_rows_sha256(polys, cols, rows, out=dig), reps=3, warm=1)
    row("hash_rows_sha256", "hal_hash_rows_sha256_kernel", ms, 4.0 * rows * cols + 32.0 * rows, "2^%d rows x %d columns (integer-ALU bound: 9 compressions per row)" % (args.log_rows, cols))
    par = ctx.alloc(4 * rows)
    ms, _ = t.time(lambda: ctx.hash_fold_sha256(dig, rows // 2, out=par), reps=5, warm=1)
    row("hash_fold_sha256", "hal_hash_fold_sha256_kernel", ms, 96.0 * (rows // 2), "2^%d parents (2 compressions each)" % (args.log_rows - 1))
    head = ["# RISC Zero `Hal` operators (csrc/hal.hip) on one MI355X, HIP events", "",
            "| operator | kernel | ms | MiB moved | GB/s | of 8 TB/s | workload |", "|---|---|---|---|---|---|---|"]
    text = "\n".join(head + lines) + "\n"
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(text)
    ctx.close()


if __name__ == "__main__":
    main()
