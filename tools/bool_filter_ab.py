#!/usr/bin/env python
"""A/B of filter_boolean at low selectivity: the bit-only sparse kernel (round 4) against the tiled W == 0 kernel, 1e9 Boolean
rows with 10 % nulls.  usage (GPU box): python tools/bool_filter_ab.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_rs_amd as A  # noqa: E402
from arrow_rs_amd import compute as K  # noqa: E402
import bench as B  # noqa: E402

ctx = A.Context(0)
A.set_default_context(ctx)
n = 1_000_000_000
vals = B.gen_predicate(A, ctx, n, 7, 0.5, 0)
valid = B.gen_predicate(A, ctx, n, 8, 0.9, 0)
col = A.Array(ctx, A.Boolean, n, vals.values, 0, valid.values, 0, n - B.count_bits(ctx, valid.values, n))
print("| selected | tiled ms | sparse ms | heuristic ms |")
print("|---|---|---|---|")
for sel in (0.0001, 0.001, 0.01, 0.03, 0.1):
    pred = K.FilterBuilder(B.gen_predicate(A, ctx, n, 44, sel, 0)).build()
    row = []
    for force in ("0", "1", None):
        if force is None:
            os.environ.pop("AH_FILTER_SPARSE", None)
        else:
            os.environ["AH_FILTER_SPARSE"] = force
        for _ in range(2):
            pred.filter(col)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            pred.filter(col)
        ctx.synchronize()
        row.append((time.perf_counter() - t0) / 10 * 1e3)
    print(f"| {sel:.2%} | {row[0]:.3f} | {row[1]:.3f} | {row[2]:.3f} |")
