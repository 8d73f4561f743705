#!/usr/bin/env python
"""filter_record_batch latency at query-engine batch sizes: N nullable Int64 columns x rows, 10 % selectivity.
mirror = K.filter_record_batch through the Python mirror (view structs, result objects, finalizers); C ABI = the same
ah_filter_record_batch call on prebuilt views + ah_array_release per column (what a Rust / C++ host pays); per-column =
one predicate + ah_filter_predicate_apply per column (the general two-pass path, AH_FILTER_SMALL has no say there)."""
import ctypes as C
import sys, time
import numpy as np
sys.path.insert(0, ".")
import arrow_rs_amd as A
from arrow_rs_amd import compute as K
sys.path.insert(0, "tests")
from orc import HostArray

ctx = A.Context(0)
rng = np.random.default_rng(1)
from arrow_rs_amd import _lib as L
print("| rows | columns | mirror us | C ABI us | per-column us |")
print("|---|---|---|---|---|")
for rows in (8192, 65536, 1 << 20):
    for ncol in (2, 8, 16):
        cols = [HostArray(A.Int64, rng.integers(-2**62, 2**62, rows), rng.random(rows) < 0.9).to_device(ctx) for _ in range(ncol)]
        mask = HostArray(A.Boolean, rng.random(rows) < 0.1).to_device(ctx)
        rb = A.RecordBatch([f"c{i}" for i in range(ncol)], cols)
        def fused():
            return K.filter_record_batch(rb, mask)
        def percol():
            p = K.FilterBuilder.new(mask).build()
            return [p.filter(c) for c in cols]
        views = (L.ArrayView * ncol)(*[c.view() for c in cols])
        mv = mask.view()
        outs = (L.ArrayOut * ncol)()
        nrows = C.c_int64()
        def cabi():
            st = ctx.lib.ah_filter_record_batch(ctx.handle, ncol, views, C.byref(mv), outs, C.byref(nrows))
            assert st == 0
            for i in range(ncol):
                ctx.lib.ah_array_release(ctx.handle, C.byref(outs[i]))
        res = []
        for fn in (fused, cabi, percol):
            for _ in range(20): fn()
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(200): fn()
            ctx.synchronize()
            res.append((time.perf_counter() - t0) / 200 * 1e6)
        print(f"| {rows} | {ncol} | {res[0]:.0f} | {res[1]:.0f} | {res[2]:.0f} |")
