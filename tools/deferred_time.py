"""Per-batch cost of a kernel chain over engine-sized batches, synchronous vs deferred mode
(ah_context_set_deferred).  The chain is lt(a, scalar) -> filter(a), filter(b) -> add_wrapping -> cast
to Float64 (5 kernels + the predicate count, which is the one host read-back a filter needs).
Also the pure elementwise chain mul_wrapping -> add_wrapping -> cast (no read-back at all when deferred).
Two call paths: the Python mirror (ctypes; what the tests use) and the raw C ABI driven from C-speed loops is
not available here, so the Python overhead per call (~2-3 us of ctypes) is inside both columns.  One JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_rs_amd as A  # noqa: E402
from arrow_rs_amd import compute as K  # noqa: E402
import bench  # noqa: E402

ctx = A.Context(0)
A.set_default_context(ctx)
res = {"rows": {}}
for n in (8192, 65536, 1 << 20, 1 << 24):
    a = bench.gen_i64_column(A, ctx, n, 42, 0.9, 0, -1000, 1000)
    b = bench.gen_i64_column(A, ctx, n, 43, 0.9, 0)
    sc = A.Scalar.new(100, A.Int64, ctx)

    def pipeline():
        pred = K.lt(a, sc)
        fa, fb = K.filter(a, pred), K.filter(b, pred)
        return K.cast(K.add_wrapping(fa, fb), A.Float64)

    def elementwise():
        return K.cast(K.add_wrapping(K.mul_wrapping(a, b), a), A.Float64)

    def timeit(fn, iters):
        for _ in range(20):
            fn()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            r = fn()
        ctx.synchronize()
        del r
        return round((time.perf_counter() - t0) / iters * 1e6, 1)

    iters = 400 if n <= 1 << 20 else 50
    row = {}
    for name, fn in (("pipeline", pipeline), ("elementwise", elementwise)):
        ctx.set_deferred(False)
        row[name + "_sync_us"] = timeit(fn, iters)
        ctx.set_deferred(True)
        row[name + "_deferred_us"] = timeit(fn, iters)
        ctx.set_deferred(False)
    res["rows"][str(n)] = row
print(json.dumps(res))
