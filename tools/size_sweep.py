"""Wall time per call vs array size (synchronous C-ABI calls, inputs resident in HBM): shows where
the fixed per-call cost (count read-back, null-count read-back, allocations) stops mattering."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_rs_amd as A
from arrow_rs_amd import compute as K
import bench as B

ctx = A.Context(0)
A.set_default_context(ctx)
print("| rows | filter ms | take(10%) ms | add_wrapping ms | lt ms | filter Mrows/s |")
print("|---|---|---|---|---|---|")
for n in [10**4, 10**5, 10**6, 10**7, 10**8, 10**9]:
    col = B.gen_i64_column(A, ctx, n, 42, 0.9, 0)
    col2 = B.gen_i64_column(A, ctx, n, 52, 0.9, 0)
    pred = B.gen_predicate(A, ctx, n, 44, 0.1, 0)
    m = max(1, n // 10)
    ib = ctx.alloc(m * 4)
    ctx.check(ctx.lib.ah_gen_uniform_u32(ctx.handle, ib.ptr, m, 45, n, 0))
    idx = B.mk_array(A, ctx, A.UInt32, m, ib)
    res = []
    for fn in (lambda: K.filter(col, pred), lambda: K.take(col, idx), lambda: K.add_wrapping(col, col2),
               lambda: K.lt(col, col2)):
        for _ in range(3):
            fn()
        reps = 200 if n <= 10**6 else (30 if n <= 10**8 else 8)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        res.append((time.perf_counter() - t0) / reps * 1e3)
    print(f"| {n:.0e} | {res[0]:.3f} | {res[1]:.3f} | {res[2]:.3f} | {res[3]:.3f} | {n / res[0] / 1e3:.0f} |")
    del col, col2, pred, idx, ib
