"""Time the ordered (checked) aggregate kernels on the bench column; prints avg kernel ms (library HIP events)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_rs_amd as A  # noqa: E402
from arrow_rs_amd.compute import aggregate as G  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ctx = A.Context(0)
A.set_default_context(ctx)
res = {}
for name, lo, hi in (("small", -10**6, 10**6), ("ones", 1, 1)):
    col = bench.gen_i64_column(A, ctx, n, 42, 0.9, 0, lo, hi)
    for fn in ("sum_checked", "product_checked", "sum"):
        if fn == "product_checked" and name == "small":
            continue  # overflows at once: error path
        f = getattr(G, fn)
        f(col)
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(5):
            r = f(col)
        k = "aggregate" if fn == "sum" else "aggregate_checked"
        ms, cnt = ctx.profile_get(k)
        ctx.profile(False)
        res[f"{fn}[{name}]"] = {"ms": round(ms / cnt, 4), "GBps": round((n * 8 + n / 8) / (ms / cnt * 1e-3) / 1e9, 1), "result": int(r)}
print(json.dumps(res))
