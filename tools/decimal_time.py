"""Kernel times (ah_profile_* HIP events) of Decimal128 arithmetic on 2^26 rows, 10 % nulls on each side: one JSON line with
ms per launch and algorithmic GB/s (two 16-byte operands + validity in, one 16-byte result + validity out = 48.4 B/row)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_rs_amd as A  # noqa: E402
from arrow_rs_amd import compute as K  # noqa: E402

ctx = A.Context(0)
A.set_default_context(ctx)
n = int(os.environ.get("ROWS", 1 << 26))
rng = np.random.default_rng(1)


def column(t, lo, hi, seed):
    v = np.random.default_rng(seed).integers(lo, hi, n, dtype=np.int64)
    v[v == 0] = 1
    vals = np.zeros(n, dtype=t.np_dtype)
    vals["lo"] = v.view(np.uint64)
    vals["hi"] = np.where(v < 0, -1, 0)
    return A.Array.from_numpy(vals, np.random.default_rng(seed + 1).random(n) < 0.9, t, ctx)


a = column(A.Decimal128(20, 4), -10**17, 10**17, 2)
b = column(A.Decimal128(18, 2), -10**15, 10**15, 4)
c = column(A.Decimal128(20, 4), -10**17, 10**17, 6)
res = {"rows": n}
for name, fn, l, r in (("add_same_scale", K.add, a, c), ("add_rescaled", K.add, a, b), ("mul", K.mul, a, b), ("div", K.div, a, b),
                       ("rem", K.rem, a, b)):
    fn(l, r)
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(5):
        fn(l, r)
    ctx.synchronize()
    ms, launches = ctx.profile_get("arith_decimal")
    ctx.profile(False)
    per = ms / launches
    alg = n * 48 + 3 * (n // 8)
    res[name] = {"ms": round(per, 4), "algorithmic_GBps": round(alg / (per * 1e-3) / 1e9, 1), "frac_of_8TBps": round(alg / (per * 1e-3) / 1e9 / 8000, 4)}
print(json.dumps(res))
