"""Kernel times (the library's HIP events on its launch stream, ah_profile_*) of the temporal casts on 2^29 rows of
Timestamp(µs) with 10 % nulls: one JSON line with ms per launch and algorithmic GB/s (input values + validity read once,
output values + validity written once) against the 8 TB/s HBM peak.  gpurun -- python tools/cast_temporal_time.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_rs_amd as A  # noqa: E402
from arrow_rs_amd import compute as K  # noqa: E402
import bench  # noqa: E402

ctx = A.Context(0)
A.set_default_context(ctx)
n = int(os.environ.get("ROWS", 1 << 29))
col = bench.gen_i64_column(A, ctx, n, 42, 0.9, 0, lo=-4 * 10**15, hi=4 * 10**15)  # +-126 years of microseconds
col.data_type = A.TimestampMicrosecond
res = {"rows": n}
# the numeric cast of the same shape (8 B in, 8 B out), timed in the same process as the yardstick
col.data_type = A.Int64
K.cast(col, A.Float64)
ctx.profile(True)
ctx.profile_reset()
for _ in range(5):
    K.cast(col, A.Float64)
ctx.synchronize()
ms, launches = ctx.profile_get("cast_numeric")
ctx.profile(False)
res["i64_to_f64_numeric_cast_ms"] = round(ms / launches, 4)
col.data_type = A.TimestampMicrosecond
for name, to, out_w in (("ts_us_to_date32", A.Date32, 4), ("ts_us_to_time64_us", A.Time64Microsecond, 8),
                        ("ts_us_to_ts_ns_checked", A.TimestampNanosecond, 8), ("ts_us_to_ts_s_div", A.TimestampSecond, 8),
                        ("ts_us_to_ts_us_zone_adjust", A.Timestamp(A.MICROSECOND, "+05:45"), 8)):
    K.cast(col, to)  # warm the pool
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(5):
        K.cast(col, to)
    ctx.synchronize()
    ms, launches = ctx.profile_get("cast_temporal")
    ctx.profile(False)
    per = ms / launches
    alg = n * (8 + out_w) + 2 * (n // 8)
    res[name] = {"ms": round(per, 4), "algorithmic_GBps": round(alg / (per * 1e-3) / 1e9, 1),
                 "frac_of_8TBps": round(alg / (per * 1e-3) / 1e9 / 8000, 4), "launches": launches}
print(json.dumps(res))
