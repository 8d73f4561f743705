"""Wall-clock timings (synchronous C-ABI calls, median of 5) of the SURVEY §8f rows that have no bench.py
workload: RowSelection algebra at 2^30 rows, IPC encode/decode of a 2^28-row 3-column batch.  One JSON line."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_rs_amd as A  # noqa: E402
from arrow_rs_amd import ipc  # noqa: E402
from arrow_rs_amd.selection import RowSelection  # noqa: E402
import bench  # noqa: E402

ctx = A.Context(0)
A.set_default_context(ctx)


def med(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return round(statistics.median(ts) * 1e3, 3)


res = {}
n = 1 << 30
buf = ctx.alloc(n // 8)
ctx.check(ctx.lib.ah_gen_bernoulli_bits(ctx.handle, buf.ptr, n, 11, 0.1, 0))
s = RowSelection(A.Array(ctx, A.Boolean, n, A.array._RawMem(buf.ptr, n // 8, buf)))
k = s.row_count()
ob = ctx.alloc((k + 63) // 64 * 8)
ctx.check(ctx.lib.ah_gen_bernoulli_bits(ctx.handle, ob.ptr, k, 12, 0.5, 0))
o = RowSelection(A.Array(ctx, A.Boolean, k, A.array._RawMem(ob.ptr, ob.nbytes, ob)))
res["selection_rows"] = n
res["and_then_ms"] = med(lambda: s.and_then(o))
res["intersection_ms"] = med(lambda: s.intersection(s))
res["boundaries_ms"] = med(lambda: s.boundaries())
res["boundaries"] = s.boundaries().length
res["find_nth_ms"] = med(lambda: s._find_nth(k // 2))
b = s.boundaries()
res["from_boundaries_ms"] = med(lambda: RowSelection.from_boundaries.__func__(RowSelection, [], 0, ctx) if False else
                                ctx.check(ctx.lib.ah_selection_from_boundaries(
                                    ctx.handle, __import__("ctypes").byref(b.view()), n,
                                    __import__("ctypes").byref(A._lib.ArrayOut()))))
del s, o, b, buf, ob

m = 1 << 28
a = bench.gen_i64_column(A, ctx, m, 42, 0.9, 0)
f = bench.gen_f64_column(A, ctx, m, 52, 0.9, 0)
p = bench.gen_predicate(A, ctx, m, 44, 0.1, 0)
rb = A.RecordBatch(["a", "f", "p"], [a, f, p], m)
meta, body = ipc.encode_batch(rb)
res["ipc_rows"], res["ipc_body_bytes"] = m, body.nbytes
res["ipc_encode_ms"] = med(lambda: ipc.encode_batch(rb))
res["ipc_encode_GBps"] = round(2 * body.nbytes / (res["ipc_encode_ms"] * 1e-3) / 1e9, 1)  # read + write
sch = ipc.Schema.of(rb)
res["ipc_decode_ms"] = med(lambda: ipc.decode_batch(meta, body.ptr, body.nbytes, sch, ctx, keepalive=(body,)))
del a, f, p, rb, body

# string predicates on the config-4 cast output (LargeUtf8, ~9 bytes per row): offsets + bytes read once, 1 bit/row out
from arrow_rs_amd import compute as K  # noqa: E402
ns = 1 << 27
src = bench.gen_i64_column(A, ctx, ns, 42, 0.9, 0, -10**6, 10**6)
scol = K.cast(K.cast(src, A.Float64), A.LargeUtf8)
res["like_rows"], res["like_text_bytes"] = ns, scol.values.nbytes
for name, fn, pat in (("like_contains", K.like, "%99%"), ("like_prefix", K.like, "-1%"), ("like_general", K.like, "%1_3%.0"),
                      ("starts_with", K.starts_with, "12"), ("length", lambda c, _p: K.length(c), None)):
    res[name + "_ms"] = med(lambda: fn(scol, pat))
alg = scol.values.nbytes + (ns + 1) * 8 + ns // 8

# Utf8 -> numeric (cast_parse.hip): the config-4 text parsed back; offsets + text + validity in, 8 B + 1 bit out
res["parse_f64_ms"] = med(lambda: K.cast(scol, A.Float64))
res["parse_f64_GBps"] = round((alg + ns * 8 + ns // 8) / (res["parse_f64_ms"] * 1e-3) / 1e9, 1)
itext = K.cast(src, A.LargeUtf8)
res["parse_i64_text_bytes"] = itext.values.nbytes
res["parse_i64_ms"] = med(lambda: K.cast(itext, A.Int64))
res["parse_i64_GBps"] = round((itext.values.nbytes + (ns + 1) * 8 + ns // 8 + ns * 8 + ns // 8) / (res["parse_i64_ms"] * 1e-3) / 1e9, 1)
res["parse_i32_ms"] = med(lambda: K.cast(itext, A.Int32))
wide = K.cast(K.cast(bench.gen_i64_column(A, ctx, ns, 43, 0.9, 0), A.Float64), A.LargeUtf8)  # 17-digit exponent forms
res["parse_f64_wide_text_bytes"] = wide.values.nbytes
res["parse_f64_wide_ms"] = med(lambda: K.cast(wide, A.Float64))
del itext, wide
res["like_contains_GBps"] = round(alg / (res["like_contains_ms"] * 1e-3) / 1e9, 1)
print(json.dumps(res))
