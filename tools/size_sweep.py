"""Wall time per call vs array size (synchronous calls, inputs resident in HBM): shows where the fixed per-call cost
(launches, the host wait) stops mattering.  Two timings per kernel: through the Python mirror (`K.filter(...)`: view
structs, result objects and finalizers included) and straight through the C ABI (ctypes call on prebuilt views +
ah_array_release: what a Rust / C++ host pays).  Also the 1-core oracle on the same rows, for the crossover."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import arrow_rs_amd as A
from arrow_rs_amd import compute as K
from arrow_rs_amd import _lib as L
import bench as B

ctx = A.Context(0)
A.set_default_context(ctx)
lib, h = ctx.lib, ctx.handle
try:
    import orc
    oracle = orc.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so"))
except Exception:  # noqa: BLE001
    oracle = None
SIZES = [10**4, 10**5, 10**6, 10**7, 10**8, 10**9] if len(sys.argv) < 2 else [int(float(x)) for x in sys.argv[1:]]
print("| rows | filter us (mirror / C ABI) | take 10% us | add_wrapping us | lt us | filter Mrows/s (C ABI) | 1-core oracle filter us |")
print("|---|---|---|---|---|---|---|")
cross = None
for n in SIZES:
    col = B.gen_i64_column(A, ctx, n, 42, 0.9, 0)
    col2 = B.gen_i64_column(A, ctx, n, 52, 0.9, 0)
    pred = B.gen_predicate(A, ctx, n, 44, 0.1, 0)
    m = max(1, n // 10)
    ib = ctx.alloc(m * 4)
    ctx.check(lib.ah_gen_uniform_u32(h, ib.ptr, m, 45, n, 0))
    idx = B.mk_array(A, ctx, A.UInt32, m, ib)
    vc, vc2, vp, vi = col.view(), col2.view(), pred.view(), idx.view()

    def raw(call):
        def fn():
            out = L.ArrayOut()
            st = call(out)
            assert st == 0, st
            lib.ah_array_release(h, C.byref(out))
        return fn
    pairs = [
        (lambda: K.filter(col, pred), raw(lambda o: lib.ah_filter(h, C.byref(vc), C.byref(vp), C.byref(o)))),
        (lambda: K.take(col, idx), raw(lambda o: lib.ah_take(h, C.byref(vc), C.byref(vi), 0, C.byref(o)))),
        (lambda: K.add_wrapping(col, col2), raw(lambda o: lib.ah_arith_binary(h, 1, C.byref(vc), 0, C.byref(vc2), 0, C.byref(o)))),
        (lambda: K.lt(col, col2), raw(lambda o: lib.ah_compare(h, 2, C.byref(vc), 0, C.byref(vc2), 0, C.byref(o)))),
    ]
    res = []
    reps = 300 if n <= 10**6 else (30 if n <= 10**8 else 8)
    for mirror, cabi in pairs:
        r2 = []
        for fn in (mirror, cabi):
            for _ in range(5):
                fn()
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            ctx.synchronize()
            r2.append((time.perf_counter() - t0) / reps * 1e6)
        res.append(r2)
    cpu_us = None
    if oracle is not None and n <= 10**7:
        vals = oracle.gen_i64(n, 42, -2**63, 2**63 - 1)
        valid = oracle.gen_bits(n, 43, 0.9)
        mask = oracle.gen_bits(n, 44, 0.1)
        hv, hm = orc._Held(orc.HostArray(A.Int64, vals, valid)), orc._Held(orc.HostArray(A.Boolean, mask))
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 0.2:
            o = orc.Out()
            oracle.lib.orc_filter(C.byref(hv.view), C.byref(hm.view), C.byref(o))
            oracle.lib.orc_release(C.byref(o))
            k += 1
        cpu_us = (time.perf_counter() - t0) / k * 1e6
        if cross is None and res[0][1] < cpu_us:
            cross = n
    f = lambda r: f"{r[0]:.1f} / {r[1]:.1f}"  # noqa: E731
    print(f"| {n:.0e} | {f(res[0])} | {f(res[1])} | {f(res[2])} | {f(res[3])} | {n / res[0][1]:.0f} | "
          f"{'%.1f' % cpu_us if cpu_us else '-'} |")
    del col, col2, pred, idx, ib
print(f"crossover_rows (first size where the C-ABI filter call beats the 1-core oracle): {cross}")
