"""Guard-mode micro probe: cost of one guarded alloc/free, and two tiny ops whose results are checked on the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import ctypes as C
import arrow_rs_amd as A
from arrow_rs_amd import compute as K
from orc import HostArray
ctx = A.Context(0)
A.set_default_context(ctx)
lib = ctx.lib
t0 = time.perf_counter()
for i in range(50):
    p = C.c_void_p()
    ctx.check(lib.ah_device_alloc(ctx.handle, 1000 + i, C.byref(p)))
    lib.ah_device_free(ctx.handle, p)
print("alloc+free us:", (time.perf_counter() - t0) / 50 * 1e6, flush=True)
for n in (1, 5, 64, 1000):
    v = np.arange(n, dtype=np.int64)
    valid = (np.arange(n) % 3) != 0
    h = HostArray(A.Int64, v, valid)
    d = h.to_device()
    back = HostArray.from_device(d)
    print(n, "roundtrip ok:", np.array_equal(back.values, v), np.array_equal(back.valid, valid), flush=True)
    t0 = time.perf_counter()
    s = K.add_wrapping(d, d)
    hs = HostArray.from_device(s)
    print(n, "add ok:", np.array_equal(np.asarray(hs.values)[valid], (v + v)[valid]), "valid ok:", np.array_equal(hs.valid, valid),
          "ms", (time.perf_counter() - t0) * 1e3, flush=True)
    if not np.array_equal(hs.valid, valid):
        print("  got", hs.valid[:16], "exp", valid[:16])
