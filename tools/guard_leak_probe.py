"""How a long guard-mode process grows: RSS / VmSize / host mappings per 5 000 guarded alloc + free pairs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import arrow_rs_amd as A
ctx = A.Context(0)
lib = ctx.lib
def stat():
    d = {}
    for l in open("/proc/self/status"):
        if l.startswith(("VmRSS", "VmSize", "VmPTE")):
            d[l.split(":")[0]] = int(l.split()[1]) // 1024
    d["maps"] = sum(1 for _ in open("/proc/self/maps"))
    return d
print("start", stat(), flush=True)
t0 = time.perf_counter()
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    for i in range(5000):
        p = C.c_void_p()
        ctx.check(lib.ah_device_alloc(ctx.handle, 1000 + (i % 50) * 4096, C.byref(p)))
        lib.ah_device_free(ctx.handle, p)
    print(rnd, stat(), round(time.perf_counter() - t0, 1), "s", flush=True)
