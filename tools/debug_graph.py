import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import arrow_rs_amd as A
from arrow_rs_amd import compute as K, _lib as L
hip = C.CDLL("libamdhip64.so")
hip.hipGetLastError.restype = C.c_int
hip.hipStreamIsCapturing.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
hip.hipGetErrorString.restype = C.c_char_p
ctx = A.Context(0)
lib, h = ctx.lib, ctx.handle
lib.ah_context_stream.restype = C.c_void_p
st_ptr = lib.ah_context_stream(h)
def cap():
    s = C.c_int(-1)
    r = hip.hipStreamIsCapturing(st_ptr, C.byref(s))
    return (r, s.value)
a = A.Array.from_numpy(np.arange(1000, dtype=np.int64), ctx=ctx)
b = A.Array.from_numpy(np.arange(1000, dtype=np.int64), ctx=ctx)
av, bv = a.view(), b.view()
print("capturing before", cap(), flush=True)
print("begin", lib.ah_graph_begin(h), cap(), flush=True)
out = L.ArrayOut()
print("checked add in capture ->", lib.ah_arith_binary(h, 0, C.byref(av), 0, C.byref(bv), 0, C.byref(out)), lib.ah_last_error(h), cap(), flush=True)
g = C.c_void_p()
print("end ->", lib.ah_graph_end(h, C.byref(g)), lib.ah_last_error(h), cap(), "lasterr", hip.hipGetLastError(), flush=True)
if g.value:
    lib.ah_graph_destroy(h, g)
out2 = L.ArrayOut()
print("add_wrapping after ->", lib.ah_arith_binary(h, 1, C.byref(av), 0, C.byref(bv), 0, C.byref(out2)), lib.ah_last_error(h), cap(), flush=True)
# second: clean capture
print("begin2", lib.ah_graph_begin(h), cap(), flush=True)
o3 = L.ArrayOut()
print("wrapping add in capture ->", lib.ah_arith_binary(h, 1, C.byref(av), 0, C.byref(bv), 0, C.byref(o3)), cap(), flush=True)
print("end2 ->", lib.ah_graph_end(h, C.byref(g)), lib.ah_last_error(h), cap(), flush=True)
print("launch", lib.ah_graph_launch(h, g), "sync", lib.ah_synchronize(h), flush=True)
print("done", flush=True)
