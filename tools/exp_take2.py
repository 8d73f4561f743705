import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_rs_amd as A
from arrow_rs_amd import compute as K
ctx = A.Context(0)
n, m = 1_000_000_000, 100_000_000
R = A.array._RawMem
vals = ctx.alloc(n * 8); valid = ctx.alloc(((n + 63) // 64) * 8); ib = ctx.alloc(m * 4)
ctx.check(ctx.lib.ah_gen_uniform_i64(ctx.handle, vals.ptr, n, 42, -2**63, 2**63 - 1, 0))
ctx.check(ctx.lib.ah_gen_bernoulli_bits(ctx.handle, valid.ptr, n, 43, 0.9, 0))
ctx.check(ctx.lib.ah_gen_uniform_u32(ctx.handle, ib.ptr, m, 45, n, 0))
col = A.Array(ctx, A.Int64, n, R(vals.ptr, n * 8, vals), 0, R(valid.ptr, valid.nbytes, valid), 0, n // 10)
col_nonull = A.Array(ctx, A.Int64, n, R(vals.ptr, n * 8, vals), 0)
bits = A.Array(ctx, A.Boolean, n, R(valid.ptr, valid.nbytes, valid), 0)
idx = A.Array(ctx, A.UInt32, m, R(ib.ptr, m * 4, ib), 0)
def timeit(name, fn):
    ctx.profile(True); ctx.profile_reset()
    for _ in range(5): fn()
    ms, cnt = ctx.profile_get("take_gather")
    print(name, round(ms / cnt, 4), "ms")
timeit("values+validity", lambda: K.take(col, idx))
timeit("values only", lambda: K.take(col_nonull, idx))
timeit("bits only", lambda: K.take(bits, idx))
