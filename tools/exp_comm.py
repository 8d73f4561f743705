import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import numpy as np
import arrow_rs_amd as A
from arrow_rs_amd import compute as K, distributed as D
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
ctx = A.Context(0)
n = 100_000_000
buf = ctx.alloc(n * 8); valid = ctx.alloc(((n + 63) // 64) * 8)
ctx.check(ctx.lib.ah_gen_uniform_i64(ctx.handle, buf.ptr, n, 1, -5, 5, 0))
ctx.check(ctx.lib.ah_gen_bernoulli_bits(ctx.handle, valid.ptr, n, 2, 0.9, 0))
R = A.array._RawMem
arr = A.Array(ctx, A.Int64, n, R(buf.ptr, n * 8, buf), 0, R(valid.ptr, valid.nbytes, valid), 0, n // 10)
comm = D.Communicator(ctx, dist)
for it in range(4):
    ctx.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
    g = comm.all_gatherv(arr)
    ctx.synchronize(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("all_gatherv ms", (t1 - t0) * 1e3, getattr(comm, "timings", None))
dist.destroy_process_group()
