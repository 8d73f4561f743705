"""Worker for tests/test_gpu_parity.py::test_communicator_two_ranks_one_gpu: rank r of a 2-rank job,
BOTH ranks on GPU 0 (the box has one GPU), transport = gloo with device tensors.  Everything except
the transport is the production path: device filter per shard, Communicator.all_gatherv with direct
placement of value pieces and the funnel-shift bitmap merge at a non-aligned bit offset."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import arrow_rs_amd as A  # noqa: E402
from arrow_rs_amd import compute as K, distributed as D  # noqa: E402
import orc  # noqa: E402
from orc import HostArray  # noqa: E402

rank, world, rdzv = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dist.init_process_group("gloo", init_method=f"file://{rdzv}", rank=rank, world_size=world)
torch.cuda.set_device(0)
oracle = orc.load(os.path.join(ROOT, "oracle", "liboracle.so"))
ctx = A.Context(0)
comm = D.Communicator(ctx, dist)
n = 1_000_003
for case, (p_valid, sel, dt) in enumerate([(0.9, 0.1, A.Int64), (1.0, 0.37, A.Float64), (0.5, 0.9, A.Int32)]):
    vals = oracle.gen_i64(n, 42 + case, -2**40, 2**40)
    valid = oracle.gen_bits(n, 43 + case, p_valid) if p_valid < 1 else None
    mask = oracle.gen_bits(n, 44 + case, sel)
    vals = vals.astype(dt.np_dtype)
    s, e = D.shard_range(n, rank, world)
    shard = HostArray(dt, vals[s:e], None if valid is None else valid[s:e])
    f = K.filter(shard.to_device(ctx), HostArray(A.Boolean, mask[s:e]).to_device(ctx))
    g = comm.all_gatherv(f)
    exp = oracle.filter(HostArray(dt, vals, valid), HostArray(A.Boolean, mask))
    got = HostArray.from_device(g)
    orc.assert_logical_eq(got, exp, f"rank {rank} case {case}")
    assert g.null_count() == exp.null_count and (g.nulls() is None) == (exp.valid is None)
# multi-column form: one IPC-framed message per rank, decoded as zero-copy views (chunked result)
vals = oracle.gen_i64(n, 77, -10**9, 10**9)
valid = oracle.gen_bits(n, 78, 0.9)
mask = oracle.gen_bits(n, 79, 0.2)
strs = [None if i % 13 == 0 else f"r{i}" * (i % 4) for i in range(20_000)]
s, e = D.shard_range(n, rank, world)
ss, se = D.shard_range(len(strs), rank, world)
cols = [HostArray(A.Int64, vals[s:e], valid[s:e]).to_device(ctx, 3), HostArray(A.Float64, vals[s:e].astype(np.float64)).to_device(ctx)]
rb = K.filter_record_batch(A.RecordBatch(["a", "b"], cols), HostArray(A.Boolean, mask[s:e]).to_device(ctx))
parts = comm.all_gather_batches(rb)
assert len(parts) == world
exp_a = oracle.filter(HostArray(A.Int64, vals, valid), HostArray(A.Boolean, mask))
exp_b = oracle.filter(HostArray(A.Float64, vals.astype(np.float64)), HostArray(A.Boolean, mask))
orc.assert_logical_eq(HostArray.from_device(K.concat([p.columns[0] for p in parts])), exp_a, f"rank {rank} ipc a")
orc.assert_logical_eq(HostArray.from_device(K.concat([p.columns[1] for p in parts])), exp_b, f"rank {rank} ipc b")
sb = A.RecordBatch(["s"], [A.Array.from_strings([x or "" for x in strs[ss:se]], [x is not None for x in strs[ss:se]], ctx=ctx)])
got = [x for p in comm.all_gather_batches(sb) for x in p.columns[0].to_pylist()]
assert got == strs, f"rank {rank} ipc strings"
dist.barrier()
dist.destroy_process_group()
print("RANK_OK", rank)
