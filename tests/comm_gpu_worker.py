"""RCCL world-1 exercise of the C-ABI communicator, run as a subprocess by tests/test_gpu_comm.py.
TEST INFRASTRUCTURE (uses the oracle)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import arrow_rs_amd as A  # noqa: E402
from arrow_rs_amd import compute as K  # noqa: E402
from arrow_rs_amd import distributed as D  # noqa: E402
import orc  # noqa: E402
from orc import HostArray, assert_logical_eq  # noqa: E402

ctx = A.Context(0)
A.set_default_context(ctx)
oracle = orc.load(os.path.join(ROOT, "oracle", "liboracle.so"))
comm = D.CApiCommunicator(ctx, 0, 1)  # world 1 through real RCCL (unique id, ncclCommInitRank)


def test_rccl_world1_all_gatherv_is_concat_of_one():
    rng = np.random.default_rng(5)
    n = 300_001
    for dt, vals in ((A.Int64, rng.integers(-2**62, 2**62, n)), (A.Float64, rng.standard_normal(n)),
                     (A.Int32, rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)), (A.UInt8, rng.integers(0, 255, n).astype(np.uint8))):
        for valid in (None, rng.random(n) < 0.9):
            h = HostArray(dt, np.asarray(vals, dtype=dt.np_dtype), valid)
            m = HostArray(A.Boolean, rng.random(n) < 0.1)
            f = K.filter(h.to_device(ctx), m.to_device(ctx))  # a real kernel result, as in the bench
            g = comm.all_gatherv(f)
            assert_logical_eq(HostArray.from_device(g), oracle.filter(h, m), f"all_gatherv {dt}")
            assert g.null_count() == f.null_count() and comm.last_exchange["peers"] == 0
    # zero-copy slice with a validity bit offset: the packed piece is re-aligned to bit 0
    h = HostArray(A.Int64, rng.integers(0, 100, 1000), rng.random(1000) < 0.5)
    d = h.to_device(ctx).slice(13, 900)
    assert_logical_eq(HostArray.from_device(comm.all_gatherv(d)), h.slice(13, 900), "sliced")
    e = HostArray(A.Int64, np.zeros(0, dtype=np.int64)).to_device(ctx)
    assert comm.all_gatherv(e).length == 0
    comm.barrier()
    assert comm.allreduce_max([1.5, -2.0, 7.25]) == [1.5, -2.0, 7.25]


def test_rccl_world1_record_batch():
    rng = np.random.default_rng(6)
    n = 100_000
    a = HostArray(A.Int64, rng.integers(-10, 10, n), rng.random(n) < 0.9)
    b = HostArray(A.Float64, rng.standard_normal(n))
    rb = A.RecordBatch(["a", "b"], [a.to_device(ctx), b.to_device(ctx)], n)
    out = comm.all_gather_record_batch(rb)
    assert out.num_rows() == n
    assert_logical_eq(HostArray.from_device(out.columns[0]), a, "col a")
    assert_logical_eq(HostArray.from_device(out.columns[1]), b, "col b")
    assert out.columns[1].validity is None
    # Boolean values and strings go through the same call (concat_boolean / concat_bytes)
    hb = HostArray(A.Boolean, rng.random(1000) < 0.5, rng.random(1000) < 0.9)
    assert_logical_eq(HostArray.from_device(comm.all_gatherv(hb.to_device(ctx).slice(3, 900))), hb.slice(3, 900), "bool")
    hs = HostArray(A.Utf8, [f"s{i % 97}" * (i % 5) for i in range(1000)], rng.random(1000) < 0.9)
    assert_logical_eq(HostArray.from_device(comm.all_gatherv(hs.to_device(ctx).slice(5, 800))), hs.slice(5, 800), "utf8")
    try:
        import ctypes as C
        v = a.to_device(ctx).view()
        v.type = A._lib.AH_UTF8_VIEW
        o_, s_ = A._lib.ArrayOut(), A._lib.ExchangeStats()
        ctx.check(ctx.lib.ah_all_gatherv(ctx.handle, comm._h, C.byref(v), C.byref(o_), C.byref(s_)))
        raise AssertionError("view columns must be refused")
    except A.ArrowError:
        pass


def test_rccl_world1_shared_cases():
    """the N-rank case list at world 1 over real RCCL (tests/comm_cases.py)"""
    import comm_cases as cc
    cc.run_rank(ctx, comm, oracle, 0, 1, heavy=False)


test_rccl_world1_all_gatherv_is_concat_of_one()
test_rccl_world1_record_batch()
test_rccl_world1_shared_cases()
del comm
print("COMM_WORKER_OK")
