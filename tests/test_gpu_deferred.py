"""Deferred (opt-in asynchronous) mode of the C ABI — SURVEY.md §8(b) "opt-in async variant".

With `ah_context_set_deferred(ctx, 1)` the infallible fixed-shape entry points only enqueue their kernels:
no host synchronisation, `null_count = -1`.  These tests check, through the C ABI on the GPU, that
(1) every covered entry point produces the same logical result as its synchronous form and as the oracle,
(2) results chain into further deferred calls with no synchronisation in between (predicate built on the
    device -> filter -> arithmetic -> cast), bit-exact against the oracle,
(3) entry points that can fail on the device (checked arithmetic, unsafe casts, take with check_bounds) stay synchronous and
    still raise the reference's errors,
(4) the lazy `null_count()` / `ah_array_resolve` agree with a count of the validity bits."""
import ctypes as C

import numpy as np
import os

import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K
from arrow_rs_amd import _lib as L
from orc import HostArray, assert_logical_eq
from test_oracle_golden import BOOL_BIN, BOOL_UN

pytestmark = pytest.mark.gpu
# AH_DEBUG_GUARD=1 maps every pool block with the virtual-memory API, which a stream capture does not allow: the graph tests
# are not part of the guard-page run (tests/test_gpu_guard.py)
GUARD = pytest.mark.skipif(__import__("os").environ.get("AH_DEBUG_GUARD") == "1", reason="guard-page allocator cannot map memory during a capture")


def host(arr):
    return HostArray.from_device(arr)


def same(got_dev, exp_host, msg=""):
    assert got_dev._null_count == -1 or got_dev.validity is None or got_dev._null_count >= 0
    assert_logical_eq(host(got_dev), exp_host, msg)
    assert got_dev.null_count() == exp_host.null_count, f"{msg}: lazily counted nulls"


def _cols(rng, n, dt, p_valid=0.85):
    if dt.physical == L.AH_BOOL:
        v = rng.random(n) < 0.5
    elif np.dtype(dt.np_dtype).kind == "f":
        v = (rng.normal(size=n) * 1e3).astype(dt.np_dtype)
    else:
        info = np.iinfo(dt.np_dtype)
        v = rng.integers(info.min, info.max, n, dtype=dt.np_dtype, endpoint=True)
    valid = rng.random(n) < p_valid if p_valid is not None else None
    return HostArray(dt, v, valid)


@pytest.mark.parametrize("dt", [A.Int32, A.Int64, A.UInt8, A.Float64], ids=str)
def test_deferred_arith_matches_oracle(ctx, oracle, dt):
    rng = np.random.default_rng(11)
    for n, pv_a, pv_b in [(1, 0.9, None), (777, 0.9, 0.9), (70_001, None, 0.8), (4096, None, None)]:
        ha, hb = _cols(rng, n, dt, pv_a), _cols(rng, n, dt, pv_b)
        da, db = ha.to_device(ctx), hb.to_device(ctx)
        with ctx.deferred_mode():
            outs = [(op, fn(da, db)) for op, fn in ((1, K.add_wrapping), (3, K.sub_wrapping), (5, K.mul_wrapping))]
            neg = K.neg_wrapping(da)
            sc = A.Scalar.new(3, dt, ctx)
            rs = K.add_wrapping(da, sc)
            for _, o in outs:
                assert o._null_count == (-1 if o.validity is not None else 0)
        for op, o in outs:
            exp = oracle.arith(op, ha, hb)
            if np.dtype(dt.np_dtype).kind == "f":
                same(o, exp, f"{dt} op {op} n {n}")  # finite normals only: 0 ULP, NaN sign caveat not in play
            else:
                same(o, exp, f"{dt} op {op} n {n}")
        same(neg, oracle.neg(ha, wrapping=True), "neg_wrapping")
        same(rs, oracle.arith(1, ha, HostArray.from_pylist([3], dt), r_scalar=True), "scalar rhs")


def test_deferred_bitwise(ctx, oracle):
    rng = np.random.default_rng(5)
    ha, hb = _cols(rng, 5000, A.UInt32, 0.9), _cols(rng, 5000, A.UInt32, None)
    da, db = ha.to_device(ctx), hb.to_device(ctx)
    with ctx.deferred_mode():
        got = {name: getattr(K, name)(da, db) for name in ("bitwise_and", "bitwise_or", "bitwise_xor")}
        gnot = K.bitwise_not(da)
    for name, o in got.items():
        same(o, oracle.bitwise({"bitwise_and": 8, "bitwise_or": 9, "bitwise_xor": 10}[name], ha, hb), name)
    same(gnot, oracle.bitwise(14, ha), "bitwise_not")


@pytest.mark.parametrize("dt", [A.Int64, A.Float64, A.Boolean, A.UInt16], ids=str)
def test_deferred_compare_matches_oracle(ctx, oracle, dt):
    rng = np.random.default_rng(12)
    fns = {0: K.eq, 1: K.neq, 2: K.lt, 3: K.lt_eq, 4: K.gt, 5: K.gt_eq, 6: K.distinct, 7: K.not_distinct}
    for n, pa, pb in [(1, 0.5, 0.5), (63, 0.8, None), (64, None, None), (10_007, 0.8, 0.8)]:
        ha, hb = _cols(rng, n, dt, pa), _cols(rng, n, dt, pb)
        if dt is not A.Boolean and np.dtype(dt.np_dtype).kind != "f":
            hb = HostArray(dt, np.where(rng.random(n) < 0.3, ha.values, hb.values).astype(dt.np_dtype), hb.valid)
        da, db = ha.to_device(ctx), hb.to_device(ctx)
        sc = HostArray(dt, ha.values[:1], None)
        nsc = HostArray(dt, ha.values[:1], np.array([False]))
        with ctx.deferred_mode():
            got = {op: fn(da, db) for op, fn in fns.items()}
            got_s = {op: fn(da, A.Scalar(sc.to_device(ctx))) for op, fn in fns.items()}
            got_n = {op: fn(A.Scalar(nsc.to_device(ctx)), db) for op, fn in fns.items()}
        for op in fns:
            same(got[op], oracle.compare(op, ha, hb), f"{dt} op {op} n {n}")
            same(got_s[op], oracle.compare(op, ha, sc, r_scalar=True), f"{dt} op {op} scalar")
            same(got_n[op], oracle.compare(op, nsc, hb, l_scalar=True), f"{dt} op {op} null scalar")


def test_deferred_boolean_and_nullif(ctx, oracle):
    rng = np.random.default_rng(13)
    n = 9001
    ha, hb = _cols(rng, n, A.Boolean, 0.8), _cols(rng, n, A.Boolean, 0.7)
    hv = _cols(rng, n, A.Int64, 0.9)
    da, db, dv = ha.to_device(ctx), hb.to_device(ctx), hv.to_device(ctx)
    with ctx.deferred_mode():
        got = {name: getattr(K, name)(da, db) for name in ("and_", "or_", "and_kleene", "or_kleene", "and_not")}
        gnot, gnull, gnn = K.not_(da), K.is_null(dv), K.is_not_null(dv)
        gnif = K.nullif(dv, da)
    for name, o in got.items():
        same(o, oracle.boolean_binary(BOOL_BIN[name.rstrip("_")], ha, hb), name)
    same(gnot, oracle.boolean_unary(BOOL_UN["not"], ha), "not")
    same(gnull, oracle.boolean_unary(BOOL_UN["is_null"], hv), "is_null")
    same(gnn, oracle.boolean_unary(BOOL_UN["is_not_null"], hv), "is_not_null")
    same(gnif, oracle.nullif(hv, ha), "nullif")


def test_deferred_cast_numeric(ctx, oracle):
    rng = np.random.default_rng(14)
    n = 33_333
    hi = _cols(rng, n, A.Int64, 0.9)
    hf = HostArray(A.Float64, (rng.normal(size=n) * 1e10), rng.random(n) < 0.9)
    di, df = hi.to_device(ctx), hf.to_device(ctx)
    with ctx.deferred_mode():
        a = K.cast(di, A.Float64)
        b = K.cast(df, A.Int32)       # safe: out-of-range -> null, so the null count is data dependent
        c = K.cast(di, A.Int64)       # same type: a clone
        assert a._null_count == -1 and b._null_count == -1
    same(a, oracle.cast(hi, A.Float64), "i64->f64")
    same(b, oracle.cast(hf, A.Int32), "f64->i32 safe")
    same(c, hi, "clone")
    # unsafe casts report the failing value: synchronous, even in deferred mode
    with ctx.deferred_mode():
        with pytest.raises(A.array.CastError):
            K.cast_with_options(df, A.Int8, K.CastOptions(safe=False))


@pytest.mark.parametrize("dt", [A.Int64, A.Int16, A.Boolean, A.Float32], ids=str)
def test_deferred_filter(ctx, oracle, dt):
    rng = np.random.default_rng(15)
    for n, sel in [(5, 0.5), (4097, 0.1), (100_003, 0.9), (3000, 1.0), (3000, 0.0)]:
        hv = _cols(rng, n, dt, 0.85)
        hm = HostArray(A.Boolean, rng.random(n) < sel, rng.random(n) < 0.95 if n % 2 else None)
        dv, dm = hv.to_device(ctx), hm.to_device(ctx)
        with ctx.deferred_mode():
            got = K.filter(dv, dm)
        same(got, oracle.filter(hv, hm), f"{dt} n {n} sel {sel}")


def test_deferred_pipeline_no_sync_between_calls(ctx, oracle):
    """lt(col, scalar) -> and(is_not_null(col)) -> filter both columns -> add_wrapping -> cast, all enqueued
    back to back; the only host synchronisations are the predicate's row count and the final read."""
    rng = np.random.default_rng(16)
    n = 300_000
    ha = HostArray(A.Int64, rng.integers(-1000, 1000, n), rng.random(n) < 0.9)
    hb = HostArray(A.Int64, rng.integers(-10**12, 10**12, n), rng.random(n) < 0.9)
    da, db = ha.to_device(ctx), hb.to_device(ctx)
    sc, hsc = A.Scalar.new(100, A.Int64, ctx), HostArray.from_pylist([100], A.Int64)
    with ctx.deferred_mode():
        pred = K.and_(K.lt(da, sc), K.is_not_null(db))
        assert pred._null_count == -1
        fa, fb = K.filter(da, pred), K.filter(db, pred)
        total = K.add_wrapping(fa, fb)
        out = K.cast(total, A.Float64)
        assert out._null_count == -1
    epred = oracle.boolean_binary(BOOL_BIN["and"], oracle.compare(2, ha, hsc, r_scalar=True),
                                  oracle.boolean_unary(BOOL_UN["is_not_null"], hb))
    efa, efb = oracle.filter(ha, epred), oracle.filter(hb, epred)
    same(out, oracle.cast(oracle.arith(1, efa, efb), A.Float64), "pipeline")
    same(fa, efa, "pipeline column a")


def test_fallible_ops_stay_synchronous(ctx):
    a = HostArray(A.Int32, np.array([2**31 - 1, 1], dtype=np.int32)).to_device(ctx)
    b = HostArray(A.Int32, np.array([1, 1], dtype=np.int32)).to_device(ctx)
    with ctx.deferred_mode():
        with pytest.raises(A.array.ArithmeticOverflow) as ei:
            K.add(a, b)
        assert ei.value.message == "Overflow happened on: 2147483647 + 1"
        ok = K.sub(a, b)  # checked and fine: complete at return, count known
        assert ok._null_count == 0
        # take with check_bounds reports at return in every mode (without it, round 5: the gather is only enqueued and the
        # reference's panic surfaces at the next synchronisation — test_take_deferred_mode_parity_and_late_oob)
        with pytest.raises(A.array.ComputeError):
            K.take(a, HostArray(A.UInt32, np.array([0, 7], dtype=np.uint32)).to_device(ctx), K.TakeOptions(True))
        t = K.take(a, HostArray(A.UInt32, np.array([1, 0], dtype=np.uint32)).to_device(ctx))
    assert host(t).values.tolist() == [1, 2**31 - 1]
    assert host(ok).values.tolist() == [2**31 - 2, 0]


def test_array_resolve_c_abi(ctx):
    """ah_array_resolve: synchronise + count the nulls of a deferred ah_array_out in place."""
    rng = np.random.default_rng(17)
    n = 50_000
    ha, hb = _cols(rng, n, A.Int64, 0.9), _cols(rng, n, A.Int64, 0.9)
    da, db = ha.to_device(ctx), hb.to_device(ctx)
    lib, h = ctx.lib, ctx.handle
    out = L.ArrayOut()
    va, vb = da.view(), db.view()
    lib.ah_context_set_deferred(h, 1)
    try:
        assert lib.ah_context_deferred(h) == 1
        ctx.check(lib.ah_arith_binary(h, 1, C.byref(va), 0, C.byref(vb), 0, C.byref(out)))
        assert out.null_count == -1 and out.validity
        ctx.check(lib.ah_array_resolve(h, C.byref(out)))
    finally:
        lib.ah_context_set_deferred(h, 0)
    assert lib.ah_context_deferred(h) == 0
    expect_nulls = int(n - (ha.valid & hb.valid).sum())
    assert out.null_count == expect_nulls
    ctx.check(lib.ah_array_resolve(h, C.byref(out)))  # idempotent
    assert out.null_count == expect_nulls
    lib.ah_array_release(h, C.byref(out))


def test_deferred_small_batches_are_cheaper(ctx):
    """The point of the mode: a chain of kernels over engine-sized batches (8 192 rows) is bound by the
    per-call host synchronisation.  Not a strict performance assertion, only 'not slower'."""
    import time
    rng = np.random.default_rng(18)
    n = 8192
    da = _cols(rng, n, A.Int64, 0.9).to_device(ctx)
    db = _cols(rng, n, A.Int64, 0.9).to_device(ctx)

    def chain():
        return K.cast(K.add_wrapping(K.mul_wrapping(da, db), da), A.Float64)

    def run(iters):
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            r = chain()
        ctx.synchronize()
        return (time.perf_counter() - t0) / iters, r

    run(50)
    t_sync, r_sync = run(300)
    with ctx.deferred_mode():
        run(50)
        t_def, r_def = run(300)
    assert_logical_eq(host(r_def), host(r_sync), "deferred chain result")
    print(f"8192-row chain of 3 kernels: synchronous {t_sync * 1e6:.1f} us, deferred {t_def * 1e6:.1f} us")
    assert t_def < t_sync * 1.1


@pytest.mark.skipif(os.environ.get("AH_DEBUG_REDZONE") == "1", reason="the redzone check synchronizes: not capturable")
@GUARD
def test_deferred_chain_captured_in_a_hip_graph(ctx, oracle):
    """Deferred calls make no host synchronisation and (with a warm pool) no hipMalloc, so a chain of them can be
    stream-captured into a hipGraph and replayed on new input bytes: the launch-bound small-batch loop as ONE
    graph launch.  torch is only the capture/replay plumbing (torch.cuda.CUDAGraph = hipGraph on ROCm)."""
    import torch
    rng = np.random.default_rng(19)
    n = 8192
    ha, hb = _cols(rng, n, A.Int64, 0.9), _cols(rng, n, A.Int64, 0.9)
    gctx = A.Context(0)
    stream = torch.cuda.Stream()
    gctx.lib.ah_context_set_stream(gctx.handle, stream.cuda_stream)
    da, db = ha.to_device(gctx), hb.to_device(gctx)

    def chain():
        return K.cast(K.add_wrapping(K.mul_wrapping(da, db), da), A.Float64)

    gctx.set_deferred(True)
    try:
        warm = [chain() for _ in range(2)]  # fill the pool with every block size the chain needs
        gctx.synchronize()
        del warm
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
            out = chain()
        # new input bytes behind the same device pointers, then replay
        ha2 = HostArray(A.Int64, rng.integers(-2**40, 2**40, n), ha.valid)
        gctx.check(gctx.lib.ah_memcpy_htod(gctx.handle, da.values.ptr, np.ascontiguousarray(ha2.values).ctypes.data, n * 8))
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
    finally:
        gctx.set_deferred(False)
        gctx.lib.ah_context_set_stream(gctx.handle, None)
    exp = oracle.cast(oracle.arith(1, oracle.arith(5, ha2, hb), ha2), A.Float64)
    same(out, exp, "graph replay")

    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        graph.replay()
    torch.cuda.synchronize()
    print(f"hipGraph replay of the 3-kernel chain: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per chain")


@GUARD
def test_graph_capture_through_the_c_abi(ctx, oracle):
    """ah_graph_begin / _end / _launch (round 4): the launch-bound small-batch loop as ONE graph launch WITHOUT torch —
    deferred calls are recorded on the context's stream (arithmetic chain, compare, a prebuilt FilterPredicate applied to
    a nullable column), the inputs' bytes are then replaced behind the same device pointers and the graph is replayed:
    every captured output must equal the oracle on the NEW bytes.  Scratch released while recording stays parked.  (What
    happens when a call that must WAIT is recorded: test_graph_capture_refuses_calls_that_wait, in a child process.)"""
    rng = np.random.default_rng(23)
    n = 65536
    gctx = A.Context(0)
    ha, hb = _cols(rng, n, A.Int64, 0.9), _cols(rng, n, A.Int64, None)
    hf = _cols(rng, n, A.Float32, None)
    hm = HostArray(A.Boolean, rng.random(n) < 0.5)
    da, db, df, dm = ha.to_device(gctx), hb.to_device(gctx), hf.to_device(gctx), hm.to_device(gctx)
    pred = K.FilterBuilder(dm).build()  # built OUTSIDE the capture: its count is a host wait
    with gctx.graph_capture() as g:
        chain = K.cast(K.add_wrapping(K.mul_wrapping(da, db), da), A.Float64)
        cmpd = K.lt(df, df)
        summed = K.add_wrapping(df, df)
        filt = pred.filter(da)
        nd = K.distinct(da, db)  # (allocates and releases a scratch bitmap while recording)
    assert g.node_count() >= 6
    # new bytes behind the same pointers (validity of `a` stays: shapes are frozen at capture)
    ha2 = HostArray(A.Int64, rng.integers(-2**40, 2**40, n), ha.valid)
    hb2 = HostArray(A.Int64, rng.integers(-2**20, 2**20, n))
    hf2 = _cols(rng, n, A.Float32, None)
    for d, h in ((da, ha2), (db, hb2), (df, hf2)):
        gctx.check(gctx.lib.ah_memcpy_htod(gctx.handle, d.values.ptr, np.ascontiguousarray(h.values).ctypes.data, n * h.values.dtype.itemsize))
    for _ in range(3):
        g.launch()
    gctx.synchronize()
    same(chain, oracle.cast(oracle.arith(1, oracle.arith(5, ha2, hb2), ha2), A.Float64), "graph: arithmetic chain")
    same(cmpd, oracle.compare(2, hf2, hf2), "graph: lt")
    same(summed, oracle.arith(1, hf2, hf2), "graph: add f32")
    same(filt, oracle.filter(ha2, hm), "graph: prebuilt predicate on a nullable column")
    same(nd, oracle.compare(6, ha2, hb2), "graph: distinct")
    # other work on the context between replays does not disturb the parked scratch / outputs
    other = K.add_wrapping(db, db)
    g.launch()
    gctx.synchronize()
    same(other, oracle.arith(1, hb2, hb2), "call between replays")
    same(chain, oracle.cast(oracle.arith(1, oracle.arith(5, ha2, hb2), ha2), A.Float64), "graph: chain after other work")
    import time
    gctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.launch()
    gctx.synchronize()
    print(f"hipGraph replay of 7 recorded calls on 65 536 rows: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per replay")


@GUARD
def test_graph_capture_refuses_calls_that_wait():
    """An entry point that has to wait on the device (checked arithmetic reads its error word back; take; strings; a
    host copy) cannot be recorded: it must FAIL FAST — never spin on a mailbox whose posting kernel is only being
    recorded — and the context must work afterwards, also when the failure invalidated the capture (ROCm can leave the
    stream refusing every later launch: ah_graph_end then replaces the context's own stream).  Runs in a child process
    under a hard timeout: the failure mode being guarded against is a hang."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "graph_fail_worker.py")], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "GRAPH_FAIL_WORKER_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
