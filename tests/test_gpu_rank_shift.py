"""rank (arrow_ord::rank, arrow-ord/src/rank.rs) and shift (arrow_select::window, arrow-select/src/window.rs) on the
device, through the C ABI, against the reference's literals and the CPU oracle."""
import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K
from orc import HostArray, assert_logical_eq, golden_array, load_golden
from test_rank_shift_cpu import golden_values

pytestmark = pytest.mark.gpu


def host(a):
    return HostArray.from_device(a)


@pytest.mark.parametrize("case", load_golden("rank_shift"), ids=lambda c: c["name"])
def test_rank_shift_golden(ctx, case):
    v = golden_values(case).to_device(ctx)
    if case["op"] == "rank":
        got = K.rank(v, K.SortOptions(case["descending"], case["nulls_first"]))
        assert got.dtype == np.uint32 and got.tolist() == case["expected"]
    else:
        assert_logical_eq(host(K.shift(v, case["offset"])), golden_array(case["expected"]), case["name"])


@pytest.mark.parametrize("dt", [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt32, A.UInt64, A.Float32, A.Float64, A.Boolean], ids=str)
def test_rank_fuzz_vs_oracle(ctx, oracle, dt):
    rng = np.random.default_rng(31)
    for it, n in enumerate((1, 2, 63, 64, 65, 1000, 40_000)):
        if dt.physical == A._lib.AH_BOOL:
            vals = rng.random(n) < 0.4
        elif np.dtype(dt.np_dtype).kind == "f":
            vals = rng.integers(-20, 20, n).astype(dt.np_dtype) / 4
            vals[rng.random(n) < 0.05] = np.nan
            vals[rng.random(n) < 0.05] = -0.0
        elif it % 2:
            info = np.iinfo(dt.np_dtype)
            vals = rng.integers(info.min, info.max, n, dtype=dt.np_dtype, endpoint=True)  # mostly distinct
        else:
            vals = rng.integers(0, 7, n).astype(dt.np_dtype)  # long runs of ties
        valid = None if it % 3 == 0 else rng.random(n) < 0.8
        hv = HostArray(dt, vals, valid)
        dv = hv.to_device(ctx)
        for desc in (False, True):
            for nf in (True, False):
                got = K.rank(dv, K.SortOptions(desc, nf))
                assert np.array_equal(got, oracle.rank(hv, desc, nf)), (dt, n, desc, nf)
        off = n // 3
        assert np.array_equal(K.rank(dv.slice(off, n - off)), oracle.rank(hv.slice(off, n - off))), "sliced"


def test_rank_edges(ctx, oracle):
    assert K.rank(A.Array.from_numpy(np.array([], dtype=np.int32), None, A.Int32, ctx)).tolist() == []
    allnull = HostArray(A.Int64, np.arange(5, dtype=np.int64), np.zeros(5, dtype=bool))
    assert K.rank(allnull.to_device(ctx)).tolist() == oracle.rank(allnull).tolist() == [5] * 5
    assert K.rank(allnull.to_device(ctx), K.SortOptions(False, False)).tolist() == [5] * 5
    same = HostArray(A.Float64, np.full(10_000, 2.5))
    assert K.rank(same.to_device(ctx)).tolist() == [10_000] * 10_000  # one run: every row shares the top rank
    assert K.rank(A.Array.from_strings(["foo", "fo", "bar", "bar"], None, A.Utf8, ctx)).tolist() == [4, 3, 2, 2]  # test_bytes
    r = K.rank_array(HostArray(A.Int32, np.array([3, 1, 2], dtype=np.int32)).to_device(ctx))
    assert r.data_type == A.UInt32 and r.validity is None and r.to_pylist() == [3, 1, 2]


def test_rank_at_scale(ctx):
    """2^26 rows: ranks are a function of the value (equal values, equal ranks), lie in [nulls + 1, n], the largest
    value's rank is n, and sum(rank) over distinct-valued input equals the triangular number."""
    import bench
    n = 1 << 26
    col = bench.gen_i64_column(A, ctx, n, 42, 0.9, 0)  # full-range Int64: distinct with overwhelming probability
    r = K.rank_array(col)
    nulls = col.null_count()
    m = n - nulls
    assert r.length == n and r.validity is None
    assert K.aggregate.min(r) == nulls and K.aggregate.max(r) == n  # null rows share rank `nulls`
    total = K.aggregate.sum(K.cast(r, A.UInt64))
    assert total == nulls * nulls + (nulls + 1 + n) * m // 2
    # the permutation property: taking the column in rank order is sorted
    order = K.sort_to_indices(col)
    rv = K.take(r, order)
    head = rv.slice(nulls, min(m, 1 << 20)).values_numpy().astype(np.int64)
    assert np.array_equal(head, np.arange(nulls + 1, nulls + 1 + len(head)))


@pytest.mark.parametrize("dt", [A.Int8, A.Int32, A.Int64, A.Float64, A.Boolean, A.Utf8, A.LargeUtf8], ids=str)
def test_shift_fuzz_vs_oracle(ctx, oracle, dt):
    rng = np.random.default_rng(32)
    for it, n in enumerate((1, 2, 64, 65, 1000, 5000)):
        if dt.physical in (A._lib.AH_UTF8, A._lib.AH_LARGE_UTF8):
            vals = ["s" * int(rng.integers(0, 9)) + str(i) for i in range(n)]
        elif dt.physical == A._lib.AH_BOOL:
            vals = rng.random(n) < 0.5
        else:
            vals = rng.integers(-100, 100, n).astype(dt.np_dtype)
        valid = None if it % 2 else rng.random(n) < 0.8
        hv = HostArray(dt, vals, valid)
        dv = hv.to_device(ctx)
        for off in (0, 1, -1, 7, -7, 63, -64, n - 1, -(n - 1), n, -n, n + 5, -2**63, 2**63 - 1):
            got = K.shift(dv, off)
            exp = oracle.shift(hv, off)
            assert_logical_eq(host(got), exp, f"{dt} n={n} off={off}")
            if off != 0:
                assert got.validity is not None and got.null_count() == exp.null_count
        sl = dv.slice(n // 4, n - n // 4)
        assert_logical_eq(host(K.shift(sl, 3)), oracle.shift(hv.slice(n // 4, n - n // 4), 3), "sliced input")


def test_shift_zero_shares_buffers_and_empty(ctx):
    dv = HostArray.from_pylist([1, None, 4], A.Int32).to_device(ctx)
    same = K.shift(dv, 0)
    assert same.values.ptr == dv.values.ptr and same.validity.ptr == dv.validity.ptr  # make_array(array.to_data())
    e = K.shift(A.Array.from_numpy(np.array([], dtype=np.int64), None, A.Int64, ctx), 3)
    assert e.length == 0 and e.to_pylist() == []
