"""arrow_ord::sort::sort_to_indices / sort / sort_limit / partition on the device vs the oracle
(arrow-ord/src/sort.rs, partition.rs).  Index results are compared exactly against the oracle's STABLE order
(the reference's `sort_unstable_by` leaves the order of equal keys open; stable is the order its own tests show
for every tie, and the order the reference goldens pin)."""
import zlib

import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K
import orc
from orc import HostArray, golden_array, load_golden

pytestmark = pytest.mark.gpu

OPTS = [(False, True), (False, False), (True, True), (True, False)]


@pytest.mark.parametrize("case", load_golden("sort"), ids=lambda c: c["name"])
@pytest.mark.parametrize("bit_offset", [0, 5])
def test_reference_goldens(ctx, case, bit_offset):
    v = golden_array(case["values"]).to_device(ctx, bit_offset)
    got = K.sort_to_indices(v, K.SortOptions(case.get("descending", False), case.get("nulls_first", True)), case.get("limit"))
    assert got.data_type == A.UInt32 and got.nulls() is None
    assert got.values_numpy().tolist() == case["expected"]


def _vals(rng, dt, n, narrow):
    npdt = np.dtype(dt.np_dtype)
    if dt == A.Boolean:
        return rng.random(n) < 0.5
    if npdt.kind == "f":
        x = (rng.standard_normal(n) * (3 if narrow else 1e6)).astype(npdt)
        if narrow:
            x = np.round(x)
        k = max(1, n // 50)
        sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan], dtype=npdt)
        x[rng.integers(0, n, k)] = sp[rng.integers(0, len(sp), k)]
        return x
    info = np.iinfo(npdt)
    if narrow:
        return rng.integers(max(info.min, -20), min(info.max, 20), n, dtype=npdt, endpoint=True)
    return rng.integers(info.min, info.max, n, dtype=npdt, endpoint=True)


@pytest.mark.parametrize("dt", [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64, A.Float16,
                                A.Float32, A.Float64, A.Boolean], ids=repr)
def test_sort_to_indices_fuzz(ctx, oracle, dt):
    """Sizes around tile boundaries (4096 pairs), many ties (narrow) and none (wide), nulls, all four
    SortOptions, limits, sliced inputs."""
    rng = np.random.default_rng(zlib.crc32(dt.name.encode()))
    for n in (1, 2, 63, 64, 4095, 4096, 4097, 10_000, 70_001):
        for narrow in (True, False):
            vals = _vals(rng, dt, n, narrow)
            for p_valid in (None, 0.9, 0.1):
                valid = None if p_valid is None else rng.random(n) < p_valid
                h = HostArray(dt, vals, valid)
                d = h.to_device(ctx, bit_offset=n % 7)
                for desc, nf in OPTS:
                    want = oracle.sort_to_indices(h, desc, nf)
                    got = K.sort_to_indices(d, K.SortOptions(desc, nf))
                    assert np.array_equal(got.values_numpy(), want.values), (dt, n, narrow, p_valid, desc, nf)
                lim = int(rng.integers(0, n + 3))
                want = oracle.sort_to_indices(h, True, False, lim)
                got = K.sort_to_indices(d, K.SortOptions(True, False), lim)
                assert np.array_equal(got.values_numpy() if got.length else np.zeros(0, np.uint32), want.values), (dt, n, "limit", lim)
    # sliced view (value pointer and validity bit offset advanced)
    vals = _vals(rng, dt, 9000, True)
    valid = rng.random(9000) < 0.8
    h = HostArray(dt, vals, valid)
    d = h.to_device(ctx, bit_offset=3).slice(1234, 5000)
    want = oracle.sort_to_indices(h.slice(1234, 5000), False, True)
    assert np.array_equal(K.sort_to_indices(d).values_numpy(), want.values)


def test_sort_and_sort_limit(ctx, oracle):
    rng = np.random.default_rng(3)
    n = 50_000
    h = HostArray(A.Float64, _vals(rng, A.Float64, n, False), rng.random(n) < 0.9)
    d = h.to_device(ctx)
    for desc, nf in OPTS:
        s = K.sort(d, K.SortOptions(desc, nf))
        want = oracle.take(h, oracle.sort_to_indices(h, desc, nf))
        orc.assert_logical_eq(HostArray.from_device(s), want, f"sort {desc} {nf}")
        top = K.sort_limit(d, K.SortOptions(desc, nf), 100)
        orc.assert_logical_eq(HostArray.from_device(top), want.slice(0, 100), f"sort_limit {desc} {nf}")
    assert K.sort_to_indices(A.Array.from_numpy(np.zeros(0, np.int32), ctx=ctx)).length == 0
    assert K.sort_to_indices(d, None, 0).length == 0
    assert K.sort_to_indices(A.Array.from_strings(["b", "a"], ctx=ctx)).values_numpy().tolist() == [1, 0]
    dec = A.Decimal128(10, 2)
    with pytest.raises(A.array.ComputeError, match="Sort not supported for data type"):
        K.sort_to_indices(HostArray(dec, np.zeros(3, dtype=dec.np_dtype)).to_device(ctx))
    assert str(K.SortOptions()) == "ASC NULLS FIRST" and str(K.SortOptions(True, False)) == "DESC NULLS LAST"


def test_large_sort_properties(ctx):
    """2^27 Int64 rows with 10 % nulls (bench generators): permutation, sortedness of the gathered values,
    null block = ascending null rows, stability on a low-cardinality key."""
    import bench
    n = 1 << 27
    col = bench.gen_i64_column(A, ctx, n, 42, 0.9, 0, -10**6, 10**6)
    idx = K.sort_to_indices(col, K.SortOptions(False, False))
    nulls = col.null_count()
    m = n - nulls
    assert idx.length == n
    sv = K.take(col, idx)
    assert sv.null_count() == nulls
    body = sv.slice(0, m)
    inc = K.lt_eq(body.slice(0, m - 1), body.slice(1, m - 1))
    from arrow_rs_amd.compute import aggregate as G
    assert G.min_boolean(inc) is True                      # ascending
    tail = idx.slice(m, nulls)
    assert G.min_boolean(K.lt(tail.slice(0, nulls - 1), tail.slice(1, nulls - 1))) is True  # null rows ascending
    assert G.min_boolean(K.is_null(K.take(col, tail))) is True
    # permutation: sorting the indices themselves gives iota, checked through the sum and strict increase
    again = K.take(idx, K.sort_to_indices(idx))
    assert G.min_boolean(K.lt(again.slice(0, n - 1), again.slice(1, n - 1))) is True
    assert int(G.min(again)) == 0 and int(G.max(again)) == n - 1
    # stability: among equal keys the row numbers increase
    eqk = K.eq(body.slice(0, m - 1), body.slice(1, m - 1))
    rows = idx.slice(0, m)
    asc = K.lt(rows.slice(0, m - 1), rows.slice(1, m - 1))
    assert G.min_boolean(K.or_(K.not_(eqk), asc)) is True


def test_partition(ctx):
    """partition.rs:126 and its tests (:238-330): ranges of equal consecutive rows, nulls equal to nulls."""
    a = A.Array.from_pylist([1, 1, 1, 2, 2, None, None, 9], A.Int64, ctx=ctx)
    assert K.partition([a]).ranges() == [(0, 3), (3, 5), (5, 7), (7, 8)]
    b = A.Array.from_pylist([1, 1, 2, 2, 2, 2, 3, 3], A.Int64, ctx=ctx)
    p = K.partition([a, b])
    assert p.ranges() == [(0, 2), (2, 3), (3, 5), (5, 6), (6, 7), (7, 8)] and len(p) == 6
    assert K.partition([A.Array.from_pylist([7], A.Int64, ctx=ctx)]).ranges() == [(0, 1)]
    assert K.partition([A.Array.from_numpy(np.zeros(0, np.int64), ctx=ctx)]).ranges() == []
    f = A.Array.from_numpy(np.array([np.nan, np.nan, 1.0, 1.0, -0.0, 0.0]), ctx=ctx)
    assert K.partition([f]).ranges() == [(0, 2), (2, 4), (4, 5), (5, 6)]   # bitwise distinct: -0.0 != 0.0, NaN == NaN
    with pytest.raises(A.array.InvalidArgumentError):
        K.partition([])


# ------------------------------------------------------------------ lexsort
def _lex_cols(ctx, hs, opts):
    return [K.SortColumn(h.to_device(ctx, bit_offset=i + 1), K.SortOptions(*o)) for i, (h, o) in enumerate(zip(hs, opts))]


def test_lexsort_reference_cases(ctx, oracle):
    """sort.rs:4109-4160 `test_lex_sort_mixed_types` (numeric part), :4061 single column, :4090 unaligned rows."""
    i64 = lambda xs: HostArray.from_pylist(xs, A.Int64)
    u32 = lambda xs: HostArray.from_pylist(xs, A.UInt32)
    hs = [i64([0, 2, -1, 0]), u32([101, 8, 7, 102]), i64([-1, -2, -3, -4])]
    cols = [K.SortColumn(h.to_device(ctx)) for h in hs]
    got = K.lexsort(cols)
    assert [c.to_pylist() for c in got] == [[-1, 0, 0, 2], [7, 101, 102, 8], [-3, -1, -4, -2]]
    assert [c.to_pylist() for c in K.lexsort(cols, 2)] == [[-1, 0], [7, 101], [-3, -1]]
    one = [K.SortColumn(i64([17, 2, -1, 0]).to_device(ctx))]
    assert K.lexsort(one)[0].to_pylist() == [-1, 0, 2, 17]
    assert K.lexsort(one, 3)[0].to_pylist() == [-1, 0, 2]
    with pytest.raises(A.array.ComputeError, match="lexical sort columns have different row counts"):
        K.lexsort([K.SortColumn(i64([None, -1]).to_device(ctx)), K.SortColumn(i64([5]).to_device(ctx))])
    with pytest.raises(A.array.InvalidArgumentError, match="Sort requires at least one column"):
        K.lexsort_to_indices([])
    # the null-ordering cases of :4190-4290 with the string column replaced by integers of the same order
    a = i64([None, -1, 2, None])
    b = i64([1, 3, 2, None])  # "foo" < "hello" < "world" -> 1 < 2 < 3
    dn = (True, True)
    got = K.lexsort(_lex_cols(ctx, [a, b], [dn, dn]))
    assert [c.to_pylist() for c in got] == [[None, None, 2, -1], [None, 1, 2, 3]]
    dl = (True, False)
    got = K.lexsort(_lex_cols(ctx, [a, b], [dl, dl]))
    assert [c.to_pylist() for c in got] == [[2, -1, None, None], [2, 3, 1, None]]
    a = i64([None, -1, 2, -1, None])
    b = i64([2, 1, 4, 3, None])  # "bar" < "foo" < "hello" < "world" -> 1 < 2 < 3 < 4
    got = K.lexsort(_lex_cols(ctx, [a, b], [(False, False), (True, True)]))
    assert [c.to_pylist() for c in got] == [[-1, -1, 2, None, None], [3, 1, 4, None, 2]]
    assert [c.to_pylist() for c in K.lexsort(_lex_cols(ctx, [a, b], [(False, False), (True, True)]), 10)] == \
        [[-1, -1, 2, None, None], [3, 1, 4, None, 2]]


def test_lexsort_fuzz(ctx, oracle):
    """2-4 columns of mixed types, few distinct values (so later columns matter), nulls, every option mix:
    exact index equality with the oracle's stable lexicographic sort."""
    rng = np.random.default_rng(17)
    types = [A.Int8, A.Int64, A.UInt16, A.Float64, A.Float32, A.Boolean, A.Int32, A.UInt64]
    for trial in range(40):
        n = int(rng.choice([1, 2, 100, 4097, 20_000]))
        k = int(rng.integers(2, 5))
        hs, opts = [], []
        for c in range(k):
            dt = types[int(rng.integers(0, len(types)))]
            vals = _vals(rng, dt, n, True)
            valid = (rng.random(n) < 0.8) if rng.random() < 0.6 else None
            hs.append(HostArray(dt, vals, valid))
            opts.append((bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
        lim = None if trial % 3 else int(rng.integers(0, n + 2))
        want = oracle.lexsort_to_indices([(h, o[0], o[1]) for h, o in zip(hs, opts)], lim)
        got = K.lexsort_to_indices(_lex_cols(ctx, hs, opts), lim)
        g = got.values_numpy() if got.length else np.zeros(0, np.uint32)
        assert np.array_equal(g, want.values), (trial, n, k, opts, lim)


# ------------------------------------------------------------------ Utf8 / LargeUtf8 (sort_bytes: bytewise, prefix first)
def _strings(rng, n, kind):
    if kind == "short":      # many ties, lengths 0..3
        alpha = ["a", "b", "", "ab", "ba", "a\x00", "\x00", "ß", "aa", "b\x00a"]
        return [str(alpha[i]) for i in rng.integers(0, len(alpha), n)]
    if kind == "prefix":     # long common prefixes: the order is decided in the 3rd / 4th 8-byte chunk
        base = "shared-prefix-of-24-bytes"
        return [base[: int(rng.integers(0, 25))] + "".join(rng.choice(list("xyz\x00"), int(rng.integers(0, 12)))) for _ in range(n)]
    return ["".join(rng.choice(list("abcdefghij😈é"), int(rng.integers(0, 20)))) for _ in range(n)]


@pytest.mark.parametrize("dt", [A.Utf8, A.LargeUtf8], ids=repr)
@pytest.mark.parametrize("kind", ["short", "prefix", "random"])
def test_sort_strings_fuzz(ctx, oracle, dt, kind):
    rng = np.random.default_rng(77)
    for n in (1, 2, 64, 4097, 20_000):
        rows = _strings(rng, n, kind)
        for p_valid in (None, 0.8):
            valid = None if p_valid is None else rng.random(n) < p_valid
            h = HostArray(dt, rows, valid)
            d = h.to_device(ctx)
            for desc, nf in OPTS:
                want = oracle.sort_to_indices(h, desc, nf)
                got = K.sort_to_indices(d, K.SortOptions(desc, nf))
                assert np.array_equal(got.values_numpy(), want.values), (dt, kind, n, p_valid, desc, nf)
            lim = int(rng.integers(0, n + 2))
            want = oracle.sort_to_indices(h, False, False, lim)
            got = K.sort_to_indices(d, K.SortOptions(False, False), lim)
            assert np.array_equal(got.values_numpy() if got.length else np.zeros(0, np.uint32), want.values), (kind, n, "limit", lim)
            # rank on the same column (bytes_rank, rank.rs:90-101)
            assert np.array_equal(K.rank(d), oracle.rank(h)), (kind, n, "rank")
            assert np.array_equal(K.rank(d, K.SortOptions(True, False)), oracle.rank(h, True, False)), (kind, n, "rank desc")
    h = HostArray(dt, _strings(rng, 3000, "prefix"), rng.random(3000) < 0.9)
    d = h.to_device(ctx).slice(700, 1500)
    assert np.array_equal(K.sort_to_indices(d).values_numpy(), oracle.sort_to_indices(h.slice(700, 1500)).values), "sliced"
    s = K.sort(h.to_device(ctx), K.SortOptions(True, True))  # sort = take(values, sort_to_indices)
    want = oracle.take(h, HostArray(A.UInt32, oracle.sort_to_indices(h, True, True).values))
    orc.assert_logical_eq(HostArray.from_device(s), want, "sort values")


def test_lexsort_with_string_columns(ctx, oracle):
    rng = np.random.default_rng(78)
    n = 6000
    names = HostArray(A.Utf8, [str(x) for x in rng.choice(["ann", "bob", "", "anna", "an", "bo"], n)], rng.random(n) < 0.9)
    ints = HostArray(A.Int32, rng.integers(0, 4, n).astype(np.int32), rng.random(n) < 0.9)
    tags = HostArray(A.LargeUtf8, _strings(rng, n, "short"))
    dn, di, dt_ = names.to_device(ctx), ints.to_device(ctx), tags.to_device(ctx)
    for opts in ([(False, True), (True, False), (False, False)], [(True, True), (False, True), (True, False)]):
        cols = [K.SortColumn(dn, K.SortOptions(*opts[0])), K.SortColumn(di, K.SortOptions(*opts[1])), K.SortColumn(dt_, K.SortOptions(*opts[2]))]
        want = oracle.lexsort_to_indices([(names, *opts[0]), (ints, *opts[1]), (tags, *opts[2])])
        got = K.lexsort_to_indices(cols)
        assert np.array_equal(got.values_numpy(), want.values), opts
