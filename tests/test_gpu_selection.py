"""parquet RowSelection on the device (arrow_rs_amd.selection) vs the reference's own unit tests and the oracle
model (tests/selection_model.py), bit-exact."""
import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K
from arrow_rs_amd.selection import RowSelection, RowSelector
import selection_cases
from selection_model import DeviceAdapter, ModelSelection, bits_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", selection_cases.ALL_CASES, ids=lambda f: f.__name__)
def test_reference_cases_on_the_device(ctx, case):
    case(DeviceAdapter(ctx))


def _dev(ctx, bits, off=0):
    return RowSelection(A.Array.from_numpy(np.asarray(bits, dtype=bool), ctx=ctx, bit_offset=off))


def test_and_then_fuzz_vs_model(ctx, oracle):
    """Sizes around word / wave / workgroup-tile boundaries, all densities incl. the two fast paths."""
    ModelSelection.oracle = oracle
    rng = np.random.default_rng(5)
    for n in (1, 63, 64, 65, 4095, 4096, 4097, 262_143, 262_144, 262_145, 1_000_003):
        for pa, pb in ((0.2, 0.8), (0.5, 0.5), (0.999, 0.01), (0.01, 0.999), (1.0, 0.3), (0.3, 1.0), (0.3, 0.0), (0.0, 0.5)):
            a = rng.random(n) < pa
            b = rng.random(int(a.sum())) < pb
            want = ModelSelection(a).and_then(ModelSelection(b))
            got = _dev(ctx, a, n % 7).and_then(_dev(ctx, b, n % 3))
            assert np.array_equal(bits_of(got), want.bits), (n, pa, pb)
            assert got.row_count() == int(b.sum())


def test_combine_fuzz_vs_model(ctx, oracle):
    ModelSelection.oracle = oracle
    rng = np.random.default_rng(6)
    for n, k in ((100, 100), (100, 37), (37, 100), (64, 128), (128, 64), (1, 1000), (70_001, 65_536), (262_145, 262_144), (5, 0)):
        a, b = rng.random(n) < 0.4, rng.random(k) < 0.6
        for op in ("intersection", "union"):
            want = getattr(ModelSelection(a), op)(ModelSelection(b))
            got = getattr(_dev(ctx, a, 5), op)(_dev(ctx, b, 2))
            assert np.array_equal(bits_of(got), want.bits), (n, k, op)


def test_runs_round_trip_fuzz(ctx, oracle):
    """mask -> boundaries -> selectors == mask_to_selectors of the oracle; selectors -> mask is the inverse
    (boolean.rs:727 `test_boolean_mask_from_selectors_fuzz_equivalence`)."""
    ModelSelection.oracle = oracle
    rng = np.random.default_rng(7)
    for n in (1, 64, 65, 1000, 262_145, 600_001):
        for p, run in ((0.5, 1), (0.1, 1), (0.5, 40), (0.02, 3000), (1.0, 1), (0.0, 1)):
            bits = np.repeat(rng.random(n // run + 1) < p, run)[:n]
            d = _dev(ctx, bits, 3)
            sels = d.selectors()
            assert sels == ModelSelection(bits).selectors(), (n, p, run)
            assert sum(s.row_count for s in sels) == n
            assert all(x.skip != y.skip for x, y in zip(sels, sels[1:])) and all(s.row_count > 0 for s in sels)
            back = RowSelection.from_selectors(sels, ctx)
            assert np.array_equal(bits_of(back), bits)


def test_offset_limit_trim_split_fuzz(ctx, oracle):
    """boolean.rs:576-636 `test_mask_backing_fuzz_equivalence`: every transform agrees with the model."""
    ModelSelection.oracle = oracle
    rng = np.random.default_rng(8)
    for n in (10, 64, 1000, 262_200, 700_000):
        bits = rng.random(n) < 0.3
        bits[-int(rng.integers(0, min(n, 70))):] = False
        cnt = int(bits.sum())
        for k in sorted({0, 1, 2, cnt // 2, cnt - 1, cnt, cnt + 5} - {-1}):
            d, m = _dev(ctx, bits, 1), ModelSelection(bits)
            assert np.array_equal(bits_of(d.offset(k)), m.offset(k).bits), (n, "offset", k)
            assert np.array_equal(bits_of(d.limit(k)), m.limit(k).bits), (n, "limit", k)
            assert d._find_nth(k) == m._find_nth(k)
            assert d._find_nth(k, n // 3) == m._find_nth(k, n // 3)
        assert np.array_equal(bits_of(_dev(ctx, bits).trim()), ModelSelection(bits).trim().bits)
        for cut in (0, 1, n // 2, n - 1, n, n + 10):
            d, m = _dev(ctx, bits, 6), ModelSelection(bits)
            hd, hm = d.split_off(cut), m.split_off(cut)
            assert np.array_equal(bits_of(hd), hm.bits) and np.array_equal(bits_of(d), m.bits)


def test_null_masks_panic_like_the_reference(ctx):
    m = A.Array.from_numpy(np.ones(10, bool), np.arange(10) % 2 == 0, ctx=ctx)
    with pytest.raises(A.Panic, match="left: 5"):
        RowSelection.from_filters([m])
    with pytest.raises(A.Panic, match="left: 5"):
        RowSelection(m)


def test_row_filter_loop_stays_on_device(ctx, oracle):
    """The caller this row exists for (arrow_reader/read_plan.rs `with_predicate`): predicate per batch ->
    from_filters -> and_then onto the running selection -> second predicate evaluated only on selected rows."""
    ModelSelection.oracle = oracle
    n, batch = 1 << 20, 1 << 17
    from orc import HostArray
    a = oracle.gen_i64(n, 1, -1000, 1000)
    b = oracle.gen_i64(n, 2, -1000, 1000)
    da, db = HostArray(A.Int64, a).to_device(ctx), HostArray(A.Int64, b).to_device(ctx)
    zero = A.Scalar.new(0, A.Int64, ctx)
    filters = [K.lt(da.slice(i, batch), zero) for i in range(0, n, batch)]
    sel1 = RowSelection.from_filters(filters, ctx)
    b_sel = K.filter(db, sel1.as_mask())                    # rows the first predicate kept
    sel2 = sel1.and_then(RowSelection.from_filters([K.gt(b_sel, A.Scalar.new(500, A.Int64, ctx))]))
    want = (a < 0) & (b > 500)
    assert sel2.row_count() == int(want.sum()) and sel2.total_row_count() == n
    assert np.array_equal(sel2.as_mask().values_numpy(), want)
    got = K.filter(da, sel2.as_mask())
    assert np.array_equal(got.values_numpy(), a[want])
    assert sel2.selectors() == ModelSelection(want).selectors()


def test_billion_row_selection_properties(ctx):
    """BASELINE-size masks (2^30 bits): size-independent properties of the algebra."""
    n = 1 << 30
    buf = ctx.alloc(n // 8)
    ctx.check(ctx.lib.ah_gen_bernoulli_bits(ctx.handle, buf.ptr, n, 11, 0.1, 0))
    m = A.Array(ctx, A.Boolean, n, A.array._RawMem(buf.ptr, n // 8, buf))
    s = RowSelection(m)
    k = s.row_count()
    assert abs(k / n - 0.1) < 1e-3
    ob = ctx.alloc((k + 63) // 64 * 8)
    ctx.check(ctx.lib.ah_gen_bernoulli_bits(ctx.handle, ob.ptr, k, 12, 0.5, 0))
    other = RowSelection(A.Array(ctx, A.Boolean, k, A.array._RawMem(ob.ptr, ob.nbytes, ob)))
    r = s.and_then(other)
    assert r.total_row_count() == n and r.row_count() == other.row_count()
    assert r.intersection(s) == r and r.union(s).row_count() == k      # r is a subset of s
    # filter(mask = s) of r's bits gives back `other` (and_then is the inverse of compaction)
    back = K.filter(r.as_mask(), s.as_mask())
    assert back.length == k
    eq = K.eq(back, other.as_mask())
    from arrow_rs_amd.compute import aggregate as G
    assert G.min_boolean(eq) is True
    lim = s.limit(1000)
    assert lim.row_count() == 1000 and lim.total_row_count() == s._find_nth(1000)
    assert s.offset(k - 5).row_count() == 5
