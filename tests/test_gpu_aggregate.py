"""arrow_arith::aggregate on the device vs the oracle (arrow-arith/src/aggregate.rs).
Bit-exact for every integer / boolean / min / max result and for the checked error texts; float
sums and products within the summation error bound stated in each test (the reference's own
association order depends on its compile-time vector width, aggregate.rs:300-307)."""
import math

import zlib

import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd.compute import aggregate as G
import orc
from orc import HostArray, golden_array, load_golden
from test_oracle_golden import _same_scalar, ERR

pytestmark = pytest.mark.gpu

FN = {"sum": G.sum, "sum_checked": G.sum_checked, "product": G.product, "product_checked": G.product_checked,
      "min": G.min, "max": G.max, "bit_and": G.bit_and, "bit_or": G.bit_or, "bit_xor": G.bit_xor}
INTS = [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64]


@pytest.mark.parametrize("case", load_golden("aggregate"), ids=lambda c: c["name"])
@pytest.mark.parametrize("bit_offset", [0, 3])
def test_reference_goldens(ctx, case, bit_offset):
    v = golden_array(case["values"]).to_device(ctx, bit_offset)
    if "error" in case:
        with pytest.raises(ERR[case["error"]]) as e:
            FN[case["op"]](v)
        assert e.value.message == case["message"]
        return
    got = FN[case["op"]](v)
    assert _same_scalar(got, case["expected"], v.data_type), (got, case["expected"])


def _rand(rng, dt, n, small=False):
    npdt = np.dtype(dt.np_dtype)
    if npdt.kind == "f":
        x = rng.standard_normal(n).astype(npdt) * npdt.type(1000)
        return x
    info = np.iinfo(npdt)
    if small:
        return rng.integers(max(info.min, -3), min(info.max, 3), n, dtype=npdt, endpoint=True)
    return rng.integers(info.min, info.max, n, dtype=npdt, endpoint=True)


def _bits_equal(a, b):
    return (a is None and b is None) or (a is not None and b is not None and np.array([a]).tobytes() == np.array([b]).tobytes())


@pytest.mark.parametrize("dt", INTS, ids=repr)
def test_integer_aggregates_fuzz(ctx, oracle, dt):
    """Every order-independent aggregate, all lengths around the vector / workgroup boundaries, sliced
    (unaligned) value pointers and validity bit offsets."""
    rng = np.random.default_rng(zlib.crc32(dt.name.encode()))
    for n in (1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 1000, 4097, 65_537, 300_001):
        vals = _rand(rng, dt, n)
        for p_valid in (None, 0.9, 0.02):
            valid = None if p_valid is None else rng.random(n) < p_valid
            for off, ln in ((0, n), (min(1, n - 1), max(1, n - 2)), (n // 3, n - n // 3)):
                h = HostArray(dt, vals, valid).slice(off, ln)
                d = HostArray(dt, vals, valid).to_device(ctx, bit_offset=off % 7).slice(off, ln)
                for op in ("sum", "product", "min", "max", "bit_and", "bit_or", "bit_xor"):
                    want = oracle.aggregate(op, h)
                    got = FN[op](d)
                    assert _bits_equal(got, want), (dt, n, p_valid, off, op, got, want)


@pytest.mark.parametrize("dt", [A.Float32, A.Float64], ids=repr)
def test_float_min_max_total_order(ctx, oracle, dt):
    rng = np.random.default_rng(9)
    npdt = np.dtype(dt.np_dtype)
    u = np.uint32 if npdt.itemsize == 4 else np.uint64
    specials = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan, np.finfo(npdt).max, np.finfo(npdt).min,
                         np.finfo(npdt).tiny, -np.finfo(npdt).tiny], dtype=npdt)
    for n in (1, 7, 64, 1000, 100_003):
        for trial in range(6):
            vals = _rand(rng, dt, n)
            k = min(n, 1 + trial * 3)
            vals[rng.integers(0, n, k)] = specials[rng.integers(0, len(specials), k)]
            if trial == 5:  # NaNs with payloads and both signs: the order is on bits
                vals = rng.integers(0, np.iinfo(u).max, n, dtype=u).view(npdt)
            valid = None if trial % 2 == 0 else rng.random(n) < 0.7
            h = HostArray(dt, vals, valid)
            d = h.to_device(ctx, bit_offset=trial)
            for op in ("min", "max"):
                assert _bits_equal(FN[op](d), oracle.aggregate(op, h)), (dt, n, trial, op)


def test_float16_min_max(ctx):
    rng = np.random.default_rng(2)
    bits = rng.integers(0, 65535, 10_000, dtype=np.uint16)
    vals = bits.view(np.float16)
    valid = rng.random(len(vals)) < 0.8

    def key(b):  # total order on f16 bits
        b = int(b)
        return (b ^ 0x8000) if not (b & 0x8000) else (~b & 0xFFFF)
    live = bits[valid]
    d = A.Array.from_numpy(vals, valid, data_type=A.Float16, ctx=ctx)
    assert np.array([G.min(d)]).view(np.uint16)[0] == min(live, key=key)
    assert np.array([G.max(d)]).view(np.uint16)[0] == max(live, key=key)


@pytest.mark.parametrize("dt", [A.Float32, A.Float64], ids=repr)
def test_float_sum_product_within_bound(ctx, oracle, dt):
    """|device - exact| <= 2*log2(n)*eps*sum|x| (pairwise-summation bound with slack) and the oracle's
    three lane widths (16/32/64-byte reference builds) lie within n*eps*sum|x| of it; deterministic."""
    rng = np.random.default_rng(4)
    npdt = np.dtype(dt.np_dtype)
    eps = float(np.finfo(npdt).eps)
    for n in (1, 5, 100, 4096, 1_000_003):
        vals = _rand(rng, dt, n)
        for valid in (None, rng.random(n) < 0.9):
            h = HostArray(dt, vals, valid)
            d = h.to_device(ctx, bit_offset=2)
            live = vals if valid is None else vals[valid]
            exact = math.fsum(float(x) for x in live)
            sabs = math.fsum(abs(float(x)) for x in live)
            got = G.sum(d)
            assert float(got) == float(G.sum(d))
            assert abs(float(got) - exact) <= (2 * max(1.0, math.log2(n)) + 2) * eps * sabs + 1e-300
            for vb in (16, 32, 64):
                ref = float(oracle.aggregate("sum", h, vb))
                assert abs(float(got) - ref) <= n * eps * sabs + 1e-300
            assert float(G.sum_checked(d)) == float(got)  # float add_checked never fails
    # products: relative bound on a well-conditioned input
    vals = (1.0 + rng.standard_normal(10_000) * 1e-3).astype(npdt)
    h = HostArray(dt, vals)
    got = float(G.product(h.to_device(ctx)))
    ref = float(oracle.aggregate("product", h))
    assert abs(got - ref) <= 10_000 * eps * abs(ref)
    assert float(G.product_checked(h.to_device(ctx))) == got


def test_none_results(ctx):
    for dt in INTS + [A.Float32, A.Float64]:
        empty = A.Array.from_numpy(np.zeros(0, dtype=dt.np_dtype), ctx=ctx)
        alln = A.Array.from_numpy(np.ones(100, dtype=dt.np_dtype), np.zeros(100, bool), ctx=ctx)
        for f in FN.values():
            if dt.np_dtype in (np.float32, np.float64) and f in (G.bit_and, G.bit_or, G.bit_xor):
                with pytest.raises(A.array.InvalidArgumentError):
                    f(alln)
                continue
            assert f(empty) is None and f(alln) is None
    b = A.Array.from_numpy(np.ones(10, bool), np.zeros(10, bool), ctx=ctx)
    assert G.min_boolean(b) is None and G.bool_or(b) is None


def test_boolean_aggregates_fuzz(ctx, oracle):
    rng = np.random.default_rng(8)
    for n in (1, 63, 64, 65, 1000, 100_000):
        for pt in (0.0, 1.0, 0.5, 0.999):
            for pv in (None, 0.0005, 0.5):
                vals = rng.random(n) < pt
                valid = None if pv is None else rng.random(n) < pv
                h = HostArray(A.Boolean, vals, valid)
                for off, ln in ((0, n), (n // 2, n - n // 2)):
                    d = h.to_device(ctx, bit_offset=5).slice(off, ln)
                    hs = h.slice(off, ln)
                    for op, f in (("min", G.min_boolean), ("max", G.max_boolean)):
                        want = oracle.aggregate(op, hs)
                        got = f(d)
                        assert (got is None and want is None) or (got is not None and bool(got) == bool(want))


@pytest.mark.parametrize("dt", INTS, ids=repr)
def test_checked_sum_and_product_match_the_sequential_reference(ctx, oracle, dt):
    """The first overflowing PREFIX decides: same Ok value, or the same message (accumulator and element
    of the failing step), wherever in the array it happens."""
    rng = np.random.default_rng(31 + dt.physical)
    info = np.iinfo(dt.np_dtype)
    cases = []
    for n in (1, 2, 64, 1000, 70_001):
        cases.append((_rand(rng, dt, n, small=True), None))
        cases.append((_rand(rng, dt, n, small=True), rng.random(n) < 0.8))
        v = _rand(rng, dt, n)          # full range: overflows almost immediately
        cases.append((v, rng.random(n) < 0.5))
        z = _rand(rng, dt, n, small=True)  # an early zero protects the product
        z[rng.integers(0, n)] = 0
        cases.append((z, None))
    # hand-made edges: reach MAX exactly, MIN exactly, MIN * -1, overflow undone later (still an error)
    mx, mn = int(info.max), int(info.min)
    cases += [(np.array([mx, 0, 0], dt.np_dtype), None), (np.array([mx, 1], dt.np_dtype), None),
              (np.array([mx - 1, 1, 1], dt.np_dtype), None), (np.array([1] * 300 + [mx], dt.np_dtype), None),
              (np.array([2] * 70, dt.np_dtype), None), (np.array([mx, 2, 0], dt.np_dtype), None),
              (np.array([0, mx, 2], dt.np_dtype), None)]
    if mn < 0:
        half = -(mn // 2)  # 2^(w-2)
        cases += [(np.array([mn, -1], dt.np_dtype), None), (np.array([mn, 1, 1, -1], dt.np_dtype), None),
                  (np.array([mn, 0, -1], dt.np_dtype), None), (np.array([mn, 1, 0], dt.np_dtype), None),
                  (np.array([half, -2], dt.np_dtype), None), (np.array([half, -2, -1], dt.np_dtype), None),
                  (np.array([half, -2, 1, 1, 0, -1], dt.np_dtype), None), (np.array([half, 2], dt.np_dtype), None),
                  (np.array([-half, 2, 1], dt.np_dtype), None), (np.array([-half, -2], dt.np_dtype), None),
                  (np.array([mn, mn], dt.np_dtype), None), (np.array([-1] * 1001 + [mn], dt.np_dtype), None),
                  (np.array([-1] * 1000 + [mn], dt.np_dtype), None), (np.array([mx, 1, -5], dt.np_dtype), None)]
    # long runs where only one tile overflows late
    big = np.ones(200_000, dt.np_dtype)
    big[150_000] = info.max
    big[150_001] = 2
    cases.append((big, None))
    for vals, valid in cases:
        h = HostArray(dt, vals, valid)
        d = h.to_device(ctx, bit_offset=1)
        for op in ("sum_checked", "product_checked"):
            try:
                want = ("ok", oracle.aggregate(op, h))
            except A.array.ArithmeticOverflow as e:
                want = ("err", e.message)
            try:
                got = ("ok", FN[op](d))
            except A.array.ArithmeticOverflow as e:
                got = ("err", e.message)
            assert want[0] == got[0], (dt, op, vals[:8], want, got)
            if want[0] == "ok":
                assert _bits_equal(got[1], want[1]), (dt, op, vals[:8], want, got)
            else:
                assert got[1] == want[1], (dt, op, vals[:8])


def test_aggregate_after_filter_stays_on_device(ctx, oracle):
    """The typical consumer: lt(col, scalar) -> filter -> sum / min / max."""
    from arrow_rs_amd import compute as K
    n = 1_000_000
    vals = oracle.gen_i64(n, 42, -10**6, 10**6)
    valid = oracle.gen_bits(n, 43, 0.9)
    h = HostArray(A.Int64, vals, valid)
    d = h.to_device(ctx)
    thr = A.Scalar.new(1000, A.Int64, ctx)
    f = K.filter(d, K.lt(d, thr))
    hf = oracle.filter(h, oracle.compare(2, h, HostArray(A.Int64, np.array([1000])), r_scalar=True))
    for op in ("sum", "min", "max", "sum_checked"):
        assert _bits_equal(FN[op](f), oracle.aggregate(op, hf))
