"""GPU parity suite (-m gpu): the HIP path, called through the C ABI, against
(1) the reference's golden vectors, (2) the CPU oracle on seeded random inputs incl. the
reference's fuzz recipes (filter.rs:1890-1977), sliced / offset / unaligned inputs, and
(3) size-independent properties at large sizes.  Integer/byte/index work is compared
bit-exactly; float arithmetic and casts are correctly rounded on both sides so the
tolerance is 0 ULP (stated bar: <= 1 ULP)."""
import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K
import orc
from orc import HostArray, golden_array, load_golden, assert_logical_eq, assert_same_nulls_presence
from test_oracle_golden import ARITH, CMP, ERR

pytestmark = pytest.mark.gpu

ARITH_FN = {0: K.add, 1: K.add_wrapping, 2: K.sub, 3: K.sub_wrapping, 4: K.mul, 5: K.mul_wrapping, 6: K.div, 7: K.rem}
CMP_FN = {0: K.eq, 1: K.neq, 2: K.lt, 3: K.lt_eq, 4: K.gt, 5: K.gt_eq, 6: K.distinct, 7: K.not_distinct}


def dev(spec_or_host, ctx):
    """golden spec / HostArray -> device Array; slices are REAL zero-copy device slices."""
    if isinstance(spec_or_host, HostArray):
        return spec_or_host.to_device(ctx)
    spec = spec_or_host
    full = dict(spec)
    sl = full.pop("slice", None)
    arr = golden_array(full).to_device(ctx)
    return arr.slice(*sl) if sl else arr


def host(arr):
    return HostArray.from_device(arr)


def expect_err(case, fn):
    if "panic" in case:
        with pytest.raises(A.Panic) as ei:
            fn()
        assert str(ei.value) == case["panic"]
    else:
        with pytest.raises(ERR[case["error"]]) as ei:
            fn()
        assert ei.value.message == case["message"]


def check(got_dev, exp_host, msg=""):
    got = host(got_dev)
    assert_logical_eq(got, exp_host, msg)
    assert got_dev.null_count() == exp_host.null_count, f"{msg} reported null_count"


def check_exact(got_dev, exp_host, msg=""):
    """logical equality + identical null-buffer presence + identical raw value bytes."""
    check(got_dev, exp_host, msg)
    got = host(got_dev)
    assert_same_nulls_presence(got, exp_host, msg)
    if not isinstance(exp_host.values, list) and len(exp_host):
        g, e = np.asarray(got.values), np.asarray(exp_host.values)
        assert g.tobytes() == e.tobytes(), f"{msg} raw value bytes differ (incl. null slots)"


# ------------------------------------------------------------------ golden
@pytest.mark.parametrize("case", [c for c in load_golden("filter") if "values" in c], ids=lambda c: c["name"])
def test_filter_golden(ctx, case):
    v, p = dev(case["values"], ctx), dev(case["predicate"], ctx)
    if "error" in case:
        return expect_err(case, lambda: K.filter(v, p))
    got = K.filter(v, p)
    assert got.data_type == v.data_type  # data_type preserved (filter.rs:783-787)
    check(got, golden_array(case["expected"]), case["name"])
    if "expected_null_count" in case:
        assert got.null_count() == case["expected_null_count"]


def test_filter_record_batch_no_columns(ctx):
    case = next(c for c in load_golden("filter") if c["name"] == "test_filter_record_batch_no_columns")
    rb = A.RecordBatch([], [], num_rows=case["record_batch_rows"])
    out = K.filter_record_batch(rb, dev(case["predicate"], ctx))
    assert out.num_rows() == case["expected_rows"] and out.num_columns() == 0


@pytest.mark.parametrize("case", load_golden("take"), ids=lambda c: c["name"])
def test_take_golden(ctx, case):
    v, i = dev(case["values"], ctx), dev(case["indices"], ctx)
    opts = K.TakeOptions(check_bounds=case.get("check_bounds", False))
    if "error" in case or "panic" in case:
        return expect_err(case, lambda: K.take(v, i, opts))
    got = K.take(v, i, opts)
    assert got.data_type == v.data_type
    check(got, golden_array(case["expected"]), case["name"])


@pytest.mark.parametrize("case", load_golden("arith"), ids=lambda c: c["name"])
def test_arith_golden(ctx, case, oracle):
    l, r = dev(case["lhs"], ctx), dev(case["rhs"], ctx)
    fn = ARITH_FN[ARITH[case["op"]]]
    if "error" in case:
        return expect_err(case, lambda: fn(l, r))
    exp = oracle.arith(ARITH[case["op"]], golden_array(case["lhs"]), golden_array(case["rhs"]))
    check_exact(fn(l, r), exp, case["name"])


@pytest.mark.parametrize("case", load_golden("cmp"), ids=lambda c: c["name"])
def test_cmp_golden(ctx, case):
    l = dev(case["lhs"], ctx)
    fn = CMP_FN[CMP[case["op"]]]
    if "rhs_scalar" in case:
        r = A.Scalar.new(case["rhs_scalar"]["value"], orc.TYPES[case["rhs_scalar"]["type"]], ctx)
    else:
        r = dev(case["rhs"], ctx)
    if "error" in case:
        return expect_err(case, lambda: fn(l, r))
    check(fn(l, r), golden_array(case["expected"]), case["name"])


@pytest.mark.parametrize("case", [c for c in load_golden("cast") if c["to"] not in ("Utf8", "LargeUtf8")],
                         ids=lambda c: c["name"])
def test_cast_golden(ctx, case):
    got = K.cast(dev(case["values"], ctx), orc.TYPES[case["to"]])
    check(got, golden_array(case["expected"]), case["name"])


# ------------------------------------------------------------- fuzz: filter
def _rand_values(rng, dt, n):
    if dt.physical == A._lib.AH_BOOL:
        return rng.random(n) < 0.5
    if np.dtype(dt.np_dtype).kind == "f":
        return (rng.normal(size=n) * 1e6).astype(dt.np_dtype)
    info = np.iinfo(dt.np_dtype)
    return rng.integers(info.min, info.max, n, dtype=dt.np_dtype, endpoint=True)


@pytest.mark.parametrize("dt", [A.Int8, A.Int16, A.Int32, A.Int64, A.Float64, A.Boolean], ids=str)
def test_fuzz_filter(ctx, oracle, dt):
    """fuzz_filter (filter.rs:1890-1977): random length, array offset, predicate offset and
    truncation, validity %, selectivity forced to 1.0 / 0.0 for the first iterations."""
    rng = np.random.default_rng(hash(dt.name) % 2**32)
    for it in range(60):
        n = int(rng.integers(32, 256)) if it < 40 else int(rng.integers(3000, 40000))
        sel = 1.0 if it < 5 else (0.0 if it <= 10 else float(rng.random()))
        vp = float(rng.random())
        full = HostArray(dt, _rand_values(rng, dt, n + 10), (rng.random(n + 10) < vp) if it % 2 == 0 else None)
        a_off = int(rng.integers(0, 10))
        pfull = HostArray(A.Boolean, rng.random(n + 20) < sel, (rng.random(n + 20) < 0.9) if it % 3 == 0 else None)
        p_off = int(rng.integers(0, 10))
        plen = n - int(rng.integers(0, 10))
        dv = full.to_device(ctx).slice(a_off, n)
        dp = pfull.to_device(ctx).slice(p_off, plen)
        exp = oracle.filter(full.slice(a_off, n), pfull.slice(p_off, plen))
        got = K.filter(dv, dp)
        check(got, exp, f"{dt} iter {it}")
        assert_same_nulls_presence(host(got), exp, f"{dt} iter {it}") if 0 < len(exp) < plen else None


def test_filter_wide_and_all_widths(ctx, oracle):
    rng = np.random.default_rng(5)
    n = 10000
    mask = HostArray(A.Boolean, rng.random(n) < 0.37)
    dm = mask.to_device(ctx)
    for dt in [A.UInt8, A.UInt16, A.UInt32, A.UInt64, A.Float32, A.Decimal128(38, 2)]:
        if dt.physical == A._lib.AH_FIXED16:
            raw = np.zeros(n, dtype=dt.np_dtype)
            raw["lo"] = rng.integers(0, 2**63, n, dtype=np.uint64)
            raw["hi"] = rng.integers(-2**62, 2**62, n, dtype=np.int64)
            h = HostArray(dt, raw, rng.random(n) < 0.8)
        else:
            h = HostArray(dt, _rand_values(rng, dt, n), rng.random(n) < 0.8)
        check(K.filter(h.to_device(ctx), dm), oracle.filter(h, mask), str(dt))


def test_filter_predicate_reuse_and_record_batch(ctx, oracle):
    rng = np.random.default_rng(9)
    n = 50000
    mask = HostArray(A.Boolean, rng.random(n) < 0.1, rng.random(n) < 0.97)
    cols = [HostArray(A.Int64, _rand_values(rng, A.Int64, n), rng.random(n) < 0.9),
            HostArray(A.Float64, _rand_values(rng, A.Float64, n), rng.random(n) < 0.9),
            HostArray(A.Int32, _rand_values(rng, A.Int32, n))]
    dm = mask.to_device(ctx)
    pred = K.FilterBuilder.new(dm).optimize().build()
    exp = [oracle.filter(c, mask) for c in cols]
    assert pred.count() == len(exp[0])
    for c, e in zip(cols, exp):
        check(pred.filter(c.to_device(ctx)), e)
    rb = A.RecordBatch(["a", "b", "c"], [c.to_device(ctx) for c in cols])
    out = K.filter_record_batch(rb, dm)
    assert out.num_rows() == len(exp[0])
    for c, e in zip(out.columns, exp):
        check(c, e)


# --------------------------------------------------------------- fuzz: take
@pytest.mark.parametrize("idt", [A.UInt8, A.Int8, A.UInt16, A.Int16, A.UInt32, A.Int32, A.UInt64, A.Int64], ids=str)
def test_fuzz_take(ctx, oracle, idt):
    rng = np.random.default_rng(hash(idt.name) % 2**32)
    for it in range(12):
        vlen = int(rng.integers(1, min(120, np.iinfo(idt.np_dtype).max)))
        n = int(rng.integers(1, 3000))
        vdt = [A.Int64, A.Int32, A.Int8, A.Float64, A.Boolean, A.UInt16][it % 6]
        v = HostArray(vdt, _rand_values(rng, vdt, vlen), (rng.random(vlen) < 0.8) if it % 2 else None)
        idx = rng.integers(0, vlen, n).astype(idt.np_dtype)
        ivalid = (rng.random(n) < 0.85) if it % 3 == 0 else None
        if ivalid is not None:  # garbage (out-of-bounds) under null indices must be masked
            idx = np.where(ivalid, idx, np.iinfo(idt.np_dtype).max).astype(idt.np_dtype)
        i = HostArray(idt, idx, ivalid)
        exp = oracle.take(v, i)
        got = K.take(v.to_device(ctx), i.to_device(ctx))
        check_exact(got, exp, f"{idt}->{vdt} iter {it}")


def test_take_oob_semantics(ctx):
    v = HostArray(A.Int32, np.arange(4, dtype=np.int32)).to_device(ctx)
    with pytest.raises(A.Panic) as ei:
        K.take(v, HostArray(A.Int32, np.array([1, -1], dtype=np.int32)).to_device(ctx))
    assert str(ei.value) == "index out of bounds: the len is 4 but the index is 4294967295"
    with pytest.raises(A.array.ComputeError) as ei:
        K.take(v, HostArray(A.Int32, np.array([1, -1], dtype=np.int32)).to_device(ctx), K.TakeOptions(True))
    assert ei.value.message == "Array index out of bounds, cannot get item at index -1 from 4 entries"
    i = HostArray(A.UInt32, np.array([1, 400, 2], dtype=np.uint32), np.array([True, True, False])).to_device(ctx)
    with pytest.raises(A.Panic) as ei:
        K.take(v, i)
    assert str(ei.value) == "Out-of-bounds index 400"


# -------------------------------------------------------------- fuzz: arith
@pytest.mark.parametrize("dt", [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64,
                                A.Float32, A.Float64], ids=str)
def test_fuzz_arith(ctx, oracle, dt):
    rng = np.random.default_rng(hash(dt.name) % 2**32)
    is_f = np.dtype(dt.np_dtype).kind == "f"
    for it in range(10):
        n = int(rng.integers(1, 5000))
        if is_f:
            a, b = _rand_values(rng, dt, n), _rand_values(rng, dt, n)
            b[rng.random(n) < 0.02] = 0
            a[rng.random(n) < 0.01] = np.inf
            a[rng.random(n) < 0.01] = np.nan
        else:  # small magnitudes so checked ops mostly succeed
            hi = min(np.iinfo(dt.np_dtype).max, 11)
            lo = max(np.iinfo(dt.np_dtype).min, -11)
            a = rng.integers(lo, hi, n).astype(dt.np_dtype)
            b = rng.integers(lo, hi, n).astype(dt.np_dtype)
        av = (rng.random(n) < 0.9) if it % 2 == 0 else None
        bv = (rng.random(n) < 0.9) if it % 3 == 0 else None
        ha, hb = HostArray(dt, a, av), HostArray(dt, b, bv)
        da, db = ha.to_device(ctx), hb.to_device(ctx)
        for op in range(8):
            if not is_f and op in (6, 7):
                hb2 = HostArray(dt, np.where(b == 0, 1, b).astype(dt.np_dtype), bv)
                exp = oracle.arith(op, ha, hb2)
                got = ARITH_FN[op](da, hb2.to_device(ctx))
            else:
                exp = oracle.arith(op, ha, hb)
                got = ARITH_FN[op](da, db)
            # floats: correctly rounded on both sides -> compare raw bytes (0 ULP <= 1 ULP bar);
            # NaN payloads from inf-inf / 0/0 are the hardware default quiet NaN on both sides
            if is_f:
                g, e = host(got), exp
                assert_same_nulls_presence(g, e)
                m = ~(np.isnan(g.values) & np.isnan(e.values))
                assert np.array_equal(g.values[m].view(np.uint8), e.values[m].view(np.uint8)), f"{dt} op {op}"
            else:
                check_exact(got, exp, f"{dt} op {op} iter {it}")


def test_arith_scalar_rules(ctx, oracle):
    a = HostArray.from_pylist([1, None, 3, 2**31 - 1], A.Int32)
    da = a.to_device(ctx)
    s = A.Scalar.new(10, A.Int32, ctx)
    hs = HostArray.from_pylist([10], A.Int32)
    check_exact(K.add_wrapping(da, s), oracle.arith(1, a, hs, r_scalar=True))
    check_exact(K.sub_wrapping(s, da), oracle.arith(3, hs, a, l_scalar=True))
    ns = A.Scalar.new(None, A.Int32, ctx)
    check_exact(K.add(da, ns), oracle.arith(0, a, HostArray.from_pylist([None], A.Int32), r_scalar=True))
    with pytest.raises(A.array.ArithmeticOverflow) as ei:
        K.add(da, s)
    assert ei.value.message == "Overflow happened on: 2147483647 + 10"
    with pytest.raises(A.array.ComputeError) as ei:
        K.add_wrapping(da, HostArray.from_pylist([1], A.Int32).to_device(ctx))
    assert ei.value.message == "Cannot perform binary operation on arrays of different length"
    with pytest.raises(A.array.ComputeError) as ei:
        K.add(da, HostArray.from_pylist([1], A.Int32).to_device(ctx))
    assert ei.value.message == "Cannot perform a binary operation on arrays of different length"
    # first failing row wins, null rows are skipped
    x = HostArray(A.Int32, np.array([2**31 - 1, 5, 2**31 - 1], dtype=np.int32), np.array([False, True, True]))
    with pytest.raises(A.array.ArithmeticOverflow) as ei:
        K.add(x.to_device(ctx), HostArray(A.Int32, np.array([1, 1, 7], dtype=np.int32)).to_device(ctx))
    assert ei.value.message == "Overflow happened on: 2147483647 + 7"
    check_exact(K.neg(da.slice(0, 3)), oracle.neg(a.slice(0, 3)))
    check_exact(K.neg_wrapping(da), oracle.neg(a, wrapping=True))


# ---------------------------------------------------------------- fuzz: cmp
@pytest.mark.parametrize("dt", [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt32, A.UInt64, A.Float32,
                                A.Float64, A.Boolean], ids=str)
def test_fuzz_cmp(ctx, oracle, dt):
    rng = np.random.default_rng(hash(dt.name) % 2**32 + 1)
    is_f = dt.np_dtype in (np.float32, np.float64)
    for it in range(8):
        n = int(rng.integers(1, 6000))
        if dt.physical == A._lib.AH_BOOL:
            a, b = rng.random(n) < 0.5, rng.random(n) < 0.5
        elif is_f:
            pool = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, -np.nan, 1e-300, 3.5], dtype=dt.np_dtype)
            a, b = rng.choice(pool, n), rng.choice(pool, n)
        else:
            a = rng.integers(-5 if np.iinfo(dt.np_dtype).min < 0 else 0, 6, n).astype(dt.np_dtype)
            b = rng.integers(-5 if np.iinfo(dt.np_dtype).min < 0 else 0, 6, n).astype(dt.np_dtype)
        av = (rng.random(n) < 0.8) if it % 2 == 0 else None
        bv = (rng.random(n) < 0.8) if it % 3 == 0 else None
        ha, hb = HostArray(dt, a, av), HostArray(dt, b, bv)
        da, db = ha.to_device(ctx), hb.to_device(ctx)
        off = int(rng.integers(0, min(n, 7)))
        for op in range(8):
            check(CMP_FN[op](da, db), oracle.compare(op, ha, hb), f"{dt} op {op}")
            assert_same_nulls_presence(host(CMP_FN[op](da, db)), oracle.compare(op, ha, hb), f"{dt} op {op}")
            # sliced (unaligned) inputs
            check(CMP_FN[op](da.slice(off, n - off), db.slice(off, n - off)),
                  oracle.compare(op, ha.slice(off, n - off), hb.slice(off, n - off)), f"{dt} op {op} sliced")
            # scalar on either side, null scalar
            sc = HostArray(dt, a[:1], None)
            dsc = A.Scalar(sc.to_device(ctx))
            check(CMP_FN[op](da, dsc), oracle.compare(op, ha, sc, r_scalar=True), f"{dt} op {op} rscalar")
            check(CMP_FN[op](dsc, db), oracle.compare(op, sc, hb, l_scalar=True), f"{dt} op {op} lscalar")
            nsc = HostArray(dt, a[:1], np.array([False]))
            check(CMP_FN[op](da, A.Scalar(nsc.to_device(ctx))), oracle.compare(op, ha, nsc, r_scalar=True),
                  f"{dt} op {op} null scalar")


# --------------------------------------------------------------- fuzz: cast
def test_fuzz_cast_numeric(ctx, oracle):
    rng = np.random.default_rng(21)
    types = [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64, A.Float32, A.Float64]
    for src in types:
        n = 3000
        if src.np_dtype in (np.float32, np.float64):
            vals = (rng.normal(size=n) * 10.0 ** rng.integers(0, 20, n)).astype(src.np_dtype)
            vals[:6] = [np.nan, np.inf, -np.inf, -0.9, 255.9, 256.0]
        else:
            vals = _rand_values(rng, src, n)
            vals[: n // 2] = (vals[: n // 2] % 100).astype(src.np_dtype)
        h = HostArray(src, vals, rng.random(n) < 0.9)
        d = h.to_device(ctx)
        for dst in types:
            exp = oracle.cast(h, dst)
            check_exact(K.cast(d, dst), exp, f"{src}->{dst}")
    h = HostArray(A.Float64, np.array([1.0, 256.0, 3.0]))
    with pytest.raises(A.array.CastError) as ei:
        K.cast_with_options(h.to_device(ctx), A.UInt8, K.CastOptions(safe=False))
    assert ei.value.message == "Can't cast value 256.0 to type UInt8"
    ok = HostArray(A.Int64, np.array([1, 2, 3], dtype=np.int64))
    got = K.cast_with_options(ok.to_device(ctx), A.Float64, K.CastOptions(safe=False))
    check_exact(got, oracle.cast(ok, A.Float64, safe=False))


def test_cast_i64_f64_rounding_edges(ctx, oracle):
    """>= 2^53 needs round-to-nearest-even (cast/mod.rs:2596-2603: `as f64`)."""
    base = np.array([2**53, 2**53 + 1, 2**53 + 2, 2**53 + 3, 2**62 + 1, 2**63 - 1, -(2**63), -(2**53) - 1,
                     2**63 - 513, 2**63 - 512, 2**63 - 511, 123456789012345678], dtype=np.int64)
    rng = np.random.default_rng(2)
    vals = np.concatenate([base, rng.integers(-2**63, 2**63 - 1, 20000, dtype=np.int64)])
    h = HostArray(A.Int64, vals)
    got = host(K.cast(h.to_device(ctx), A.Float64))
    assert np.array_equal(got.values.view(np.uint64), vals.astype(np.float64).view(np.uint64))
    got32 = host(K.cast(h.to_device(ctx), A.Float32))
    assert np.array_equal(got32.values.view(np.uint32), oracle.cast(h, A.Float32).values.view(np.uint32))


# ------------------------------------------------------------------- concat
def test_concat(ctx, oracle):
    rng = np.random.default_rng(31)
    for dt in [A.Int64, A.Int32, A.Boolean, A.Float64]:
        pieces = []
        for k in range(5):
            n = int(rng.integers(0, 700))
            pieces.append(HostArray(dt, _rand_values(rng, dt, n), (rng.random(n) < 0.8) if k % 2 == 0 else None))
        exp = oracle.concat(pieces)
        got = K.concat([p.to_device(ctx) for p in pieces])
        check_exact(got, exp, str(dt))
    nonull = [HostArray(A.Int32, np.arange(5, dtype=np.int32)), HostArray(A.Int32, np.arange(3, dtype=np.int32))]
    assert K.concat([p.to_device(ctx) for p in nonull]).nulls() is None


# ------------------------------------------------- generators + large sizes
def test_device_generators_match_host_twins(ctx, oracle):
    import ctypes as C
    n = 100003
    lib, h = ctx.lib, ctx.handle
    buf = ctx.alloc(n * 8)
    ctx.check(lib.ah_gen_uniform_i64(h, buf.ptr, n, 42, -10**6, 10**6, 17))
    assert np.array_equal(buf.to_numpy(np.int64, count=n), oracle.gen_i64(n, 42, -10**6, 10**6, row0=17))
    ctx.check(lib.ah_gen_uniform_i64(h, buf.ptr, n, 42, -2**63, 2**63 - 1, 0))
    assert np.array_equal(buf.to_numpy(np.int64, count=n), oracle.gen_i64(n, 42, -2**63, 2**63 - 1))
    ctx.check(lib.ah_gen_uniform_f64(h, buf.ptr, n, 7, -1e6, 1e6, 5))
    assert np.array_equal(buf.to_numpy(np.float64, count=n), oracle.gen_f64(n, 7, -1e6, 1e6, row0=5))
    ctx.check(lib.ah_gen_uniform_u32(h, buf.ptr, n, 8, 12345, 0))
    assert np.array_equal(buf.to_numpy(np.uint32, count=n), oracle.gen_u32(n, 8, 12345))
    ctx.check(lib.ah_gen_uniform_i32(h, buf.ptr, n, 9, 3))
    assert np.array_equal(buf.to_numpy(np.int32, count=n), oracle.gen_i32(n, 9, row0=3))
    ctx.check(lib.ah_gen_bernoulli_bits(h, buf.ptr, n, 10, 0.1, 64))
    got = A.unpack_bits(buf.to_numpy(np.uint8, count=((n + 63) // 64) * 8), 0, n)
    assert np.array_equal(got, oracle.gen_bits(n, 10, 0.1, row0=64))


def _gen_table(ctx, n, seed=42, sel=0.1, valid_p=0.9):
    """Config-2 shaped synthetic column, generated on device."""
    lib, h = ctx.lib, ctx.handle
    vals = ctx.alloc(n * 8)
    valid = ctx.alloc(((n + 63) // 64) * 8)
    mask = ctx.alloc(((n + 63) // 64) * 8)
    ctx.check(lib.ah_gen_uniform_i64(h, vals.ptr, n, seed, -2**63, 2**63 - 1, 0))
    ctx.check(lib.ah_gen_bernoulli_bits(h, valid.ptr, n, seed + 1, valid_p, 0))
    ctx.check(lib.ah_gen_bernoulli_bits(h, mask.ptr, n, seed + 2, sel, 0))
    ctx.check(lib.ah_zero_null_slots(h, vals.ptr, 8, valid.ptr, n))
    import ctypes as C
    cnt = C.c_int64()
    ctx.check(lib.ah_count_set_bits(h, valid.ptr, 0, n, C.byref(cnt)))
    R = A.array._RawMem
    col = A.Array(ctx, A.Int64, n, R(vals.ptr, n * 8, vals), 0, R(valid.ptr, valid.nbytes, valid), 0, n - cnt.value)
    pred = A.Array(ctx, A.Boolean, n, R(mask.ptr, mask.nbytes, mask), 0)
    return col, pred


def test_filter_medium_vs_oracle_window(ctx, oracle):
    """16M rows generated on device; the oracle re-generates the same rows on the host."""
    n = 1 << 24
    col, pred = _gen_table(ctx, n)
    got = K.filter(col, pred)
    vals = oracle.gen_i64(n, 42, -2**63, 2**63 - 1)
    valid = oracle.gen_bits(n, 43, 0.9)
    mask = oracle.gen_bits(n, 44, 0.1)
    vals[~valid] = 0
    exp = oracle.filter(HostArray(A.Int64, vals, valid), HostArray(A.Boolean, mask))
    check_exact(got, exp, "16M filter")


def test_filter_take_large_properties(ctx):
    """Size-independent properties at 2^28 rows (the full 1e9-row config runs in bench.py):
    filter == take(positions of the mask); complementary masks partition the column; counts and
    null counts add up; filtering twice is idempotent."""
    import ctypes as C
    n = 1 << 28
    col, pred = _gen_table(ctx, n)
    f = K.filter(col, pred)
    k = f.length
    assert abs(k / n - 0.1) < 1e-3
    # complement mask
    npred = K.eq(pred, A.Scalar.new(False, A.Boolean, ctx))
    g = K.filter(col, npred)
    assert f.length + g.length == n
    assert f.null_count() + g.null_count() == col.null_count()
    # positions of the mask = filter(iota, mask); take(col, positions) == filter(col, mask)
    iota = ctx.alloc(n * 8)
    # iota via cast-free trick: generate with range lo=hi is constant, so build with numpy in chunks
    step = 1 << 24
    for s in range(0, n, step):
        chunk = np.arange(s, s + step, dtype=np.int64)
        ctx.check(ctx.lib.ah_memcpy_htod(ctx.handle, iota.ptr + s * 8, chunk.ctypes.data, chunk.nbytes))
    R = A.array._RawMem
    iota_arr = A.Array(ctx, A.Int64, n, R(iota.ptr, n * 8, iota), 0)
    pos = K.filter(iota_arr, pred)
    t = K.take(col, pos)
    assert t.length == k and t.null_count() == f.null_count()
    eqv = K.not_distinct(t, f)
    cnt = C.c_int64()
    ctx.check(ctx.lib.ah_count_set_bits(ctx.handle, eqv.values.ptr, 0, k, C.byref(cnt)))
    assert cnt.value == k, "take(positions) != filter"
    # positions strictly increasing (order preserving)
    inc = K.lt(pos.slice(0, k - 1), pos.slice(1, k - 1))
    ctx.check(ctx.lib.ah_count_set_bits(ctx.handle, inc.values.ptr, 0, k - 1, C.byref(cnt)))
    assert cnt.value == k - 1
