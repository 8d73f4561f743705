"""Temporal casts on the device (-m gpu): ``ah_cast_with_types`` through the Python mirror's ``cast`` /
``cast_with_options`` against the reference's test vectors (tests/golden/cast_temporal.json) and against the oracle on
every (from, to, safe) pair — values, validity, null-buffer presence, error kind and text — with validity bitmaps at
non-zero bit offsets, sliced inputs, sizes that straddle the kernel's 1024-row block, and a property check at 2^26 rows.
Integer work: bit-exact."""
import itertools

import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K
from orc import HostArray, golden_array, load_golden, lookup_type, assert_logical_eq, assert_same_nulls_presence
from test_oracle_golden import ERR
from test_temporal_cast_cpu import all_types, _edge_values, I32, I64, MULT, UNITS

pytestmark = pytest.mark.gpu


def host(arr):
    return HostArray.from_device(arr)


def check_exact(got_dev, exp, msg=""):
    got = host(got_dev)
    assert_logical_eq(got, exp, msg)
    assert_same_nulls_presence(got, exp, msg)
    assert got_dev.null_count() == exp.null_count, f"{msg} reported null_count"


@pytest.mark.parametrize("case", load_golden("cast_temporal"), ids=lambda c: c["name"])
def test_cast_temporal_golden(ctx, case):
    v = golden_array(case["values"]).to_device(ctx)
    to = lookup_type(case["to"])
    opts = K.CastOptions(safe=case["safe"])
    if "error" in case:
        with pytest.raises(ERR[case["error"]]) as ei:
            K.cast_with_options(v, to, opts)
        assert ei.value.message == case["message"]
        return
    got = K.cast_with_options(v, to, opts)
    assert got.data_type == to
    assert_logical_eq(host(got), golden_array(case["expected"]), case["name"])


def _values_for(rng, f, n):
    if f.np_dtype == np.int32:
        rnd = rng.integers(I32.min, I32.max, n, dtype=np.int64)
    else:
        rnd = np.concatenate([rng.integers(I64.min, I64.max, n // 2, dtype=np.int64),
                              rng.integers(-4 * 10**12, 4 * 10**12, n - n // 2, dtype=np.int64)])
    return np.array(_edge_values(f) + [int(x) for x in rnd], dtype=f.np_dtype)


def test_every_pair_matches_the_oracle(ctx, oracle):
    rng = np.random.default_rng(17)
    types = all_types(zones=(None, "+05:45"))
    n_ok = n_err = 0
    for f, t in itertools.product(types, types):
        if f.logical is None and t.logical is None:
            continue
        vals = _values_for(rng, f, 2200)  # > 2 blocks of 1024 rows, ragged tail
        benign = np.array([int(x) for x in rng.integers(-10**5, 10**5, 1500)], dtype=f.np_dtype)
        for with_nulls, bit_offset in ((False, 0), (True, 3)):
            for data in (vals, benign):
                valid = (rng.random(len(data)) < 0.85) if with_nulls else None
                h = HostArray(f, data, valid)
                d = h.to_device(ctx, bit_offset=bit_offset)
                for safe in (True, False):
                    tag = f"{f} -> {t} safe={safe} nulls={with_nulls} off={bit_offset} n={len(data)}"
                    try:
                        exp = oracle.cast_with_types(h, t, safe=safe)
                    except A.array.ArrowError as e:
                        with pytest.raises(type(e)) as ei:
                            K.cast_with_options(d, t, K.CastOptions(safe=safe))
                        assert ei.value.message == e.message, tag
                        n_err += 1
                        continue
                    check_exact(K.cast_with_options(d, t, K.CastOptions(safe=safe)), exp, tag)
                    n_ok += 1
    assert n_ok > 1500 and n_err > 300, (n_ok, n_err)


def test_sliced_inputs_and_empty(ctx, oracle):
    rng = np.random.default_rng(4)
    vals = rng.integers(-10**12, 10**12, 5000, dtype=np.int64)
    h = HostArray(A.Timestamp(A.MILLISECOND, "+05:45"), vals, rng.random(5000) < 0.7)
    d = h.to_device(ctx)
    for off, ln in ((0, 5000), (1, 4999), (63, 1000), (64, 64), (1027, 2049), (4999, 1), (17, 0)):
        hs = h.slice(off, ln)
        for to in (A.Date32, A.Time32Millisecond, A.Timestamp(A.NANOSECOND), A.Timestamp(A.SECOND, "-08:00"), A.Int32):
            got, exp = K.cast(d.slice(off, ln), to), oracle.cast_with_types(hs, to)
            if ln == 0:  # an empty result owns no buffers on the device (the reference's is Some(empty NullBuffer))
                assert len(got) == 0 and got.data_type == to
                continue
            check_exact(got, exp, f"slice({off},{ln}) -> {to}")
    e = HostArray(A.Date32, np.empty(0, dtype=np.int32))
    for to in (A.Date64, A.Timestamp(A.MICROSECOND), A.Timestamp(A.SECOND, "+01:00")):
        got = K.cast(e.to_device(ctx), to)
        assert len(got) == 0 and got.data_type == to


def test_first_failure_is_reported(ctx, oracle):
    """try_unary stops at the FIRST failing valid row; failing rows under nulls are skipped."""
    n = 50_000
    vals = np.arange(n, dtype=np.int64)
    vals[[40_001, 12_345, 30_000]] = [I64.max, I64.max - 7, I64.min]
    valid = np.ones(n, dtype=bool)
    h = HostArray(A.TimestampSecond, vals, valid)
    with pytest.raises(A.array.ArithmeticOverflow) as ei:
        K.cast_with_options(h.to_device(ctx), A.TimestampMillisecond, K.CastOptions(safe=False))
    assert ei.value.message == f"Overflow happened on: {I64.max - 7} * 1000"
    valid[12_345] = False
    h = HostArray(A.TimestampSecond, vals, valid)
    with pytest.raises(A.array.CastError) as ei:
        K.cast(h.to_device(ctx), A.Date32)  # safe mode, still an error (try_unary in both modes)
    assert ei.value.message == f"Cannot convert arrow_array::types::TimestampSecondType {I64.min} to datetime"
    got = K.cast(h.to_device(ctx), A.TimestampNanosecond)  # safe: overflow -> null
    check_exact(got, oracle.cast_with_types(h, A.TimestampNanosecond))
    assert got.null_count() == 3  # the null row and the two overflowing ones


def test_cast_with_types_rejects_a_mismatched_layout(ctx):
    import ctypes as C
    from arrow_rs_amd import _lib as L
    d = HostArray(A.Int64, np.arange(4, dtype=np.int64)).to_device(ctx)
    v, f, t, out = d.view(), A.Date32.descriptor(), A.Date64.descriptor(), L.ArrayOut()  # Date32 is an i32 layout
    st = ctx.lib.ah_cast_with_types(ctx.handle, C.byref(v), C.byref(f), C.byref(t), 1, C.byref(out))
    assert st == L.AH_INVALID_ARGUMENT
    with pytest.raises(A.array.CastError) as ei:
        K.cast(HostArray(A.Date32, np.arange(4, dtype=np.int32)).to_device(ctx), A.Time32Second)
    assert ei.value.message == "Casting from Date32 to Time32(s) not supported"


def test_deferred_temporal_casts(ctx, oracle):
    rng = np.random.default_rng(8)
    h = HostArray(A.Date32, rng.integers(-10**5, 10**5, 10_000).astype(np.int32), rng.random(10_000) < 0.9)
    d = h.to_device(ctx)
    with ctx.deferred_mode():
        a = K.cast(d, A.Timestamp(A.MICROSECOND))            # checked multiply, safe: infallible shape
        b = K.cast(a, A.Timestamp(A.SECOND))                 # truncating divide
        c = K.cast(b, A.Timestamp(A.SECOND, "+05:45"))       # zone adjust, safe
        e = K.cast(c, A.Date64)
    x = oracle.cast_with_types(h, A.Timestamp(A.MICROSECOND))
    y = oracle.cast_with_types(x, A.Timestamp(A.SECOND))
    z = oracle.cast_with_types(y, A.Timestamp(A.SECOND, "+05:45"))
    w = oracle.cast_with_types(z, A.Date64)
    check_exact(e, w, "deferred chain")
    check_exact(c, z, "deferred chain (zone)")


def test_large_round_trips(ctx):
    """2^26 rows: Date32 -> Timestamp(ms) -> Date32 is the identity, Timestamp(us) -> (Date32, Time64(us)) splits and
    recombines exactly, and unit up-then-down scaling returns the input."""
    n = 1 << 26
    rng = np.random.default_rng(99)
    days = rng.integers(-700_000, 700_000, n).astype(np.int32)
    d = HostArray(A.Date32, days).to_device(ctx)
    ts = K.cast(d, A.TimestampMillisecond)
    back = K.cast(ts, A.Date32)
    assert np.array_equal(back.values_numpy(), days)
    us = rng.integers(-(10**17), 10**17, n, dtype=np.int64)
    t = HostArray(A.TimestampMicrosecond, us).to_device(ctx)
    date = K.cast(t, A.Date32).values_numpy().astype(np.int64)
    tod = K.cast(t, A.Time64Microsecond).values_numpy()
    assert tod.min() >= 0 and tod.max() < 86_400_000_000
    assert np.array_equal(date * 86_400_000_000 + tod, us)
    s = HostArray(A.DurationSecond, rng.integers(-(10**9), 10**9, n, dtype=np.int64)).to_device(ctx)
    up = K.cast(s, A.DurationNanosecond)
    assert up.null_count() == 0
    assert np.array_equal(K.cast(up, A.DurationSecond).values_numpy(), s.values_numpy())


# ------------------------------------------------------- temporal arithmetic
ARITH_FN = {0: K.add, 1: K.add_wrapping, 2: K.sub, 3: K.sub_wrapping, 4: K.mul, 5: K.mul_wrapping, 6: K.div, 7: K.rem}


@pytest.mark.parametrize("case", load_golden("arith_temporal"), ids=lambda c: c["name"])
def test_arith_temporal_golden(ctx, case):
    l, r = golden_array(case["lhs"]).to_device(ctx), golden_array(case["rhs"]).to_device(ctx)
    if "error" in case:
        with pytest.raises(ERR[case["error"]]) as ei:
            ARITH_FN[case["op"]](l, r)
        assert ei.value.message == case["message"]
        assert str(ei.value) == case.get("display", str(ei.value))
        return
    for op in ((case["op"], case["op"] + 1) if case["op"] in (0, 2, 4) else (case["op"],)):  # *_wrapping is checked too
        got = ARITH_FN[op](l, r)
        exp = golden_array(case["expected"])
        assert got.data_type == exp.data_type, case["name"]
        assert_logical_eq(host(got), exp, case["name"])


def test_arith_temporal_matches_the_oracle(ctx, oracle):
    rng = np.random.default_rng(23)
    ts, tz = A.TimestampMicrosecond, A.Timestamp(A.MICROSECOND, "-08:00")
    pairs = [(ts, ts, (2, 3)), (tz, ts, (2, 3)), (tz, A.DurationMicrosecond, (0, 1, 2, 3)), (A.DurationMicrosecond, tz, (0, 1)),
             (A.DurationNanosecond, A.DurationNanosecond, (0, 1, 2, 3)), (A.Date64, A.Date64, (2, 3)), (A.Date32, A.Date32, (2, 3))]
    n = 3000
    for lt, rt, ops in pairs:
        def column(t, with_nulls):
            if t.np_dtype == np.int32:
                v = rng.integers(I32.min, I32.max, n, dtype=np.int64).astype(np.int32)
            else:
                v = rng.integers(-2**61, 2**61, n, dtype=np.int64)  # sums and differences stay inside i64
            return HostArray(t, v, (rng.random(n) < 0.8) if with_nulls else None)
        for ln, rn in ((False, False), (True, False), (True, True)):
            hl, hr = column(lt, ln), column(rt, rn)
            dl, dr = hl.to_device(ctx, bit_offset=5 if ln else 0), hr.to_device(ctx)
            for op in ops:
                exp = oracle.arith_with_types(op, hl, hr)
                got = ARITH_FN[op](dl, dr)
                assert got.data_type == exp.data_type, (lt, rt, op)
                check_exact(got, exp, f"{lt} {op} {rt} nulls={ln},{rn}")
            # scalar operands (Datum): a length-1 array on either side, and a null scalar
            for sc_valid in (True, False):
                hs = HostArray(rt, hr.values[:1], None if sc_valid else np.array([False]))
                exp = oracle.arith_with_types(ops[0], hl, hs, r_scalar=True)
                got = ARITH_FN[ops[0]](dl, A.Scalar(hs.to_device(ctx)))
                assert got.data_type == exp.data_type
                check_exact(got, exp, f"{lt} {ops[0]} scalar {rt} valid={sc_valid}")
    # overflow is reported for the first failing valid row, through the wrapping entry point too
    big = HostArray(ts, np.array([1, I64.max, I64.min, 5], dtype=np.int64), np.array([True, False, True, True]))
    one = HostArray(A.DurationMicrosecond, np.array([1, 1, 1, 1], dtype=np.int64))
    with pytest.raises(A.array.ArithmeticOverflow) as ei:
        K.sub_wrapping(big.to_device(ctx), one.to_device(ctx))
    assert ei.value.message == f"Overflow happened on: {I64.min} - 1"
    check_exact(K.add_wrapping(big.to_device(ctx), one.to_device(ctx)), oracle.arith_with_types(1, big, one))


def test_arith_temporal_refusals(ctx):
    ts, tz = A.TimestampSecond, A.Timestamp(A.SECOND, "+05:45")
    one = lambda t: HostArray.from_pylist([10, None, 30], t).to_device(ctx)  # noqa: E731
    for fn, l, r, msg in (
            (K.mul, ts, A.DurationSecond, "Invalid timestamp arithmetic operation: Timestamp(s) * Duration(s)"),
            (K.add, ts, ts, "Invalid timestamp arithmetic operation: Timestamp(s) + Timestamp(s)"),
            (K.sub, ts, A.TimestampMillisecond, "Invalid timestamp arithmetic operation: Timestamp(s) - Timestamp(ms)"),
            (K.add, tz, A.Int64, 'Invalid timestamp arithmetic operation: Timestamp(s, "+05:45") + Int64'),
            (K.sub, A.DurationSecond, ts, "Invalid arithmetic operation: Duration(s) - Timestamp(s)"),
            (K.add, A.DurationSecond, A.DurationMillisecond, "Invalid arithmetic operation: Duration(s) + Duration(ms)"),
            (K.div, A.DurationSecond, A.DurationSecond, "Invalid duration arithmetic operation: Duration(s) / Duration(s)"),
            (K.add, A.Date32, A.Date32, "Invalid date arithmetic operation: Date32 + Date32"),
            (K.add, A.DurationSecond, A.Date32, "Invalid date arithmetic operation: Date32 + Duration(s)"),
            (K.add, A.Int64, ts, "Invalid arithmetic operation: Int64 + Timestamp(s)"),
            (K.add, A.Time32Second, A.Time32Second, "Invalid arithmetic operation: Time32(s) + Time32(s)")):
        with pytest.raises(A.array.InvalidArgumentError) as ei:
            fn(one(l), one(r))
        assert ei.value.message == msg


# ----------------------------------------------------------- Decimal128 arithmetic
from test_temporal_cast_cpu import decimal_model, decimal_operands, dec_values  # noqa: E402


def test_decimal_arith_matches_the_oracle(ctx, oracle):
    rng = np.random.default_rng(41)
    n = 2500
    for (lt, rt, digits) in (((12, 3), (12, 1), 12), ((20, 0), (20, 0), 18), ((38, 10), (38, 2), 24), ((38, 6), (38, 6), 30),
                             ((10, -2), (15, 4), 10)):
        L_, R_ = A.Decimal128(*lt), A.Decimal128(*rt)
        lv, rv = decimal_operands(rng, n, digits), decimal_operands(rng, n, digits)
        rv = [v if v != 0 else 7 for v in rv]
        for ln, rn in ((False, False), (True, True)):
            hl = HostArray(L_, HostArray.from_pylist(lv, L_).values, (rng.random(n) < 0.8) if ln else None)
            hr = HostArray(R_, HostArray.from_pylist(rv, R_).values, (rng.random(n) < 0.8) if rn else None)
            dl, dr = hl.to_device(ctx, bit_offset=3 if ln else 0), hr.to_device(ctx)
            def same(op, ho_l, ho_r, dv_l, dv_r, tag, **flags):
                try:
                    exp = oracle.arith_with_types(op, ho_l, ho_r, **flags)
                except A.array.ArrowError as e:
                    with pytest.raises(type(e)) as ei:
                        ARITH_FN[op](dv_l, dv_r)
                    assert ei.value.message == e.message, tag
                    return
                got = ARITH_FN[op](dv_l, dv_r)
                assert got.data_type == exp.data_type, tag
                check_exact(got, exp, tag)
            for op in range(8):
                same(op, hl, hr, dl, dr, f"{L_} {op} {R_} nulls={ln}")
            # one scalar side, valid and null
            for sc_valid in (True, False):
                hs = HostArray(R_, hr.values[:1], None if sc_valid else np.array([False]))
                ds = A.Scalar(hs.to_device(ctx))
                same(0, hl, hs, dl, ds, f"{L_} + scalar valid={sc_valid}", r_scalar=True)
                same(4, hs, hl, ds, dl, f"scalar * {L_} valid={sc_valid}", l_scalar=True)
                same(6, hl, hs, dl, ds, f"{L_} / scalar valid={sc_valid}", r_scalar=True)


def test_decimal_arith_errors_name_the_first_failing_row(ctx, oracle):
    big = 10**37
    L_, R_ = A.Decimal128(38, 0), A.Decimal128(38, 2)
    vals = [1, big * 2, -big * 5, 3]
    hl = HostArray(L_, HostArray.from_pylist(vals, L_).values, np.array([True, False, True, True]))
    hr = HostArray.from_pylist([5, 5, 5, 5], R_)
    for op, fn in ((0, K.add), (6, K.div), (7, K.rem)):
        with pytest.raises(A.array.ArithmeticOverflow) as ei:
            fn(hl.to_device(ctx), hr.to_device(ctx))
        with pytest.raises(A.array.ArithmeticOverflow) as eo:
            oracle.arith_with_types(op, hl, hr)
        assert ei.value.message == eo.value.message
    assert eo.value.message == "Overflow happened on: -50000000000000000000000000000000000000 * 100"
    z = HostArray.from_pylist([1, 0, 2], A.Decimal128(5, 1))
    with pytest.raises(A.array.DivideByZero) as ei:
        K.div(HostArray.from_pylist([1, 2, 3], A.Decimal128(5, 1)).to_device(ctx), z.to_device(ctx))
    assert ei.value.message == "Divide by zero error"
    with pytest.raises(A.array.InvalidArgumentError) as ei:
        K.add(z.to_device(ctx), HostArray.from_pylist([1, 2, 3], A.Int64).to_device(ctx))
    assert ei.value.message == "Invalid arithmetic operation: Decimal128(5, 1) + Int64"


def test_decimal_arith_large_vs_exact_python(ctx):
    """2^22 rows: every row of add / mul / div against exact Python integers (vectorised through object arrays)."""
    n = 1 << 22
    rng = np.random.default_rng(43)
    lt, rt = (20, 4), (18, 2)
    l = rng.integers(-10**17, 10**17, n, dtype=np.int64)
    r = rng.integers(-10**15, 10**15, n, dtype=np.int64)
    r[r == 0] = 1

    def as_dec(v, t):
        vals = np.zeros(n, dtype=t.np_dtype)
        vals["lo"] = v.view(np.uint64)
        vals["hi"] = np.where(v < 0, -1, 0)
        return HostArray(t, vals).to_device(ctx)
    dl, dr = as_dec(l, A.Decimal128(*lt)), as_dec(r, A.Decimal128(*rt))
    lo, ro = l.astype(object), r.astype(object)
    for fn, exp, et in ((K.add, lo + ro * 100, (21, 4)), (K.sub, lo - ro * 100, (21, 4)), (K.mul, lo * ro, (38, 6)),
                        (K.rem, None, None), (K.div, None, (26, 8))):
        got = fn(dl, dr)
        v = got.values_numpy()
        gv = v["lo"].astype(object) + (v["hi"].astype(object) << 64)
        if fn is K.div:
            num, den = lo * 10**6, ro
            q = np.abs(num) // np.abs(den)
            exp = np.where((num < 0) != (den < 0), -q, q)
        elif fn is K.rem:
            num, den = lo, ro * 100
            m = np.abs(num) % np.abs(den)
            exp, et = np.where(num < 0, -m, m), (20, 4)
        assert got.data_type == A.Decimal128(*et), (fn.__name__, got.data_type)
        assert (gv == exp).all(), fn.__name__


def test_decimal_compare_matches_the_oracle(ctx, oracle):
    """cmp on Decimal128 (i128 order): every operator, nulls on either side, scalars, words that straddle 64 rows."""
    rng = np.random.default_rng(47)
    t = A.Decimal128(38, 4)
    n = 5003
    lv = decimal_operands(rng, n, 36)
    rv = decimal_operands(rng, n, 36)
    rv[:600] = lv[:600]
    lv[600:608] = [0, -1, 1, (1 << 127) - 1, -(1 << 127), 1 << 64, -(1 << 64), (1 << 64) - 1]
    rv[600:608] = [0, 1, -1, -(1 << 127), (1 << 127) - 1, (1 << 64) - 1, -(1 << 64) + 1, 1 << 64]
    CMP_FN = [K.eq, K.neq, K.lt, K.lt_eq, K.gt, K.gt_eq, K.distinct, K.not_distinct]
    for ln, rn in ((False, False), (True, False), (True, True)):
        hl = HostArray(t, HostArray.from_pylist(lv, t).values, (rng.random(n) < 0.8) if ln else None)
        hr = HostArray(t, HostArray.from_pylist(rv, t).values, (rng.random(n) < 0.8) if rn else None)
        dl, dr = hl.to_device(ctx, bit_offset=5 if ln else 0), hr.to_device(ctx)
        for op, fn in enumerate(CMP_FN):
            check_exact(fn(dl, dr), oracle.compare(op, hl, hr), f"decimal cmp {op} nulls={ln},{rn}")
        hs = HostArray(t, hr.values[:1])
        for op, fn in enumerate(CMP_FN):
            check_exact(fn(dl, A.Scalar(hs.to_device(ctx))), oracle.compare(op, hl, hs, r_scalar=True), f"decimal cmp scalar {op}")
            check_exact(fn(dl.slice(3, 4000), dr.slice(3, 4000)), oracle.compare(op, hl.slice(3, 4000), hr.slice(3, 4000)), f"sliced {op}")
    with pytest.raises(A.array.InvalidArgumentError) as ei:
        K.lt(dl, HostArray.from_pylist(rv, A.Decimal128(38, 2)).to_device(ctx))
    assert ei.value.message == "Invalid comparison operation: Decimal128(38, 4) < Decimal128(38, 2)"
    # the rule lives behind the C ABI (ah_compare_with_types, round 5): any two different LOGICAL types over one physical type
    ts_s = HostArray(A.Timestamp(A.SECOND), np.arange(5, dtype=np.int64)).to_device(ctx)
    ts_ms = HostArray(A.Timestamp(A.MILLISECOND), np.arange(5, dtype=np.int64)).to_device(ctx)
    with pytest.raises(A.array.InvalidArgumentError) as ei:
        K.gt_eq(ts_s, ts_ms)
    assert ei.value.message.startswith("Invalid comparison operation: Timestamp(") and " >= Timestamp(" in ei.value.message
    assert host(K.eq(ts_s, ts_s)).to_pylist() == [True] * 5  # equal logical types compare as their physical type


def test_decimal_to_decimal_cast_matches_the_oracle(ctx, oracle):
    """cast_decimal_to_decimal_same_type: widening, narrowing with half-away-from-zero rounding, precision overflow to
    null (safe) or the reference's error (unsafe), the infallible `unary` shapes, clone, and scale drops past 38 digits."""
    rng = np.random.default_rng(59)
    shapes = [((10, 2), (12, 4)), ((10, 2), (11, 4)), ((20, 4), (10, 1)), ((20, 4), (18, 1)), ((38, 10), (38, 20)), ((38, 0), (38, 0)),
              ((10, 3), (5, 3)), ((38, 30), (10, 0)), ((12, 5), (12, 5)), ((9, 2), (20, 2)), ((38, 38), (38, -2)), ((5, -3), (38, 38))]
    n = 2100
    for ft, tt in shapes:
        F, T = A.Decimal128(*ft), A.Decimal128(*tt)
        digits = min(ft[0], 37)
        vals = [int(rng.integers(-9, 10)) * 10 ** int(rng.integers(0, digits)) + int(rng.integers(-10**6, 10**6)) for _ in range(n)]
        vals[:10] = [0, 5, -5, 15, -15, 49, 50, -50, 10 ** digits - 1, -(10 ** digits - 1)]
        benign = [int(v) for v in rng.integers(-9999, 9999, n)]
        for data in (vals, benign):
            for with_nulls in (False, True):
                h = HostArray(F, HostArray.from_pylist(data, F).values, (rng.random(n) < 0.85) if with_nulls else None)
                d = h.to_device(ctx, bit_offset=3 if with_nulls else 0)
                for safe in (True, False):
                    tag = f"{F} -> {T} safe={safe} nulls={with_nulls}"
                    try:
                        exp = oracle.cast_with_types(h, T, safe=safe)
                    except A.array.ArrowError as e:
                        with pytest.raises(type(e)) as ei:
                            K.cast_with_options(d, T, K.CastOptions(safe=safe))
                        assert ei.value.message == e.message, tag
                        continue
                    got = K.cast_with_options(d, T, K.CastOptions(safe=safe))
                    assert got.data_type == T
                    check_exact(got, exp, tag)
    with pytest.raises(A.array.InvalidArgumentError) as ei:
        K.cast_with_options(HostArray.from_pylist([1, 123456, 7], A.Decimal128(10, 3)).to_device(ctx), A.Decimal128(5, 3), K.CastOptions(safe=False))
    assert ei.value.message == "123.456 is too large to store in a Decimal128 of precision 5. Max is 99.999"
    assert K.can_cast_types(A.Decimal128(10, 3), A.Decimal128(5, 0)) and not K.can_cast_types(A.Decimal128(10, 3), A.Int64)


def test_int_to_decimal_cast_matches_the_oracle(ctx, oracle):
    """cast_integer_to_decimal: every integer type x scales (positive, zero, negative, beyond the source type) x modes."""
    from test_temporal_cast_cpu import INT_TYPES, int_column
    rng = np.random.default_rng(67)
    n = 2300
    for src in INT_TYPES:
        for data in (int_column(rng, src, n), rng.integers(0, 100, n).astype(src.np_dtype)):
            for with_nulls in (False, True):
                h = HostArray(src, data, (rng.random(n) < 0.85) if with_nulls else None)
                d = h.to_device(ctx, bit_offset=3 if with_nulls else 0)
                for tt in ((38, 0), (38, 10), (20, 2), (10, 0), (5, 2), (38, 30), (10, -2), (38, -19), (38, -20), (3, -1)):
                    T = A.Decimal128(*tt)
                    for safe in (True, False):
                        tag = f"{src} -> {T} safe={safe} nulls={with_nulls}"
                        try:
                            exp = oracle.cast_with_types(h, T, safe=safe)
                        except A.array.ArrowError as e:
                            with pytest.raises(type(e)) as ei:
                                K.cast_with_options(d, T, K.CastOptions(safe=safe))
                            assert ei.value.message == e.message, tag
                            continue
                        got = K.cast_with_options(d, T, K.CastOptions(safe=safe))
                        assert got.data_type == T
                        check_exact(got, exp, tag)
    assert K.can_cast_types(A.Int64, A.Decimal128(20, 2)) and not K.can_cast_types(A.Float64, A.Decimal128(20, 2))
    # the cast feeds decimal arithmetic without leaving the device: (int -> decimal) * decimal
    price = K.cast(HostArray.from_pylist([3, None, 12], A.Int32).to_device(ctx), A.Decimal128(10, 2))
    qty = HostArray.from_pylist([250, 100, 50], A.Decimal128(6, 1)).to_device(ctx)
    total = K.mul(price, qty)
    assert total.data_type == A.Decimal128(17, 3) and host(total).to_pylist()[0] is not None
