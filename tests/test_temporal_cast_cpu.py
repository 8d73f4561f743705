"""Temporal casts (arrow-cast/src/cast/mod.rs:1700-2260), CPU side (no GPU):

* the oracle's arm-by-arm restatement (oracle/oracle.cpp cast_temporal) against the reference's own test vectors
  (tests/golden/cast_temporal.json, 102 cases) and against an independent model built on numpy's datetime64 /
  Python's datetime (a different calendar implementation than either the oracle's civil-from-days or the product's
  day bounds);
* the PRODUCT's planner and row closure (arrow-rs_amd/csrc/temporal_cast.hpp, the header cast_temporal.hip compiles
  for gfx950) compiled for the host by tests/cpp/temporal_cast_host_test.cpp and compared with the oracle on every
  (from, to, safe) pair over edge values and random values — values, validity bits, null-buffer PRESENCE and error
  texts — before the same source runs on a GPU;
* ``ah_can_cast_data_types`` (host-only code of the library) against the reference's ``can_cast_types`` arms.
"""
import datetime as dt
import itertools
import os
import subprocess

import numpy as np
import pytest

import arrow_rs_amd as A
import orc
from orc import HostArray, assert_logical_eq, golden_array, load_golden, lookup_type
from test_oracle_golden import expect_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
I64 = np.iinfo(np.int64)
I32 = np.iinfo(np.int32)
UNITS = [A.SECOND, A.MILLISECOND, A.MICROSECOND, A.NANOSECOND]
MULT = [1, 10**3, 10**6, 10**9]
ZONES = [None, "+00:00", "+01:00", "-07:00", "+0545", "-11", "+14:00"]


def all_types(zones=(None, "+05:45", "-08:00")):
    ts = [A.Timestamp(u, z) for u in UNITS for z in zones]
    return ([A.Int32, A.Int64, A.Date32, A.Date64, A.Time32Second, A.Time32Millisecond, A.Time64Microsecond,
             A.Time64Nanosecond] + ts + [A.Duration(u) for u in UNITS])


# ------------------------------------------------------------------ goldens
@pytest.mark.parametrize("case", load_golden("cast_temporal"), ids=lambda c: c["name"])
def test_cast_temporal_golden(oracle, case):
    v = golden_array(case["values"])
    to = lookup_type(case["to"])
    if "error" in case:
        return expect_err(case, lambda: oracle.cast_with_types(v, to, safe=case["safe"]))
    got = oracle.cast_with_types(v, to, safe=case["safe"])
    assert_logical_eq(got, golden_array(case["expected"]), case["name"])


def test_derived_golden_inputs_match_python_datetime():
    """The epoch values marked "derived" in make_golden.py really are the strings the reference tests parse."""
    epoch = dt.datetime(1970, 1, 1)

    def secs(*a, tz_hours=0):
        return int((dt.datetime(*a) - epoch).total_seconds()) - tz_hours * 3600

    cases = {c["name"]: c for c in load_golden("cast_temporal")}
    z = cases["test_cast_timestamp_to_date32_zone"]["values"]["data"]
    assert z == [secs(1970, 1, 1, 0, 0, 1, tz_hours=-7) * 1000, secs(1970, 1, 1, 23, 59, 59, tz_hours=-7) * 1000, None,
                 secs(2020, 3, 1, 2, 0, 23) * 1000]
    b = cases["test_cast_below_unixtimestamp"]
    assert b["values"]["data"] == [secs(1900, 1, 3, 23, 59, 59) * 1000, secs(1969, 12, 31, 0, 0, 1) * 1000,
                                   secs(1989, 12, 31, 0, 0, 1) * 1000]
    assert b["expected"]["data"] == [(dt.date(1900, 1, 3) - dt.date(1970, 1, 1)).days, -1,
                                     (dt.date(1989, 12, 31) - dt.date(1970, 1, 1)).days]
    t1 = cases["test_cast_timestamp_with_timezone_1"]
    assert t1["values"]["data"][:2] == [secs(2000, 1, 1) * 10**9 + 123456789, secs(2010, 1, 1) * 10**9 + 123456789]
    assert t1["expected"]["data"][:2] == [secs(2000, 1, 1, tz_hours=7) * 10**6 + 123456, secs(2010, 1, 1, tz_hours=7) * 10**6 + 123456]


# ------------------------------------------------- independent calendar model
def _np_unit(u):
    return ["s", "ms", "us", "ns"][u]


def test_oracle_timestamp_to_date_and_time_vs_numpy_datetime64(oracle):
    rng = np.random.default_rng(11)
    for u in UNITS:
        # +-250 000 years for s / ms / us; the whole i64 range for ns
        span = min(I64.max, 250_000 * 366 * 86400 * MULT[u])
        vals = rng.integers(-span, span, 3000, dtype=np.int64)
        vals[:6] = [0, -1, 1, 86400 * MULT[u] - 1, -86400 * MULT[u], -86400 * MULT[u] - 1]
        for tz in (None, "+05:45", "-08:00"):
            off = A.parse_fixed_offset(tz) if tz else 0
            src = HostArray(A.Timestamp(u, tz), vals)
            got = oracle.cast_with_types(src, A.Date32, safe=True)
            # numpy: floor to seconds, shift, floor to days — datetime64 arithmetic is proleptic Gregorian
            sec = vals // MULT[u]
            exp_days = (sec + off) // 86400
            np_days = (sec + off).astype("datetime64[s]").astype("datetime64[D]").astype(np.int64)
            assert np.array_equal(exp_days, np_days)
            assert got.valid is None and np.array_equal(got.values, exp_days.astype(np.int32))
            for to, tmul in ((A.Time32Second, 1), (A.Time32Millisecond, 10**3), (A.Time64Microsecond, 10**6),
                             (A.Time64Nanosecond, 10**9)):
                got = oracle.cast_with_types(src, to)
                sod = (sec + off) % 86400
                sub_ns = (vals % MULT[u]) * (10**9 // MULT[u])
                exp = sod * tmul + sub_ns // (10**9 // tmul)
                assert np.array_equal(got.values.astype(np.int64), exp), (u, tz, to)


def test_oracle_small_range_vs_python_datetime(oracle):
    """Years 1..9999: Python's own calendar, field by field."""
    rng = np.random.default_rng(5)
    lo = int((dt.datetime(1, 1, 2) - dt.datetime(1970, 1, 1)).total_seconds())
    hi = int((dt.datetime(9999, 12, 30) - dt.datetime(1970, 1, 1)).total_seconds())
    secs = rng.integers(lo, hi, 500)
    us = rng.integers(0, 10**6, 500)
    vals = (secs * 10**6 + us).astype(np.int64)
    for tz in (None, "+05:45", "-08:00"):
        off = A.parse_fixed_offset(tz) if tz else 0
        src = HostArray(A.Timestamp(A.MICROSECOND, tz), vals)
        d = oracle.cast_with_types(src, A.Date32).values
        t = oracle.cast_with_types(src, A.Time64Microsecond).values
        for i in range(len(vals)):
            local = dt.datetime(1970, 1, 1) + dt.timedelta(seconds=int(secs[i]) + off, microseconds=int(us[i]))
            assert d[i] == (local.date() - dt.date(1970, 1, 1)).days
            assert t[i] == ((local.hour * 60 + local.minute) * 60 + local.second) * 10**6 + local.microsecond


def test_oracle_calendar_limits(oracle):
    """chrono's NaiveDate::MIN / MAX (-262143-01-01 / +262142-12-31): first and last representable seconds."""
    def days_from_civil(y, m, d):  # Hinnant
        y -= m <= 2
        era = y // 400  # Python's // already floors
        yoe = y - era * 400
        doy = (153 * (m + (-3 if m > 2 else 9)) + 2) // 5 + d - 1
        return era * 146097 + yoe * 365 + yoe // 4 - yoe // 100 + doy - 719468

    lo, hi = days_from_civil(-262143, 1, 1), days_from_civil(262142, 12, 31)
    ok = HostArray(A.TimestampSecond, np.array([lo * 86400, hi * 86400 + 86399], dtype=np.int64))
    assert oracle.cast_with_types(ok, A.Date32).to_pylist() == [lo, hi]
    for bad in (lo * 86400 - 1, hi * 86400 + 86400, I64.max, I64.min):
        with pytest.raises(A.array.CastError) as ei:
            oracle.cast_with_types(HostArray(A.TimestampSecond, np.array([bad], dtype=np.int64)), A.Date32)
        assert ei.value.message == f"Cannot convert arrow_array::types::TimestampSecondType {bad} to datetime"
    # microseconds reach past the calendar too; nanoseconds never do
    with pytest.raises(A.array.CastError):
        oracle.cast_with_types(HostArray(A.TimestampMicrosecond, np.array([I64.max], dtype=np.int64)), A.Time64Nanosecond)
    assert oracle.cast_with_types(HostArray(A.TimestampNanosecond, np.array([I64.max, I64.min], dtype=np.int64)),
                                  A.Date32).to_pylist() == [106751, -106752]


# ------------------------------------------- product header on the host vs oracle
@pytest.fixture(scope="module")
def host_harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("tc") / "temporal_cast_host_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "temporal_cast_host_test.cpp")], check=True)
    return exe


def _edge_values(dtype_type):
    info = I32 if dtype_type.np_dtype == np.int32 else I64
    day_lo, day_hi = -96465292, 95026236
    edges = [0, 1, -1, 999, 1000, -999, -1000, -1001, 86399, 86400, -86400, -86401, 86_400_000, -86_400_001,
             info.max, info.min, info.max - 1, info.min + 1, info.max // 1000, info.max // 1000 + 1, info.min // 1000,
             info.min // 1000 - 1, 106751, 106752, -106752, -106753, 106751991, 106751992]
    if info is I64:
        for m in MULT:
            edges += [day_lo * 86400 * m, day_lo * 86400 * m - 1, (day_hi + 1) * 86400 * m - 1, (day_hi + 1) * 86400 * m,
                      I64.max // m, I64.max // m + 1, I64.min // m, I64.min // m - 1, 2**31 * 86_400_000, -2**31 * 86_400_000 - 1]
        edges += [2**31, -2**31 - 1, 2**31 - 1, -2**31]
    return [int(np.clip(e, info.min, info.max)) for e in edges]


def _run_harness(exe, cases):
    """cases: list of (HostArray, to_type, safe).  Returns one parsed result per case."""
    lines = []
    for src, to, safe in cases:
        f, t = src.data_type.descriptor(), to.descriptor()
        valid = src.valid if src.valid is not None else np.ones(len(src), dtype=bool)
        lines.append(f"case {f.id} {f.unit} {f.has_tz} {f.tz_offset_seconds} {t.id} {t.unit} {t.has_tz} {t.tz_offset_seconds} "
                     f"{int(safe)} {int(src.valid is not None)} {len(src)}")
        lines += [f"{int(v)} {int(b)}" for v, b in zip(src.values, valid)]
    r = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    out = r.stdout.split("\n")
    res, i = [], 0
    for _ in cases:
        head = out[i].split(" ", 2)
        i += 1
        if head[0] == "unsupported":
            res.append(("error", A.L.AH_CAST_ERROR, out[i - 1][len("unsupported "):]))
        elif head[0] == "error":
            res.append(("error", int(head[1]), head[2]))
        else:
            _, phys, has_valid, n = out[i - 1].split()
            rows = [ln.split() for ln in out[i:i + int(n)]]
            i += int(n)
            res.append(("ok", int(phys), bool(int(has_valid)), np.array([int(a) for a, _ in rows], dtype=object),
                        np.array([bool(int(b)) for _, b in rows], dtype=bool)))
    return res


def _oracle_result(oracle, src, to, safe):
    try:
        got = oracle.cast_with_types(src, to, safe=safe)
    except A.array.ArrowError as e:
        status = {v: k for k, v in A.array._STATUS.items() if k != A.L.AH_OFFSET_OVERFLOW}[type(e)]
        return ("error", status, e.message)
    return ("ok", got)


def test_product_header_on_host_matches_oracle(oracle, host_harness):
    rng = np.random.default_rng(3)
    types = all_types()
    cases = []
    for f, t in itertools.product(types, types):
        if f.logical is None and t.logical is None:
            continue  # plain numeric pairs never reach the planner (ah_cast_with_types forwards them to ah_cast)
        if f.np_dtype == np.int32:
            rnd = rng.integers(I32.min, I32.max, 40, dtype=np.int64)
        else:
            rnd = np.concatenate([rng.integers(I64.min, I64.max, 20, dtype=np.int64),
                                  rng.integers(-4 * 10**12, 4 * 10**12, 20, dtype=np.int64)])
        vals = np.array(_edge_values(f) + [int(x) for x in rnd], dtype=f.np_dtype)
        for with_nulls in (False, True):
            valid = (rng.random(len(vals)) < 0.8) if with_nulls else None
            for safe in (True, False):
                cases.append((HostArray(f, vals, valid), t, safe))
                if not safe:
                    # unsafe mode stops at the first failure: also a benign input so the arm's success path is compared
                    benign = np.array([int(x) for x in rng.integers(-10**5, 10**5, 48)], dtype=f.np_dtype)
                    cases.append((HostArray(f, benign, valid[:48] if with_nulls else None), t, safe))
    results = _run_harness(host_harness, cases)
    n_ok = n_err = 0
    for (src, to, safe), res in zip(cases, results):
        exp = _oracle_result(oracle, src, to, safe)
        tag = f"{src.data_type} -> {to} safe={safe} nulls={src.valid is not None}"
        assert res[0] == exp[0], (tag, res[:3], exp[:3])
        if res[0] == "error":
            assert (res[1], res[2]) == (exp[1], exp[2]), tag
            n_err += 1
            continue
        got = exp[1]
        assert res[1] == to.physical, tag
        assert res[2] == (got.valid is not None), f"{tag}: null buffer presence"
        ev = got.valid if got.valid is not None else np.ones(len(got), dtype=bool)
        assert np.array_equal(res[4], ev), tag
        assert all(int(a) == int(b) for a, b, v in zip(res[3], got.values, ev) if v), tag
        n_ok += 1
    assert n_ok > 1500 and n_err > 500, (n_ok, n_err)


# ------------------------------------------------------------ can_cast_types
def test_can_cast_data_types_matches_reference_arms():
    from arrow_rs_amd.compute.kernels.cast import can_cast_types
    ts, tz = A.TimestampSecond, A.Timestamp(A.NANOSECOND, "+01:00")
    yes = [(A.Int32, A.Date32), (A.Int32, A.Date64), (A.Int32, A.Time32Second), (A.Date32, A.Int32), (A.Date32, A.Int64),
           (A.Time32Millisecond, A.Int64), (A.Int64, A.Date64), (A.Int64, A.Date32), (A.Int64, A.Time64Nanosecond),
           (A.Date64, A.Int32), (A.Time64Microsecond, A.Int64), (A.Date32, A.Date64), (A.Date64, A.Date32),
           (A.Time32Second, A.Time64Nanosecond), (A.Time64Nanosecond, A.Time32Second), (ts, A.Float64), (A.UInt8, tz),
           (A.Date64, tz), (A.Date32, ts), (ts, tz), (tz, A.Date32), (tz, A.Date64), (ts, A.Time32Millisecond),
           (tz, A.Time64Microsecond), (A.Float32, A.DurationSecond), (A.DurationNanosecond, A.Int8),
           (A.DurationSecond, A.DurationNanosecond), (ts, ts)]
    # cast/mod.rs:295-323: what the reference's can_cast_types refuses among these types
    no = [(A.Int32, A.Time64Microsecond), (A.Int64, A.Time32Second), (A.Time64Nanosecond, A.Int32), (A.Time32Second, A.Date32),
          (A.Date32, A.Time32Second), (A.DurationSecond, ts), (ts, A.DurationSecond), (A.Date32, A.DurationSecond),
          (A.Boolean, ts), (A.Time32Second, A.Float64)]
    for f, t in yes:
        assert can_cast_types(f, t), (f, t)
    for f, t in no:
        assert not can_cast_types(f, t), (f, t)


def test_named_zone_is_the_reference_parse_error():
    with pytest.raises(A.array.ParseError) as ei:
        A.Timestamp(A.SECOND, "Europe/Berlin").descriptor()
    assert ei.value.message == ('Invalid timezone "Europe/Berlin": only offset based timezones supported without '
                                "chrono-tz feature")
    for text, secs in (("+09:00", 32400), ("-09", -32400), ("+0930", 34200), ("-00:30", -1800)):
        assert A.parse_fixed_offset(text) == secs
    for bad in ("0900", "+9", "+09:0", "+0a:00", "UTC", "+24:00", ""):
        with pytest.raises(A.array.ParseError):
            A.parse_fixed_offset(bad)


# ------------------------------------------------------- temporal arithmetic
@pytest.mark.parametrize("case", load_golden("arith_temporal"), ids=lambda c: c["name"])
def test_arith_temporal_golden(oracle, case):
    l, r = golden_array(case["lhs"]), golden_array(case["rhs"])
    if "error" in case:
        return expect_err(case, lambda: oracle.arith_with_types(case["op"], l, r))
    got = oracle.arith_with_types(case["op"], l, r)
    assert_logical_eq(got, golden_array(case["expected"]), case["name"])
    if case["op"] in (0, 2, 4):  # the *_wrapping entry points are checked too (add_checked / sub_checked / mul_checked)
        assert_logical_eq(oracle.arith_with_types(case["op"] + 1, l, r), golden_array(case["expected"]), case["name"] + " wrapping")


def test_arith_temporal_type_rules_on_the_oracle(oracle):
    ts, tz = A.TimestampSecond, A.Timestamp(A.SECOND, "+05:45")
    one = lambda t: HostArray.from_pylist([10, None, 30], t)  # noqa: E731
    # result types
    assert oracle.arith_with_types(2, one(tz), one(ts)).data_type == A.DurationSecond
    assert oracle.arith_with_types(0, one(tz), one(A.DurationSecond)).data_type == tz      # the left zone is kept
    assert oracle.arith_with_types(1, one(A.DurationSecond), one(tz)).data_type == tz      # Duration + Timestamp swaps
    assert oracle.arith_with_types(2, one(A.Date32), one(A.Date32)).data_type == A.DurationSecond
    assert oracle.arith_with_types(3, one(A.Date64), one(A.Date64)).data_type == A.DurationMillisecond
    # the reference's refusals (numeric.rs:528-533, :886-890, :963-967, :270-272)
    for op, l, r, msg in (
            (4, ts, A.DurationSecond, "Invalid timestamp arithmetic operation: Timestamp(s) * Duration(s)"),
            (0, ts, ts, "Invalid timestamp arithmetic operation: Timestamp(s) + Timestamp(s)"),
            (2, ts, A.TimestampMillisecond, "Invalid timestamp arithmetic operation: Timestamp(s) - Timestamp(ms)"),
            (0, ts, A.DurationMillisecond, "Invalid timestamp arithmetic operation: Timestamp(s) + Duration(ms)"),
            (0, tz, A.Int64, 'Invalid timestamp arithmetic operation: Timestamp(s, "+05:45") + Int64'),
            (2, A.DurationSecond, ts, "Invalid arithmetic operation: Duration(s) - Timestamp(s)"),
            (0, A.DurationSecond, A.DurationMillisecond, "Invalid arithmetic operation: Duration(s) + Duration(ms)"),
            (0, A.Date32, A.Date32, "Invalid date arithmetic operation: Date32 + Date32"),
            (2, A.Date32, A.Date64, "Invalid date arithmetic operation: Date32 - Date64"),
            (0, A.DurationSecond, A.Date32, "Invalid date arithmetic operation: Date32 + Duration(s)"),
            (0, A.Int64, ts, "Invalid arithmetic operation: Int64 + Timestamp(s)"),
            (0, A.Time32Second, A.Time32Second, "Invalid arithmetic operation: Time32(s) + Time32(s)")):
        with pytest.raises(A.array.InvalidArgumentError) as ei:
            oracle.arith_with_types(op, one(l), one(r))
        assert ei.value.message == msg
    # null rules are try_op!'s: union of the nulls; a null scalar gives an all-null result
    got = oracle.arith_with_types(2, HostArray.from_pylist([5, None, 7], ts), HostArray.from_pylist([1, 2, None], ts))
    assert got.to_pylist() == [4, None, None]
    got = oracle.arith_with_types(2, one(A.Date32), HostArray.from_pylist([None], A.Date32), r_scalar=True)
    assert got.to_pylist() == [None, None, None] and got.data_type == A.DurationSecond
    got = oracle.arith_with_types(2, one(A.Date32), HostArray.from_pylist([3], A.Date32), r_scalar=True)
    assert got.to_pylist() == [7 * 86400, None, 27 * 86400]


# ----------------------------------------------------------- Decimal128 arithmetic
I128_MIN, I128_MAX = -(1 << 127), (1 << 127) - 1


def test_decimal_limb_arithmetic_header_on_host(tmp_path):
    """arrow-rs_amd/csrc/decimal_arith.hpp (checked 128-bit multiply and long division on 64-bit limbs — the device
    runtime has neither) against the compiler's native __int128 arithmetic."""
    exe = str(tmp_path / "decimal_arith_host_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "decimal_arith_host_test.cpp")], check=True)
    r = subprocess.run([exe, "300000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
    assert int(r.stdout.split()[1]) > 2_000_000


def decimal_model(op, l, lt, r, rt):
    """decimal_op (numeric.rs:971-1103) in exact Python integers: (result precision, scale, value) or an error tuple."""
    (p1, s1), (p2, s2) = lt, rt
    chk = lambda v: I128_MIN <= v <= I128_MAX  # noqa: E731
    if op in (0, 1, 2, 3):
        rs = max(s1, s2)
        prec = min(rs + max(p1 - s1, p2 - s2) + 1, 38)
        lm, rm = 10 ** (rs - s1), 10 ** (rs - s2)
        if s1 != s2:
            if not chk(l * lm):
                return ("ArithmeticOverflow", f"Overflow happened on: {l} * {lm}")
            if not chk(r * rm):
                return ("ArithmeticOverflow", f"Overflow happened on: {r} * {rm}")
            l, r = l * lm, r * rm
        v = l + r if op < 2 else l - r
        sym = "+" if op < 2 else "-"
    elif op in (4, 5):
        prec, rs, v, sym = min(p1 + p2 + 1, 38), s1 + s2, l * r, "*"
    elif op == 6:
        rs = min(s1 + 4, 38)
        mp = rs - s1 + s2
        prec = min(mp + p1, 38)
        lm, rm = (10 ** mp, 1) if mp > 0 else (1, 10 ** -mp)
        if not chk(l * lm):
            return ("ArithmeticOverflow", f"Overflow happened on: {l} * {lm}")
        if not chk(r * rm):
            return ("ArithmeticOverflow", f"Overflow happened on: {r} * {rm}")
        l, r, sym = l * lm, r * rm, "/"
        if r == 0:
            return ("DivideByZero", "Divide by zero error")
        v = abs(l) // abs(r) * (1 if (l < 0) == (r < 0) else -1)
    else:
        rs = max(s1, s2)
        prec = min(rs + min(p1 - s1, p2 - s2), 38)
        lm, rm = 10 ** (rs - s1), 10 ** (rs - s2)
        if not chk(l * lm):
            return ("ArithmeticOverflow", f"Overflow happened on: {l} * {lm}")
        if not chk(r * rm):
            return ("ArithmeticOverflow", f"Overflow happened on: {r} * {rm}")
        l, r, sym = l * lm, r * rm, "%"
        if r == 0:
            return ("DivideByZero", "Divide by zero error")
        v = abs(l) % abs(r) * (1 if l >= 0 else -1)
    if not chk(v):
        return ("ArithmeticOverflow", f"Overflow happened on: {l} {sym} {r}")
    return (prec, rs, v)


def decimal_operands(rng, n, digits):
    return [int(rng.integers(-9, 10)) * 10 ** int(rng.integers(0, digits)) + int(rng.integers(-10**9, 10**9)) for _ in range(n)]


def dec_values(h):
    return [int(lo) | (int(hi) << 64) for lo, hi in zip(h.values["lo"], h.values["hi"])]


def test_oracle_decimal_vs_exact_python_integers(oracle):
    rng = np.random.default_rng(31)
    for (lt, rt, digits) in (((12, 3), (12, 1), 12), ((20, 0), (20, 0), 20), ((38, 10), (38, 2), 30), ((38, 0), (38, 0), 38),
                             ((10, -2), (15, 4), 10), ((38, 20), (38, 18), 36)):
        L_, R_ = A.Decimal128(*lt), A.Decimal128(*rt)
        for op in range(8):
            for _ in range(40):
                l, r = decimal_operands(rng, 1, digits)[0], decimal_operands(rng, 1, digits)[0]
                if rng.random() < 0.1:
                    r = 0
                exp = decimal_model(op, l, lt, r, rt)
                hl, hr = HostArray.from_pylist([l], L_), HostArray.from_pylist([r], R_)
                if isinstance(exp[0], str):
                    with pytest.raises(getattr(A.array, exp[0])) as ei:
                        oracle.arith_with_types(op, hl, hr)
                    assert ei.value.message == exp[1], (op, l, r, lt, rt)
                else:
                    got = oracle.arith_with_types(op, hl, hr)
                    assert got.data_type == A.Decimal128(exp[0], exp[1]), (op, lt, rt, got.data_type)
                    assert dec_values(got) == [exp[2]], (op, l, r, lt, rt)


def test_oracle_decimal_compare_vs_python(oracle):
    rng = np.random.default_rng(37)
    t = A.Decimal128(38, 4)
    l = decimal_operands(rng, 400, 36) + [0, -1, 1, I128_MAX, I128_MIN, 1 << 64, -(1 << 64), (1 << 64) - 1]
    r = decimal_operands(rng, 400, 36) + [0, 1, -1, I128_MIN, I128_MAX, (1 << 64) - 1, -(1 << 64) + 1, 1 << 64]
    r[:50] = l[:50]
    hl, hr = HostArray.from_pylist(l, t), HostArray.from_pylist(r, t)
    import operator
    for op, f in enumerate((operator.eq, operator.ne, operator.lt, operator.le, operator.gt, operator.ge, operator.ne, operator.eq)):
        got = oracle.compare(op, hl, hr)
        assert got.to_pylist() == [f(a, b) for a, b in zip(l, r)], op


# ----------------------------------------------------------- Decimal128 -> Decimal128 cast
def decimal_cast_model(x, ft, tt, safe):
    """cast_decimal_to_decimal_same_type (cast/decimal.rs:448-489) in exact Python integers: value, None (null) or an error tuple."""
    (ip, is_), (op, os) = ft, tt
    mx = 10 ** op - 1
    if is_ == os and ip <= op:
        return x
    if is_ <= os:
        k = 10 ** (os - is_)
        v = x * k
        if ip + (os - is_) <= op:
            return ((v + (1 << 127)) % (1 << 128)) - (1 << 127)
        if not (I128_MIN <= v <= I128_MAX):
            return None if safe else ("CastError", f"Cannot cast to Decimal128({op}, {os}). Overflowing on {x}")
    else:
        k = 10 ** (is_ - os)
        d, r = abs(x) // k, abs(x) % k
        if r >= k // 2:
            d += 1
        v = d if x >= 0 else -d
        if ip - (is_ - os) < op:
            return v
    if -mx <= v <= mx:
        return v
    return None if safe else ("InvalidArgumentError", "too large" if v > mx else "too small")


def test_oracle_decimal_cast_vs_exact_python_integers(oracle):
    rng = np.random.default_rng(53)
    shapes = [((10, 2), (12, 4)), ((10, 2), (11, 4)), ((20, 4), (10, 1)), ((20, 4), (18, 1)), ((38, 10), (38, 20)), ((38, 0), (38, 0)),
              ((10, 3), (5, 3)), ((38, 30), (10, 0)), ((12, 5), (12, 5)), ((9, 2), (20, 2))]
    for ft, tt in shapes:
        F, T = A.Decimal128(*ft), A.Decimal128(*tt)
        vals = [int(rng.integers(-9, 10)) * 10 ** int(rng.integers(0, ft[0])) + int(rng.integers(-10**6, 10**6)) for _ in range(300)]
        vals += [0, 5, -5, 15, -15, 49, 50, -50, 10 ** ft[0] - 1, -(10 ** ft[0] - 1)]
        valid = rng.random(len(vals)) < 0.85
        h = HostArray(F, HostArray.from_pylist(vals, F).values, valid)
        exp = [decimal_cast_model(v, ft, tt, True) if ok else None for v, ok in zip(vals, valid)]
        got = oracle.cast_with_types(h, T, safe=True)
        gv = dec_values(got)
        gvalid = got.valid if got.valid is not None else np.ones(len(vals), dtype=bool)
        assert [v if ok else None for v, ok in zip(gv, gvalid)] == exp, (ft, tt)
        # unsafe: the first failing valid row is the error
        errs = [decimal_cast_model(v, ft, tt, False) for v, ok in zip(vals, valid) if ok]
        first = next((e for e in errs if isinstance(e, tuple)), None)
        if first is None:
            unsafe = oracle.cast_with_types(h, T, safe=False)
            uv = unsafe.valid if unsafe.valid is not None else np.ones(len(vals), dtype=bool)
            assert [v if ok else None for v, ok in zip(dec_values(unsafe), uv)] == exp, (ft, tt)
        else:
            with pytest.raises(getattr(A.array, first[0])) as ei:
                oracle.cast_with_types(h, T, safe=False)
            assert first[1] in ei.value.message
    # the reference's own texts (decimal.rs:330-349, arrow-data/src/decimal.rs:1036-1057)
    with pytest.raises(A.array.InvalidArgumentError) as ei:
        oracle.cast_with_types(HostArray.from_pylist([123456], A.Decimal128(10, 3)), A.Decimal128(5, 3), safe=False)
    assert ei.value.message == "123.456 is too large to store in a Decimal128 of precision 5. Max is 99.999"
    with pytest.raises(A.array.InvalidArgumentError) as ei:
        oracle.cast_with_types(HostArray.from_pylist([-123456], A.Decimal128(10, 3)), A.Decimal128(5, 3), safe=False)
    assert ei.value.message == "-123.456 is too small to store in a Decimal128 of precision 5. Min is -99.999"
    with pytest.raises(A.array.CastError) as ei:
        oracle.cast_with_types(HostArray.from_pylist([10**37], A.Decimal128(38, 0)), A.Decimal128(38, 5), safe=False)
    assert ei.value.message == f"Cannot cast to Decimal128(38, 5). Overflowing on {10**37}"


# ----------------------------------------------------------- integer -> Decimal128 cast
INT_TYPES = [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64]


def int_to_decimal_model(v, src, tt):
    """cast_integer_to_decimal (cast/mod.rs:366-443), safe mode, in exact Python integers: value or None."""
    p, s = tt
    mx = 10 ** p - 1
    if s < 0:
        k = 10 ** -s
        if k > np.iinfo(src.np_dtype).max:
            return 0
        r = abs(v) // k * (1 if v >= 0 else -1)
    else:
        r = v * 10 ** s
        if not (I128_MIN <= r <= I128_MAX):
            return None
    return r if -mx <= r <= mx else None


def int_column(rng, t, n):
    info = np.iinfo(t.np_dtype)
    v = rng.integers(info.min, info.max, n, dtype=np.int64 if info.min < 0 else np.uint64).astype(t.np_dtype)
    v[:6] = [0, 1, info.max, info.min, info.max // 10, 99 if info.max >= 99 else 9]
    return v


def test_oracle_int_to_decimal_vs_exact_python_integers(oracle):
    rng = np.random.default_rng(61)
    for src in INT_TYPES:
        vals = int_column(rng, src, 200)
        valid = rng.random(200) < 0.85
        h = HostArray(src, vals, valid)
        for tt in ((38, 0), (38, 10), (20, 2), (10, 0), (5, 2), (38, 30), (10, -2), (38, -19), (38, -20), (3, -1)):
            got = oracle.cast_with_types(h, A.Decimal128(*tt), safe=True)
            exp = [int_to_decimal_model(int(v), src, tt) if ok else None for v, ok in zip(vals, valid)]
            gvalid = got.valid if got.valid is not None else np.ones(200, dtype=bool)
            assert [v if ok else None for v, ok in zip(dec_values(got), gvalid)] == exp, (src, tt)
    h = HostArray.from_pylist([1, 123456, 7], A.Int32)
    with pytest.raises(A.array.InvalidArgumentError) as ei:
        oracle.cast_with_types(h, A.Decimal128(5, 0), safe=False)
    assert ei.value.message == "123456 is too large to store in a Decimal128 of precision 5. Max is 99999"
    with pytest.raises(A.array.ArithmeticOverflow) as ei:
        oracle.cast_with_types(HostArray.from_pylist([2**62], A.Int64), A.Decimal128(38, 30), safe=False)
    assert ei.value.message == f"Overflow happened on: {2**62} * {10**30}"
    # test_cast_integer_to_decimal-style literals (cast/mod.rs: Int32 [1, 2, 3, 4, 5, None] -> Decimal128(38, 6))
    got = oracle.cast_with_types(HostArray.from_pylist([1, 2, 3, 4, 5, None], A.Int32), A.Decimal128(38, 6))
    assert dec_values(got)[:5] == [1000000, 2000000, 3000000, 4000000, 5000000] and got.valid.tolist() == [True] * 5 + [False]
