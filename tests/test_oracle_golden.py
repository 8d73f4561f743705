"""CPU suite: pins the oracle (oracle/oracle.cpp) against the reference's own inline golden
vectors (tests/golden/*.json, transcribed from the cited Rust tests) and cross-checks it
against pyarrow (Arrow C++) where semantics coincide."""
import math
import struct

import numpy as np
import pytest

import arrow_rs_amd as A
import orc
from orc import HostArray, golden_array, load_golden, assert_logical_eq

ARITH = {"add": 0, "add_wrapping": 1, "sub": 2, "sub_wrapping": 3, "mul": 4, "mul_wrapping": 5, "div": 6, "rem": 7}
CMP = {"eq": 0, "neq": 1, "lt": 2, "lt_eq": 3, "gt": 4, "gt_eq": 5, "distinct": 6, "not_distinct": 7}
ERR = {"InvalidArgumentError": A.array.InvalidArgumentError, "ComputeError": A.array.ComputeError,
       "ArithmeticOverflow": A.array.ArithmeticOverflow, "DivideByZero": A.array.DivideByZero,
       "CastError": A.array.CastError}


def expect_err(case, fn):
    if "panic" in case:
        with pytest.raises(A.Panic) as ei:
            fn()
        assert str(ei.value) == case["panic"]
    else:
        with pytest.raises(ERR[case["error"]]) as ei:
            fn()
        assert ei.value.message == case["message"]
        if "display" in case:
            assert str(ei.value) == case["display"]


# ------------------------------------------------------------------ filter
@pytest.mark.parametrize("case", [c for c in load_golden("filter") if "values" in c], ids=lambda c: c["name"])
@pytest.mark.parametrize("bit_offset", [0, 3])
def test_filter_golden(oracle, case, bit_offset):
    v, p = golden_array(case["values"]), golden_array(case["predicate"])
    if "error" in case:
        return expect_err(case, lambda: oracle.filter(v, p, bit_offset))
    got = oracle.filter(v, p, bit_offset, elem_offset=1 if bit_offset else 0)
    assert_logical_eq(got, golden_array(case["expected"]), case["name"])
    if isinstance(v.values, list):  # string columns: also against the pure-Python model
        keep = p.values & (p.valid if p.valid is not None else True)
        assert got.to_pylist() == [x for x, k in zip(v.to_pylist()[:len(p)], keep) if k]
    if "expected_null_count" in case:
        assert got.null_count == case["expected_null_count"]


def test_filter_record_batch_no_columns(oracle):
    case = next(c for c in load_golden("filter") if c["name"] == "test_filter_record_batch_no_columns")
    p = golden_array(case["predicate"])
    # row count of a column-less batch == FilterPredicate::count == true_count(values & validity)
    dummy = HostArray(A.Int8, np.zeros(len(p), dtype=np.int8))
    assert len(oracle.filter(dummy, p)) == case["expected_rows"]


def test_slices_golden(oracle):
    case = next(c for c in load_golden("filter") if c["name"] == "test_slices")
    m = np.array(case["mask"], dtype=bool)
    assert oracle.set_slices(m) == [tuple(x) for x in case["slices"]]
    s = case["sliced"]
    sub = m[s["offset"]:s["offset"] + s["length"]]
    # the sliced BooleanArray keeps the parent buffer: emulate with a bit offset
    assert oracle.set_slices(sub, bit_offset=s["offset"]) == [tuple(x) for x in s["slices"]]


def test_fuzz_slices_and_index_iterators(oracle):
    """fuzz_test_slices_iterator (filter.rs:1784-1841): iterators agree with a bool-vec model."""
    rng = np.random.default_rng(42)
    fixed = [(64, 0, 0), (64, 8, 0), (64, 8, 8), (32, 8, 8), (32, 5, 9)]
    cases = fixed + [(int(rng.integers(0, 1024)), int(rng.integers(0, 64)), int(rng.integers(0, 128)))
                     for _ in range(200)]
    for mask_len, offset, trunc in cases:
        full = rng.random(mask_len + offset + trunc) < rng.random()
        m = full[offset:offset + mask_len]
        idx = [int(i) for i in np.nonzero(m)[0]]
        assert oracle.set_indices(m, bit_offset=offset) == idx
        runs, start = [], None
        for i, b in enumerate(m):
            if b and start is None:
                start = i
            if not b and start is not None:
                runs.append((start, i))
                start = None
        if start is not None:
            runs.append((start, len(m)))
        assert oracle.set_slices(m, bit_offset=offset) == runs
        assert oracle.count_set_bits(m, bit_offset=offset) == len(idx)


def test_fuzz_filter_vs_model(oracle):
    """fuzz_filter (filter.rs:1890-1977) against the naive model filter_rust (:1844-1851)."""
    rng = np.random.default_rng(7)
    for it in range(100):
        n = int(rng.integers(32, 256))
        sel = 1.0 if it < 5 else (0.0 if it <= 10 else rng.random())
        vp = rng.random()
        vals = rng.integers(-2**31, 2**31, n, dtype=np.int64).astype(np.int32)
        valid = rng.random(n) < vp
        plen = n - int(rng.integers(0, 10))
        pred = rng.random(plen) < sel
        pvalid = (rng.random(plen) < 0.9) if it % 3 == 0 else None
        v = HostArray(A.Int32, vals, valid if it % 2 == 0 else None)
        p = HostArray(A.Boolean, pred, pvalid)
        got = oracle.filter(v, p, bit_offset=int(rng.integers(0, 10)))
        keep = pred & (pvalid if pvalid is not None else True)
        exp_vals = vals[:plen][keep]
        exp_valid = valid[:plen][keep] if v.valid is not None else None
        assert_logical_eq(got, HostArray(A.Int32, exp_vals, exp_valid), f"iter {it}")


# -------------------------------------------------------------------- take
@pytest.mark.parametrize("case", load_golden("take"), ids=lambda c: c["name"])
def test_take_golden(oracle, case):
    v, i = golden_array(case["values"]), golden_array(case["indices"])
    cb = case.get("check_bounds", False)
    if "error" in case or "panic" in case:
        return expect_err(case, lambda: oracle.take(v, i, cb))
    for off in (0, 5):
        got = oracle.take(v, i, cb, bit_offset=off)
        assert_logical_eq(got, golden_array(case["expected"]), case["name"])


def test_take_null_buffer_presence(oracle):
    """take_nulls (take.rs:418-430): values without nulls -> the index nulls are cloned, even
    when that bitmap has no nulls; values with nulls -> None when the result has none."""
    v = HostArray(A.Int32, np.arange(5, dtype=np.int32))
    i = HostArray(A.UInt32, np.array([0, 1, 2], dtype=np.uint32), np.array([True, True, True]))
    got = oracle.take(v, i)
    assert got.valid is not None and got.null_count == 0
    vn = HostArray(A.Int32, np.arange(5, dtype=np.int32), np.array([True, True, True, True, False]))
    got = oracle.take(vn, HostArray(A.UInt32, np.array([0, 1], dtype=np.uint32)))
    assert got.valid is None


def test_take_negative_index_reinterpreted(oracle):
    v = HostArray(A.Int32, np.arange(4, dtype=np.int32))
    i = HostArray(A.Int32, np.array([-1], dtype=np.int32))
    with pytest.raises(A.Panic) as ei:
        oracle.take(v, i)
    assert str(ei.value) == "index out of bounds: the len is 4 but the index is 4294967295"
    with pytest.raises(A.array.ComputeError) as ei:
        oracle.take(v, i, check_bounds=True)
    assert ei.value.message == "Array index out of bounds, cannot get item at index -1 from 4 entries"


# ------------------------------------------------------------------- arith
@pytest.mark.parametrize("case", load_golden("arith"), ids=lambda c: c["name"])
def test_arith_golden(oracle, case):
    l, r = golden_array(case["lhs"]), golden_array(case["rhs"])
    op = ARITH[case["op"]]
    if "error" in case:
        return expect_err(case, lambda: oracle.arith(op, l, r))
    got = oracle.arith(op, l, r)
    exp = golden_array(case["expected"])
    if exp.data_type in (A.Float32, A.Float64):
        assert np.array_equal(np.isnan(got.values), np.isnan(exp.values))
        m = ~np.isnan(exp.values)
        assert np.array_equal(got.values[m], exp.values[m])
    else:
        assert_logical_eq(got, exp, case["name"])


def test_float_div_golden(oracle):
    """test_float div (numeric.rs:1384-1389): [1, 1, <EPS, 4/3, ..., NaN]"""
    a = HostArray(A.Float32, np.array([1.0, 3.4028234663852886e38, 6.0, -4.0, -1.0, 0.0], dtype=np.float32))
    b = HostArray(A.Float32, np.array([1.0, 3.4028234663852886e38, 3.4028234663852886e38, -3.0, 45.0, 0.0], dtype=np.float32))
    r = oracle.arith(6, a, b).values
    assert r[0] == 1.0 and r[1] == 1.0 and r[2] < np.finfo(np.float32).eps
    assert r[3] == np.float32(-4.0) / np.float32(-3.0) and np.isnan(r[5])


def test_arith_scalar_and_null_rules(oracle):
    a = HostArray.from_pylist([1, None, 3], A.Int32)
    s = HostArray.from_pylist([10], A.Int32)
    assert oracle.arith(0, a, s, r_scalar=True).to_pylist() == [11, None, 13]
    assert oracle.arith(2, s, a, l_scalar=True).to_pylist() == [9, None, 7]
    ns = HostArray.from_pylist([None], A.Int32)
    got = oracle.arith(0, a, ns, r_scalar=True)
    assert got.to_pylist() == [None, None, None] and list(got.values) == [0, 0, 0]
    # wrapping evaluates under nulls (arity.rs:127-133); checked leaves 0 (arity.rs:285-294)
    x = HostArray(A.Int32, np.array([5, 7], dtype=np.int32), np.array([True, False]))
    y = HostArray(A.Int32, np.array([1, 2], dtype=np.int32))
    assert list(oracle.arith(1, x, y).values) == [6, 9]
    assert list(oracle.arith(0, x, y).values) == [6, 0]
    # overflow under a null slot is not an error for checked ops
    big = HostArray(A.Int32, np.array([2**31 - 1, 1], dtype=np.int32), np.array([False, True]))
    assert oracle.arith(0, big, y).to_pylist() == [None, 3]


# --------------------------------------------------------------------- cmp
def _cmp_operand(case, side):
    """(HostArray, is_scalar) of one side of a cmp golden: `lhs` / `rhs` arrays or `lhs_scalar` / `rhs_scalar`."""
    if side + "_scalar" in case:
        sc = case[side + "_scalar"]
        return HostArray.from_pylist([sc["value"]], orc.TYPES[sc["type"]]), True
    return golden_array(case[side]), False


@pytest.mark.parametrize("case", load_golden("cmp"), ids=lambda c: c["name"])
def test_cmp_golden(oracle, case):
    op = CMP[case["op"]]
    (l, ls), (r, rs) = _cmp_operand(case, "lhs"), _cmp_operand(case, "rhs")
    if "error" in case:
        return expect_err(case, lambda: oracle.compare(op, l, r, l_scalar=ls, r_scalar=rs))
    exp = golden_array(case["expected"])
    assert_logical_eq(oracle.compare(op, l, r, l_scalar=ls, r_scalar=rs), exp, case["name"])
    if len(exp) and not (ls and rs):  # "larger x10 copy to cover the chunked part" (comparison.rs:146-161, :190-198)
        def x10(h, is_scalar):
            if is_scalar:
                return h
            return HostArray(h.data_type, np.tile(h.values, 10), None if h.valid is None else np.tile(h.valid, 10))
        e10 = HostArray(A.Boolean, np.tile(exp.values, 10), None if exp.valid is None else np.tile(exp.valid, 10))
        assert_logical_eq(oracle.compare(op, x10(l, ls), x10(r, rs), l_scalar=ls, r_scalar=rs), e10, case["name"] + " x10")


def test_cmp_total_order_and_distinct(oracle):
    neg_nan = struct.unpack("<d", struct.pack("<Q", 0xFFF8000000000000))[0]
    vals = np.array([neg_nan, -np.inf, -1.0, -0.0, 0.0, 1.0, np.inf, np.nan])
    a = HostArray(A.Float64, vals[:-1])
    b = HostArray(A.Float64, vals[1:])
    assert oracle.compare(2, a, b).values.all()          # strictly increasing in totalOrder
    assert not oracle.compare(0, HostArray(A.Float64, np.array([-0.0])), HostArray(A.Float64, np.array([0.0]))).values[0]
    x = HostArray.from_pylist([1, None, None, 4], A.Int32)
    y = HostArray.from_pylist([1, None, 3, 5], A.Int32)
    d = oracle.compare(6, x, y)
    assert d.valid is None and d.to_pylist() == [False, False, True, True]
    nd = oracle.compare(7, x, y)
    assert nd.valid is None and nd.to_pylist() == [True, True, False, False]
    ns = HostArray.from_pylist([None], A.Int32)
    assert oracle.compare(2, x, ns, r_scalar=True).to_pylist() == [None] * 4
    assert oracle.compare(6, x, ns, r_scalar=True).to_pylist() == [True, False, False, True]


# -------------------------------------------------------------------- cast
@pytest.mark.parametrize("case", load_golden("cast"), ids=lambda c: c["name"])
def test_cast_golden(oracle, case):
    v = golden_array(case["values"])
    if "error" in case:
        return expect_err(case, lambda: oracle.cast(v, orc.TYPES[case["to"]], safe=case.get("safe", True)))
    got = oracle.cast(v, orc.TYPES[case["to"]], safe=case.get("safe", True))
    assert_logical_eq(got, golden_array(case["expected"]), case["name"])


def test_cast_safe_always_has_null_buffer(oracle):
    v = HostArray(A.Int64, np.array([1, 2, 3], dtype=np.int64))
    got = oracle.cast(v, A.Float64)
    assert got.valid is not None and got.null_count == 0   # primitive_array.rs:1098-1102
    got = oracle.cast(v, A.Float64, safe=False)
    assert got.valid is None
    with pytest.raises(A.array.CastError) as ei:
        oracle.cast(HostArray(A.Float64, np.array([1.0, 256.0])), A.UInt8, safe=False)
    assert ei.value.message == "Can't cast value 256.0 to type UInt8"
    assert oracle.cast(HostArray(A.Float64, np.array([-0.9, 255.9, np.nan, 256.0])), A.UInt8).to_pylist() == [0, 255, None, None]


def _ryu_layout(v):
    """Independent model of ryu's pretty layout on top of CPython repr() digits."""
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "-inf" if v < 0 else "inf"
    sign = "-" if math.copysign(1.0, v) < 0 else ""
    if v == 0:
        return sign + "0.0"
    mant, exp = f"{abs(v):.17e}".split("e")  # placeholder, replaced by repr digits below
    r = repr(abs(v))
    if "e" in r:
        m, e = r.split("e")
        e = int(e)
    else:
        m, e = r, 0
    if "." in m:
        ip, fp = m.split(".")
    else:
        ip, fp = m, ""
    digits = (ip + fp).lstrip("0")
    k = e - len(fp)
    digits_stripped = digits.rstrip("0")
    k += len(digits) - len(digits_stripped)
    digits = digits_stripped
    nd = len(digits)
    kk = nd + k
    if 0 <= k and kk <= 16:
        return sign + digits + "0" * k + ".0"
    if 0 < kk <= 16:
        return sign + digits[:kk] + "." + digits[kk:]
    if -5 < kk <= 0:
        return sign + "0." + "0" * (-kk) + digits
    if nd == 1:
        return f"{sign}{digits}e{kk - 1}"
    return f"{sign}{digits[0]}.{digits[1:]}e{kk - 1}"


def test_format_f64_vs_python_repr(oracle):
    rng = np.random.default_rng(3)
    specials = [0.0, -0.0, 1.0, 1.5, 2.5, 3.2234, 123.564532, -556132.25, 1e15, 1e16, 1e17, 123456789012345678.0,
                0.1, 0.00001, 0.000001, 1e-5, 9.999e-6, 5e-324, 1.7976931348623157e308, 2.2250738585072014e-308,
                9007199254740993.0, 2.0**63, -(2.0**63), 1e21, 1e22, 1e23, float("nan"), float("inf"), float("-inf"),
                123456.0, 1e100, 4.35, 0.3, 299792458.0]
    bits = rng.integers(0, 2**64, 20000, dtype=np.uint64)
    rand = bits.view(np.float64)
    ints = rng.integers(-10**6, 10**6, 2000).astype(np.float64)
    big = rng.integers(-2**63, 2**63 - 1, 2000, dtype=np.int64).astype(np.float64)
    for v in list(specials) + list(rand) + list(ints) + list(big):
        v = float(v)
        assert oracle.format_f64(v) == _ryu_layout(v), repr(v)
    assert oracle.format_f64(1.5) == "1.5" and oracle.format_f64(1e16) == "1e16"
    assert oracle.format_f64(-0.0) == "-0.0" and oracle.format_f64(1e-5) == "0.00001"
    assert oracle.format_f64(1.0) == "1.0" and oracle.format_f64(float("nan")) == "NaN"


# ------------------------------------------------------- pyarrow cross-check
def test_oracle_vs_pyarrow_where_semantics_coincide(oracle):
    """Arrow C++ (pyarrow) is a different implementation; it is only used where its semantics
    match the reference: filter (null mask drops), take (null index -> null), wrapping int add,
    float add.  NOT for float lt or any cast."""
    pa = pytest.importorskip("pyarrow")
    import pyarrow.compute as pc
    rng = np.random.default_rng(11)
    n = 5000
    vals = rng.integers(-2**62, 2**62, n, dtype=np.int64)
    valid = rng.random(n) < 0.9
    mask = rng.random(n) < 0.3
    mvalid = rng.random(n) < 0.95
    pv = pa.array(vals, mask=~valid)
    pm = pa.array(mask, mask=~mvalid)
    got = oracle.filter(HostArray(A.Int64, vals, valid), HostArray(A.Boolean, mask, mvalid))
    assert got.to_pylist() == pc.filter(pv, pm, null_selection_behavior="drop").to_pylist()
    idx = rng.integers(0, n, 3000).astype(np.uint32)
    ivalid = rng.random(3000) < 0.9
    got = oracle.take(HostArray(A.Int64, vals, valid), HostArray(A.UInt32, idx, ivalid))
    assert got.to_pylist() == pc.take(pv, pa.array(idx, mask=~ivalid)).to_pylist()
    b = rng.integers(-2**62, 2**62, n, dtype=np.int64)
    got = oracle.arith(1, HostArray(A.Int64, vals, valid), HostArray(A.Int64, b))
    assert got.to_pylist() == pc.add(pv, pa.array(b)).to_pylist()
    fa, fb = rng.normal(size=n) * 1e6, rng.normal(size=n) * 1e-3
    got = oracle.arith(0, HostArray(A.Float64, fa), HostArray(A.Float64, fb))
    assert np.array_equal(got.values, pc.add(pa.array(fa), pa.array(fb)).to_numpy())


def test_host_generators_are_deterministic(oracle):
    a = oracle.gen_i64(1000, 42, -10**6, 10**6)
    b = oracle.gen_i64(500, 42, -10**6, 10**6, row0=500)
    assert np.array_equal(a[500:], b) and a.min() >= -10**6 and a.max() <= 10**6
    bits = oracle.gen_bits(100000, 43, 0.1)
    assert 0.09 < bits.mean() < 0.11
    f = oracle.gen_f64(1000, 44, -1e6, 1e6)
    assert f.min() >= -1e6 and f.max() < 1e6


# ------------------------------------------------------------------ boolean
BOOL_BIN = {"and": 0, "or": 1, "and_not": 2, "and_kleene": 3, "or_kleene": 4}
BOOL_UN = {"not": 10, "is_null": 11, "is_not_null": 12}


@pytest.mark.parametrize("case", load_golden("boolean"), ids=lambda c: c["name"])
@pytest.mark.parametrize("bit_offset", [0, 5])
def test_boolean_golden(oracle, case, bit_offset):
    l = golden_array(case["lhs"])
    if case["op"] == "nullif":
        got = oracle.nullif(l, golden_array(case["rhs"]), bit_offset)
    elif case["op"] in BOOL_UN:
        got = oracle.boolean_unary(BOOL_UN[case["op"]], l, bit_offset)
    else:
        r = golden_array(case["rhs"])
        if "error" in case:
            return expect_err(case, lambda: oracle.boolean_binary(BOOL_BIN[case["op"]], l, r))
        got = oracle.boolean_binary(BOOL_BIN[case["op"]], l, r, bit_offset)
    assert_logical_eq(got, golden_array(case["expected"]), case["name"])
    if case.get("no_null_buffer"):
        assert got.valid is None


def _rand_strings(rng, n, maxlen=12):
    alphabet = list("abcdefghijklmnopqrstuvwxyz0123456789 é漢")
    return ["".join(rng.choice(alphabet, int(rng.integers(0, maxlen)))) for _ in range(n)]


def test_fuzz_string_filter_take_vs_model(oracle):
    """filter_bytes / take_bytes (filter.rs:890-928, take.rs:499-627) against list comprehensions;
    null slots keep their bytes through filter; take gives null outputs zero-length slots."""
    rng = np.random.default_rng(17)
    for it in range(40):
        n = int(rng.integers(1, 300))
        dt = A.Utf8 if it % 2 else A.LargeUtf8
        vals = _rand_strings(rng, n)
        valid = (rng.random(n) < 0.8) if it % 3 else None
        h = HostArray(dt, vals, valid)
        m = HostArray(A.Boolean, rng.random(n) < rng.random(), (rng.random(n) < 0.9) if it % 4 == 0 else None)
        got = oracle.filter(h, m, bit_offset=it % 5, elem_offset=it % 3)
        keep = m.values & (m.valid if m.valid is not None else True)
        exp_vals = [v for v, k in zip(vals, keep) if k]
        exp_valid = valid[keep] if valid is not None else None
        if exp_valid is not None and exp_valid.all():
            exp_valid = None
        if keep.all():
            exp_valid = valid
        assert_logical_eq(got, HostArray(dt, exp_vals, exp_valid), f"filter {it}")
        assert got.values == exp_vals or got.valid is not None  # bytes of null slots are copied as well
        k = int(rng.integers(0, 200))
        idx = rng.integers(0, n, k).astype(np.uint32)
        ivalid = (rng.random(k) < 0.85) if it % 2 == 0 else None
        got = oracle.take(h, HostArray(A.UInt32, idx, ivalid), elem_offset=it % 2)
        ev = [(valid is None or valid[j]) and (ivalid is None or ivalid[i]) for i, j in enumerate(idx)]
        exp = [vals[j] if ok else None for ok, j in zip(ev, idx)]
        assert got.to_pylist() == exp, f"take {it}"
        if got.valid is not None and k:
            assert all(s == "" for s, ok in zip(got.values, got.valid) if not ok) or got.null_count == 0
    with pytest.raises(A.Panic) as ei:
        oracle.take(HostArray(A.Utf8, ["a", "b"]), HostArray(A.UInt32, np.array([2], dtype=np.uint32)))
    assert str(ei.value) == "index out of bounds: the len is 3 but the index is 3"
    with pytest.raises(A.Panic) as ei:
        oracle.take(HostArray(A.Utf8, ["a", "b"]), HostArray(A.UInt32, np.array([7], dtype=np.uint32)))
    assert str(ei.value) == "index out of bounds: the len is 3 but the index is 7"


# ---------------------------------------------------------------- aggregate
def _same_scalar(got, want, dt):
    """Bitwise for floats (NaN sign matters: total order), plain equality otherwise."""
    if want is None or got is None:
        return got is None and want is None
    if isinstance(want, str):
        want = float(want)
    w = np.array([want]).astype(np.uint8 if dt.physical == A._lib.AH_BOOL else dt.np_dtype)
    if w.dtype.kind == "f" and np.isnan(w[0]):  # golden says "nan" / "-nan": class + sign, not payload
        return bool(np.isnan(got)) and bool(np.signbit(got)) == bool(np.signbit(w[0]))
    return w.tobytes() == np.array([got]).astype(w.dtype).tobytes()


@pytest.mark.parametrize("case", load_golden("aggregate"), ids=lambda c: c["name"])
@pytest.mark.parametrize("bit_offset", [0, 5])
@pytest.mark.parametrize("vector_bytes", [16, 32, 64])
def test_aggregate_golden(oracle, case, bit_offset, vector_bytes):
    v = golden_array(case["values"])
    if "error" in case:
        return expect_err(case, lambda: oracle.aggregate(case["op"], v, vector_bytes, bit_offset))
    got = oracle.aggregate(case["op"], v, vector_bytes, bit_offset)
    assert _same_scalar(got, case["expected"], v.data_type), (got, case["expected"])


# ---------------------------------------------------------------- parquet RowSelection (oracle pinned on the
# reference's own unit tests, transcribed in tests/selection_cases.py)
import selection_cases  # noqa: E402
from selection_model import ModelSelection  # noqa: E402


@pytest.mark.parametrize("case", selection_cases.ALL_CASES, ids=lambda f: f.__name__)
def test_selection_reference_cases_on_the_oracle(oracle, case):
    ModelSelection.oracle = oracle
    case(ModelSelection)


def test_selection_and_then_fuzz_recipe(oracle):
    """algebra.rs:583-612 `test_and_fuzz`, verbatim recipe, oracle vs the naive definition."""
    ModelSelection.oracle = oracle
    rng = np.random.default_rng(0)
    for _ in range(100):
        a_len = int(rng.integers(10, 100))
        a = rng.random(a_len) < 0.2
        b = rng.random(int(a.sum())) < 0.8
        exp = np.zeros(a_len, bool)
        it = iter(b)
        for i, x in enumerate(a):
            if x and next(it):
                exp[i] = True
        got = ModelSelection(a).and_then(ModelSelection(b))
        assert got == ModelSelection(exp) and got.total_row_count() == a_len


# ------------------------------------------------------------------- concat
@pytest.mark.parametrize("case", load_golden("concat"), ids=lambda c: c["name"])
def test_concat_golden(oracle, case):
    if "error" in case:
        return expect_err(case, lambda: oracle.concat([golden_array(p) for p in case["pieces"]]))
    got = oracle.concat([golden_array(p) for p in case["pieces"]])
    assert_logical_eq(got, golden_array(case["expected"]), case["name"])


# ------------------------------------------------------------------- sort_to_indices
@pytest.mark.parametrize("case", load_golden("sort"), ids=lambda c: c["name"])
@pytest.mark.parametrize("bit_offset", [0, 3])
def test_sort_to_indices_golden(oracle, case, bit_offset):
    got = oracle.sort_to_indices(golden_array(case["values"]), case.get("descending", False), case.get("nulls_first", True),
                                 case.get("limit"), bit_offset)
    assert got.valid is None and got.values.tolist() == case["expected"]


# ------------------------------------------------------------------- zip
@pytest.mark.parametrize("case", load_golden("zip"), ids=lambda c: c["name"])
@pytest.mark.parametrize("bit_offset", [0, 5])
def test_zip_golden(oracle, case, bit_offset):
    m, t, f = golden_array(case["mask"]), golden_array(case["truthy"]), golden_array(case["falsy"])
    run = lambda: oracle.zip(m, t, f, case.get("truthy_scalar", False), case.get("falsy_scalar", False), bit_offset)  # noqa: E731
    if "error" in case:
        return expect_err(case, run)
    assert_logical_eq(run(), golden_array(case["expected"]), case["name"])


# ------------------------------------------------------------------- string compare
@pytest.mark.parametrize("case", load_golden("cmp_utf8"), ids=lambda c: c["name"])
@pytest.mark.parametrize("dt", [A.Utf8, A.LargeUtf8], ids=repr)
def test_cmp_utf8_golden(oracle, case, dt):
    l = HostArray(dt, list(case["lhs"]))
    rs = "rhs_scalar" in case
    r = HostArray(dt, [case["rhs_scalar"]] if rs else list(case["rhs"]))
    got = oracle.compare(CMP[case["op"]], l, r, r_scalar=rs)
    assert got.valid is None and got.values.tolist() == case["expected"]
    if not rs:  # x10, like the reference's macro (comparison.rs:146-161)
        got = oracle.compare(CMP[case["op"]], HostArray(dt, list(case["lhs"]) * 10), HostArray(dt, list(case["rhs"]) * 10))
        assert got.values.tolist() == case["expected"] * 10


# ------------------------------------------------------------------- like (arrow-string)
@pytest.mark.parametrize("case", load_golden("like"), ids=lambda c: c["name"])
@pytest.mark.parametrize("dt", [A.Utf8, A.LargeUtf8], ids=str)
def test_like_golden(oracle, case, dt):
    v = HostArray.from_pylist(case["values"], dt)
    got = oracle.string_like(case["op"], v, case["pattern"])
    assert_logical_eq(got, HostArray.from_pylist(case["expected"], A.Boolean), case["name"])
    assert (got.valid is None) == (v.valid is None)  # from_unary clones the input's null buffer


def _py_regex_like(pattern):
    """regex_like (arrow-string/src/predicate.rs:246-306) restated over Python's `re`: an independent regex engine
    for the constructs the reference emits (the oracle has its own small matcher; this pins it)."""
    import re
    out, chars, i = [], list(pattern), 0
    if chars and chars[0] == "%":
        i = 1
    else:
        out.append("^")
    while i < len(chars):
        c = chars[i]
        if c == "\\":
            if i + 1 < len(chars):
                out.append(re.escape(chars[i + 1]))
                i += 1
            else:
                out.append(re.escape("\\"))
        elif c == "%":
            out.append(".*")
        elif c == "_":
            out.append(".")
        else:
            out.append(re.escape(c))
        i += 1
    if out and out[-1] == ".*":
        out.pop()
    else:
        out.append(r"\Z")
    return re.compile("".join(out), re.DOTALL)


def test_like_oracle_agrees_with_a_real_regex_engine(oracle):
    rng = np.random.default_rng(7)
    alphabet = ["a", "b", "%", "_", "\\", ".", "*", "(", "ß", "😈", "\n", "ab"]
    for it in range(400):
        pat = "".join(rng.choice(alphabet, int(rng.integers(0, 7))))
        rows = ["".join(rng.choice(["a", "b", "%", "_", "\\", ".", "*", "(", "ß", "😈", "\n"], int(rng.integers(0, 9)))) for _ in range(40)]
        v = HostArray.from_pylist(rows, A.Utf8)
        rx = _py_regex_like(pat)
        want = [rx.search(r) is not None for r in rows]
        assert oracle.string_like("like", v, pat).to_pylist() == want, (pat, rows)
        assert oracle.string_like("nlike", v, pat).to_pylist() == [not w for w in want], pat
    # the literal-needle forms against Python's own string methods
    rows = ["", "a", "ab", "ba", "aab%", "%_", "ßß", "xß😈"]
    v = HostArray.from_pylist(rows, A.LargeUtf8)
    for needle in ["", "a", "ab", "%", "_", "ß", "😈", "ß😈"]:
        assert oracle.string_like("starts_with", v, needle).to_pylist() == [r.startswith(needle) for r in rows]
        assert oracle.string_like("ends_with", v, needle).to_pylist() == [r.endswith(needle) for r in rows]
        assert oracle.string_like("contains", v, needle).to_pylist() == [needle in r for r in rows]
    nulls = oracle.string_like("like", HostArray.from_pylist(["a", None], A.Utf8), None)
    assert nulls.to_pylist() == [None, None] and nulls.null_count == 2
    assert oracle.string_length(HostArray.from_pylist(["hello", None, "", "ß😈"], A.Utf8)).to_pylist() == [5, None, 0, 6]
    assert oracle.string_length(HostArray.from_pylist(["hello", "ß"], A.LargeUtf8), bits=True).to_pylist() == [40, 16]


# ------------------------------------------------------------------ round 4: float encodings + the named config
@pytest.mark.parametrize("dt", [A.Float32, A.Float64], ids=str)
def test_oracle_float_arith_on_every_encoding_vs_numpy(oracle, dt):
    """The oracle's float arithmetic (ArrowNativeTypeOp, arrow-array/src/arithmetic.rs:308-430) on operands drawn over
    the whole encoding space (tests/float_strata.py) against an independent implementation — numpy's IEEE add / sub /
    mul / div and C fmod on the same host: bit-exact wherever the result is not a NaN (NaN sign / payload rules are
    the x86 host's and are pinned separately, test_generated_nan_bits...).  This pins the CHECKER for the GPU test
    test_fuzz_arith_float_bit_patterns, which then compares the device with it."""
    import zlib
    from float_strata import float_strata
    rng = np.random.default_rng(zlib.crc32(dt.name.encode()))
    a, b = float_strata(rng, dt, 20000)
    ha, hb = HostArray(dt, a), HostArray(dt, b)
    ut = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
    with np.errstate(all="ignore"):
        ref = {0: a + b, 1: a + b, 2: a - b, 3: a - b, 4: a * b, 5: a * b, 6: a / b, 7: np.fmod(a, b)}
    for op, want in ref.items():
        got = np.asarray(oracle.arith(op, ha, hb).values)
        ok = ~np.isnan(want)
        assert np.array_equal(np.isnan(got), np.isnan(want)), f"{dt} op {op}: NaN positions"
        bad = np.nonzero(got[ok].view(ut) != want[ok].view(ut))[0]
        assert len(bad) == 0, f"{dt} op {op}: {len(bad)} results differ from numpy, first at {bad[:4]}"
    # the strata really reach the regimes they are named for
    tiny = np.finfo(a.dtype).tiny
    with np.errstate(all="ignore"):
        assert (np.abs(a) < tiny).sum() > 1000 and np.isinf(a * b).sum() > 1000
        assert ((np.abs(a / b) < tiny) & (a / b != 0)).sum() > 500          # quotients that are subnormal
        gap = np.abs(np.frexp(a)[1].astype(np.int64) - np.frexp(b)[1].astype(np.int64))
        assert (gap[np.isfinite(a) & np.isfinite(b) & (a != 0) & (b != 0)] > (200 if dt == A.Float32 else 1500)).sum() > 1000


@pytest.mark.parametrize("dt", [A.Float32, A.Float64], ids=str)
def test_oracle_float_compare_is_total_order_on_every_encoding(oracle, dt):
    """lt / eq of the oracle == IEEE totalOrder on the sign-magnitude key of the raw bits (f64::total_cmp,
    arithmetic.rs:400-410), checked with an independent integer key over every encoding class."""
    import zlib
    from float_strata import float_strata
    rng = np.random.default_rng(zlib.crc32(dt.name.encode()) + 1)
    a, b = float_strata(rng, dt, 20000)
    it, bits = ((np.int32, 31) if dt == A.Float32 else (np.int64, 63))

    def key(x):  # total_cmp: flip the magnitude bits of negative values
        i = x.view(it).copy()
        return i ^ ((i >> bits) & np.iinfo(it).max)
    ka, kb = key(a), key(b)
    ha, hb = HostArray(dt, a), HostArray(dt, b)
    for op, want in ((0, ka == kb), (1, ka != kb), (2, ka < kb), (3, ka <= kb), (4, ka > kb), (5, ka >= kb)):
        assert np.array_equal(np.asarray(oracle.compare(op, ha, hb).values, dtype=bool), want), f"{dt} compare op {op}"


INT_TYPES = [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64]


@pytest.mark.parametrize("dt", INT_TYPES, ids=str)
def test_oracle_integer_arith_full_range_vs_bigint_model(oracle, dt):
    """Pins the CHECKER of test_fuzz_arith_full_range_integers: the oracle's integer ops (ArrowNativeTypeOp for integers,
    arrow-array/src/arithmetic.rs:147-260; numeric.rs:345-351 for `MIN % -1`) over full-range operands of every width
    against an independent model in Python's unbounded integers: wrapping = reduce modulo 2^bits; checked = error iff
    the exact result leaves the type, reporting the FIRST failing valid row with the reference's text; division truncates
    toward zero, zero divisor = DivideByZero, `MIN / -1` overflows, `MIN % -1` = 0."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(("intmodel-" + dt.name).encode()))
    info = np.iinfo(dt.np_dtype)
    bits, lo, hi = info.bits, int(info.min), int(info.max)
    sym = {0: "+", 2: "-", 4: "*", 6: "/", 7: "%"}

    def wrap(v):
        v &= (1 << bits) - 1
        return v - (1 << bits) if lo < 0 and v >= (1 << (bits - 1)) else v

    def tdiv(x, y):
        q = abs(x) // abs(y)
        return q if (x < 0) == (y < 0) else -q

    for it in range(6):
        n = int(rng.integers(1, 400))
        a = rng.integers(lo, hi, n, dtype=dt.np_dtype, endpoint=True)
        b = rng.integers(lo, hi, n, dtype=dt.np_dtype, endpoint=True)
        edges = np.array([lo, hi, 0, 1] + ([-1] if lo < 0 else []), dtype=dt.np_dtype)
        for x in (a, b):
            k = rng.random(n) < 0.1
            x[k] = rng.choice(edges, int(k.sum()))
        if it >= 3:
            b = rng.integers(1, 3, n).astype(dt.np_dtype)  # the checked forms succeed now and then
        valid = (rng.random(n) < 0.8) if it % 2 else None
        ha, hb = HostArray(dt, a, valid), HostArray(dt, b)
        A_, B_ = [int(v) for v in a], [int(v) for v in b]
        ok_rows = range(n) if valid is None else [i for i in range(n) if valid[i]]
        for op in range(8):
            exact = {0: lambda x, y: x + y, 1: lambda x, y: x + y, 2: lambda x, y: x - y, 3: lambda x, y: x - y,
                     4: lambda x, y: x * y, 5: lambda x, y: x * y}.get(op)
            err = None
            want = [0] * n
            if op in (1, 3, 5):  # wrapping: every slot (arity::binary applies the op to null slots too)
                want = [wrap(exact(x, y)) for x, y in zip(A_, B_)]
            else:
                for i in ok_rows:
                    x, y = A_[i], B_[i]
                    if op in (6, 7) and y == 0:
                        err = (A.array.DivideByZero, "Divide by zero error")
                        break
                    if op == 7:
                        r = 0 if y == -1 else x - tdiv(x, y) * y
                    else:
                        r = tdiv(x, y) if op == 6 else exact(x, y)
                    if not lo <= r <= hi:
                        err = (A.array.ArithmeticOverflow, f"Overflow happened on: {x} {sym[op]} {y}")
                        break
                    want[i] = r
            if err:
                with pytest.raises(err[0]) as ei:
                    oracle.arith(op, ha, hb)
                assert ei.value.message == err[1], f"{dt} op {op} iter {it}"
                continue
            got = oracle.arith(op, ha, hb)
            gv = [int(v) for v in np.asarray(got.values)]
            rows = range(n) if op in (1, 3, 5) else ok_rows
            assert all(gv[i] == want[i] for i in rows), f"{dt} op {op} iter {it}"
            if op not in (1, 3, 5) and valid is not None:  # try_binary leaves 0 under null slots (arity.rs:285-294)
                assert all(gv[i] == 0 for i in range(n) if not valid[i]), f"{dt} op {op} iter {it}: null slots"


def f16_strata(rng, n):
    """Float16 operand pairs over the whole encoding space: uniform bit patterns (every exponent, subnormals, NaN payloads of
    both signs), plus planted zeros / infinities / the largest and smallest magnitudes / ties for the 11-bit significand."""
    a = rng.integers(0, 1 << 16, n, dtype=np.uint16)
    b = rng.integers(0, 1 << 16, n, dtype=np.uint16)
    special = np.array([0x0000, 0x8000, 0x7C00, 0xFC00, 0x7BFF, 0xFBFF, 0x0001, 0x8001, 0x03FF, 0x0400, 0x3C00, 0xBC00, 0x7E00, 0xFE00,
                        0x7C01, 0xFDFF, 0x3555, 0x4248], dtype=np.uint16)
    for x in (a, b):
        k = rng.random(n) < 0.15
        x[k] = rng.choice(special, int(k.sum()))
    k = rng.random(n) < 0.1  # near-equal magnitudes: cancellation, quotients near 1
    b[k] = a[k] ^ rng.integers(0, 4, int(k.sum())).astype(np.uint16)
    return a.view(np.float16), b.view(np.float16)


def test_oracle_float16_conversions_on_every_encoding(oracle):
    """The oracle's restated `half` 2.7.1 conversions against numpy's IEEE binary16: all 65 536 encodings through
    Float16 -> Float32 / Float64 / every integer type and `neg`, and Float32 / Float64 / Int64 -> Float16 on every f16 value,
    both of its f32 neighbours and the exact midpoints between adjacent f16 values (the ties of round-to-nearest-even)."""
    h = np.arange(1 << 16, dtype=np.uint32).astype(np.uint16)
    hv = h.view(np.float16)
    hh = HostArray(A.Float16, hv)
    nan = np.isnan(hv)
    # f16 -> f32 / f64: exact; NaN: payload << 13 with the quiet bit set (f16_to_f32_fallback)
    for dt, ut in ((A.Float32, np.uint32), (A.Float64, np.uint64)):
        got = np.asarray(oracle.cast(hh, dt).values)
        want = hv.astype(dt.np_dtype)
        assert np.array_equal(got[~nan].view(ut), want[~nan].view(ut)), dt
        assert np.isnan(got[nan]).all()
    g32 = np.asarray(oracle.cast(hh, A.Float32).values).view(np.uint32)
    want_nan = ((h[nan].astype(np.uint32) & 0x8000) << 16) | 0x7FC00000 | ((h[nan].astype(np.uint32) & 0x3FF) << 13)
    assert np.array_equal(g32[nan], want_nan)
    # neg: bit 15 flips, NaNs included
    assert np.array_equal(np.asarray(oracle.neg(hh).values).view(np.uint16), h ^ 0x8000)
    assert np.array_equal(np.asarray(oracle.neg(hh, wrapping=True).values).view(np.uint16), h ^ 0x8000)
    # f16 -> integers (safe): via f32; valid iff trunc(v) is representable, NaN / inf -> null
    f = hv.astype(np.float64)
    for dt in (A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64):
        info = np.iinfo(dt.np_dtype)
        out = oracle.cast(hh, dt)
        with np.errstate(invalid="ignore"):
            t = np.trunc(f)
        ok = np.isfinite(f) & (t >= info.min) & (t <= info.max)
        assert np.array_equal(np.asarray(out.valid, dtype=bool), ok), dt
        assert np.array_equal(np.asarray(out.values)[ok], t[ok].astype(dt.np_dtype)), dt
    # f32 -> f16: every f16 value, its two f32 neighbours, and the midpoint to the next f16 (ties to even)
    fin = hv[np.isfinite(hv)].astype(np.float32)
    with np.errstate(over="ignore"):
        nxt = np.nextafter(hv[np.isfinite(hv)], np.float16(np.inf)).astype(np.float32)
    mid = ((fin.astype(np.float64) + nxt.astype(np.float64)) / 2).astype(np.float32)  # exact: 12 significant bits
    cand = np.concatenate([fin, np.nextafter(fin, np.float32(np.inf)), np.nextafter(fin, np.float32(-np.inf)), mid,
                           np.nextafter(mid, np.float32(np.inf)), np.nextafter(mid, np.float32(-np.inf)),
                           np.array([65504, 65519.99, 65520, 65536, 1e9, np.inf, -np.inf, 2.0**-24, 2.0**-25, 2.0**-25 * 1.0001, 2.0**-26,
                                     -2.0**-25, 1e-30, 0.0, -0.0], dtype=np.float32)])
    with np.errstate(over="ignore"):
        want = cand.astype(np.float16).view(np.uint16)
    got = np.asarray(oracle.cast(HostArray(A.Float32, cand), A.Float16).values).view(np.uint16)
    assert np.array_equal(got, want)
    # f32 NaNs: sign | 0x7C00 | 0x0200 | (mantissa >> 13)   (f32_to_f16_fallback)
    nb = np.array([0x7FC00000, 0xFFC00000, 0x7F800001, 0xFF800001, 0x7FA00000, 0x7FFFFFFF, 0x7F802000], dtype=np.uint32)
    got = np.asarray(oracle.cast(HostArray(A.Float32, nb.view(np.float32)), A.Float16).values).view(np.uint16)
    assert np.array_equal(got, ((nb >> 16) & 0x8000) | 0x7C00 | 0x0200 | ((nb & 0x7FFFFF) >> 13))
    # f64 -> f16 and i64 -> f16 go THROUGH f32 (NumCast for f16 = n.to_f32().map(f16::from_f32)): double rounding is observable
    d = np.concatenate([cand.astype(np.float64), np.array([1.0 + 2.0**-11 + 2.0**-30, 2049.0000001, 65519.999999], dtype=np.float64)])
    with np.errstate(over="ignore"):
        want = d.astype(np.float32).astype(np.float16).view(np.uint16)
    assert np.array_equal(np.asarray(oracle.cast(HostArray(A.Float64, d), A.Float16).values).view(np.uint16), want)
    assert np.float64(1.0 + 2.0**-11 + 2.0**-30).astype(np.float16) != np.float64(1.0 + 2.0**-11 + 2.0**-30).astype(np.float32).astype(np.float16)
    rng = np.random.default_rng(16)
    iv = np.concatenate([rng.integers(-70000, 70000, 4000), rng.integers(-2**63, 2**63 - 1, 2000), np.array([2049, 2051, 65519, 65520, -65520, 16777217])]).astype(np.int64)
    with np.errstate(over="ignore"):
        want = iv.astype(np.float32).astype(np.float16).view(np.uint16)
    assert np.array_equal(np.asarray(oracle.cast(HostArray(A.Int64, iv), A.Float16).values).view(np.uint16), want)


def test_oracle_float16_arith_and_compare_vs_numpy(oracle):
    """Float16 add / sub / mul / div / rem = to f32, ONE f32 operation, round back (half 2.7.1 `impl Add for f16` etc.; numpy's
    float16 arithmetic is defined the same way), on a stratified sample of encoding pairs: bit-exact wherever the result is
    not a NaN.  Compare = totalOrder on the 16-bit pattern, equality = bit equality."""
    rng = np.random.default_rng(1616)
    a, b = f16_strata(rng, 60000)
    ha, hb = HostArray(A.Float16, a), HostArray(A.Float16, b)
    a32, b32 = a.astype(np.float32), b.astype(np.float32)
    with np.errstate(all="ignore"):
        ref = {0: a32 + b32, 1: a32 + b32, 2: a32 - b32, 3: a32 - b32, 4: a32 * b32, 5: a32 * b32, 6: a32 / b32, 7: np.fmod(a32, b32)}
        ref = {k: v.astype(np.float16) for k, v in ref.items()}
    for op, want in ref.items():
        got = np.asarray(oracle.arith(op, ha, hb).values)
        ok = ~np.isnan(want)
        assert np.array_equal(np.isnan(got), np.isnan(want)), f"f16 op {op}: NaN positions"
        bad = np.nonzero(got[ok].view(np.uint16) != want[ok].view(np.uint16))[0]
        assert len(bad) == 0, f"f16 op {op}: {len(bad)} results differ from numpy, first at {bad[:4]}"
    # generated NaN is the x86 default NaN through f32: 0xFFC00000 -> 0xFE00
    inf = HostArray(A.Float16, np.array([np.inf], dtype=np.float16))
    ninf = HostArray(A.Float16, np.array([-np.inf], dtype=np.float16))
    assert np.asarray(oracle.arith(0, inf, ninf).values).view(np.uint16)[0] == 0xFE00

    def key(x):
        i = x.view(np.int16).copy()
        return i ^ ((i >> 15) & 0x7FFF)
    ka, kb = key(a), key(b)
    for op, want in ((0, ka == kb), (1, ka != kb), (2, ka < kb), (3, ka <= kb), (4, ka > kb), (5, ka >= kb)):
        assert np.array_equal(np.asarray(oracle.compare(op, ha, hb).values, dtype=bool), want), f"f16 compare op {op}"
    # scalar forms and nulls ride the same generic paths as f32 (arith_typed / compare_op)
    sc = HostArray(A.Float16, b[:1].copy())
    got = np.asarray(oracle.arith(4, ha, sc, r_scalar=True).values)
    with np.errstate(all="ignore"):
        want = (a32 * b32[0]).astype(np.float16)
    ok = ~np.isnan(want)
    assert np.array_equal(got[ok].view(np.uint16), want[ok].view(np.uint16))


def test_config0_on_the_oracle():
    """BASELINE.json configs[0] on the CPU reference path: filter() on a 2^20-row Int32 PrimitiveArray, 50 % selected,
    no nulls (arrow/benches/filter_kernels.rs:39-45 at BASELINE's size).  default_strategy picks IndexIterator
    (filter.rs:346-364: selectivity <= 0.8); the result is numpy's boolean indexing."""
    import os
    o = orc.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so"))
    n = 1 << 20
    rng = np.random.default_rng(20)
    vals = HostArray(A.Int32, rng.integers(-2**31, 2**31 - 1, n, dtype=np.int32))
    mask = HostArray(A.Boolean, rng.random(n) < 0.5)
    assert o.filter_strategy(mask) == "Indices"
    out = o.filter(vals, mask)
    assert out.valid is None and np.array_equal(out.values, vals.values[mask.values])
    assert o.filter_strategy(HostArray(A.Boolean, rng.random(n) < 0.9)) == "Slices"
    assert o.filter_strategy(HostArray(A.Boolean, np.ones(8, dtype=bool))) == "All"
    assert o.filter_strategy(HostArray(A.Boolean, np.zeros(8, dtype=bool))) == "None"


@pytest.mark.parametrize("case", __import__("coalesce_reference_cases").SCENARIOS, ids=lambda c: c[0])
def test_coalescer_model_on_the_reference_scenarios(oracle, case):
    """tests/coalesce_model.py — the Python restatement of BatchCoalescer the GPU fuzz tests compare the device coalescer with —
    on every deterministic scenario of the reference's own tests (arrow-select/src/coalesce.rs mod tests: output batch sizes,
    buffered rows, the large-batch bypass rules): the model is pinned to the reference's numbers, not only to itself."""
    import numpy as np
    import arrow_rs_amd as A
    from coalesce_model import ModelCoalescer
    from coalesce_reference_cases import run_scenario
    from orc import HostArray
    name, source, target, limit, steps, tail = case
    m = ModelCoalescer(oracle, [A.Int32], target)
    m.limit = limit

    class Adapter:
        def push(self, n):
            m.push([HostArray(A.Int32, np.arange(n, dtype=np.int32))])

        def drain(self):
            out = []
            while m.completed:
                out.append([int(x) for x in m.completed.popleft()[0].values])
            return out

        def buffered(self):
            return m.buffered

        def has_completed(self):
            return bool(m.completed)

        def finish(self):
            m.finish()

    run_scenario(Adapter(), steps, tail)


def test_neg_reference_vectors_on_the_oracle(oracle):
    """test_neg (arrow-arith/src/numeric.rs:1151-1187) on the oracle: the inline vectors, the overflow texts, neg_wrapping at MIN,
    unsigned refused by neg and wrapped by neg_wrapping (:101-103, :181)."""
    import arrow_rs_amd as A
    from orc import HostArray
    src, out = [1, -5, 2, 693, 3929], [-1, 5, -2, -693, -3929]
    for dt in (A.Int32, A.Int64):
        assert oracle.neg(HostArray(dt, np.array(src, dtype=dt.np_dtype))).to_pylist() == out
        lo = np.iinfo(dt.np_dtype).min
        with pytest.raises(Exception) as ei:
            oracle.neg(HostArray(dt, np.array([lo], dtype=dt.np_dtype)))
        assert f"Arithmetic overflow: Overflow happened on: - {lo}" in str(ei.value)
        assert oracle.neg(HostArray(dt, np.array([lo], dtype=dt.np_dtype)), wrapping=True).to_pylist() == [lo]
    f = np.array([np.finfo(np.float32).max, np.finfo(np.float32).min, np.inf, 1.3, 0.5], dtype=np.float32)
    assert np.array_equal(oracle.neg(HostArray(A.Float32, f)).values.view(np.uint32), (-f).view(np.uint32))
    with pytest.raises(Exception) as ei:
        oracle.neg(HostArray(A.UInt8, np.array([1, 2], dtype=np.uint8)))
    assert "Invalid arithmetic operation: !UInt8" in str(ei.value)
    assert oracle.neg(HostArray(A.UInt8, np.array([1, 2], dtype=np.uint8)), wrapping=True).to_pylist() == [255, 254]


def test_count_set_bits_reference_vectors(oracle):
    """Buffer::count_set_bits_offset (arrow-buffer/src/buffer/immutable.rs:807-890) on the oracle: tests/count_bits_cases.py."""
    import ctypes as C
    from count_bits_cases import CASES
    for data, byte_off, bit_off, nbits, expected in CASES:
        buf = np.array(list(data) + [0] * 16, dtype=np.uint8)  # (padding: the oracle reads whole words)
        got = oracle.lib.orc_count_set_bits(buf.ctypes.data + byte_off, bit_off, nbits)
        assert got == expected, (data, byte_off, bit_off, nbits, got, expected)


def test_float_total_order_min_max_on_the_oracle(oracle):
    """test_float_total_order_min_max (arrow-array/src/arithmetic.rs:863-897): MIN_TOTAL_ORDER (all bits set) is below -inf and
    below -NaN, MAX_TOTAL_ORDER (all bits but the sign) above +inf and above NaN, for f64 / f32 / f16 — through lt / gt."""
    import arrow_rs_amd as A
    from orc import HostArray

    cases = []
    for dt, ut, nbits in ((A.Float64, np.uint64, 64), (A.Float32, np.uint32, 32), (A.Float16, np.uint16, 16)):
        fdt = np.dtype(dt.np_dtype)
        lo = np.array([(1 << nbits) - 1], dtype=ut).view(fdt)            # MIN_TOTAL_ORDER: every bit set (a negative NaN)
        hi = np.array([(1 << (nbits - 1)) - 1], dtype=ut).view(fdt)      # MAX_TOTAL_ORDER: every bit but the sign (a positive NaN)
        ninf, pinf = np.array([-np.inf], dtype=fdt), np.array([np.inf], dtype=fdt)
        nan = np.array([np.nan], dtype=fdt)
        neg_nan = (nan.view(ut) | ut(1 << (nbits - 1))).view(fdt)
        cases += [(dt, "lt", lo, ninf), (dt, "lt", lo, neg_nan), (dt, "gt", hi, pinf), (dt, "gt", hi, nan)]
        assert np.isnan(lo.astype(np.float64))[0] and np.isnan(hi.astype(np.float64))[0]
    for dt, op, a, b in cases:
        r = oracle.compare({"lt": 2, "gt": 4}[op], HostArray(dt, a), HostArray(dt, b))
        assert r.to_pylist() == [True], (dt, op)
