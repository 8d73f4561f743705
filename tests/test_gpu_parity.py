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


def check_float(got_dev, exp_host, msg="", lhs=None, rhs=None):
    """Float arithmetic: plain BIT equality (0 ULP; stated bar <= 1 ULP), NaNs included: the device reproduces the
    x86 host's NaN bits — generated NaNs are the x86 default 0xFFF8... / 0xFFC0..., a NaN operand is propagated
    quieted (arith.hip).  The one slot class left out of the bit comparison is "BOTH operands NaN": SSE returns
    the first source operand of the instruction, and which operand a compiler puts first for a commutative op is
    its register allocator's choice (the reference's own result is unpinned there); the result must still be NaN."""
    g = host(got_dev)
    assert_same_nulls_presence(g, exp_host, msg)
    gv, ev = np.asarray(g.values), np.asarray(exp_host.values)
    assert len(gv) == len(ev)
    m = np.ones(len(gv), dtype=bool)
    if lhs is not None and rhs is not None:
        both = np.isnan(np.asarray(lhs)) & np.isnan(np.asarray(rhs))
        assert np.all(np.isnan(gv[both]) & np.isnan(ev[both])), f"{msg} NaN op NaN must be NaN"
        m = ~both
    ut = {4: np.uint32, 8: np.uint64}[gv.dtype.itemsize]
    bad = np.nonzero(gv[m].view(ut) != ev[m].view(ut))[0]
    assert len(bad) == 0, (f"{msg} float bits differ at {bad[:5]}: got {gv[m].view(ut)[bad[:5]]} "
                           f"expected {ev[m].view(ut)[bad[:5]]}")
    if g.valid is not None:
        assert np.array_equal(g.valid, exp_host.valid)


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
    if exp.data_type in (A.Float32, A.Float64):
        gl, gr = golden_array(case["lhs"]), golden_array(case["rhs"])
        same_len = not isinstance(gl.values, list) and not isinstance(gr.values, list) and len(gl) == len(gr)
        check_float(fn(l, r), exp, case["name"], gl.values if same_len else None, gr.values if same_len else None)
    else:
        check_exact(fn(l, r), exp, case["name"])


def test_count_set_bits_reference_vectors(ctx):
    """Buffer::count_set_bits_offset's inline vectors (arrow-buffer/src/buffer/immutable.rs:807-890; tests/count_bits_cases.py)
    through ah_count_set_bits: byte-offset slices, bit offsets, zero lengths — buffers uploaded at their exact size."""
    import ctypes as C
    from count_bits_cases import CASES
    for data, byte_off, bit_off, nbits, expected in CASES:
        buf = A.array.DeviceBuffer.from_numpy(ctx, np.array(data, dtype=np.uint8))
        cnt = C.c_int64(-1)
        ctx.check(ctx.lib.ah_count_set_bits(ctx.handle, C.c_void_p(buf.ptr + byte_off), bit_off, nbits, C.byref(cnt)))
        assert cnt.value == expected, (data, byte_off, bit_off, nbits, cnt.value, expected)


def test_bitmap_set_bits_reference_vectors(ctx):
    """bit_mask::set_bits' inline vectors (arrow-buffer/src/util/bit_mask.rs:183-263: aligned, unaligned destination start,
    unaligned destination end, both unaligned) through ah_bitmap_set_bits — the validity merge of concat and of the multi-GPU
    reassembly.  The reference returns the NULL count of the copied range; the entry point returns the set bits (= len - nulls)."""
    import ctypes as C
    d8 = [0b11100111, 0b10100101, 0b10011001, 0b11011011, 0b11101011, 0b11000011, 0b11100111, 0b10100101]
    cases = [
        (10, d8, 8, 0, 64, [0, 0b11100111, 0b10100101, 0b10011001, 0b11011011, 0b11101011, 0b11000011, 0b11100111, 0b10100101, 0], 24),
        (10, d8, 3, 0, 64, [0b00111000, 0b00101111, 0b11001101, 0b11011100, 0b01011110, 0b00011111, 0b00111110, 0b00101111, 0b00000101, 0], 24),
        (10, d8, 8, 0, 62, [0, 0b11100111, 0b10100101, 0b10011001, 0b11011011, 0b11101011, 0b11000011, 0b11100111, 0b00100101, 0], 23),
        (13, d8[:6] * 3, 3, 5, 95, [0b01111000, 0b01101001, 0b11100110, 0b11110110, 0b11111010, 0b11110000, 0b01111001, 0b01101001,
                                        0b11100110, 0b11110110, 0b11111010, 0b11110000, 0b00000001], 35),
    ]
    for nbytes, data, off_w, off_r, n, expected, nulls in cases:
        dst = A.array.DeviceBuffer.from_numpy(ctx, np.zeros(nbytes, dtype=np.uint8))
        src = A.array.DeviceBuffer.from_numpy(ctx, np.array(data, dtype=np.uint8))
        got = C.c_int64(-1)
        ctx.check(ctx.lib.ah_bitmap_set_bits(ctx.handle, C.c_void_p(dst.ptr), off_w, C.c_void_p(src.ptr), off_r, n, C.byref(got)))
        assert dst.to_numpy().tolist() == expected, (off_w, off_r, n)
        assert got.value == n - nulls, (got.value, n - nulls)


def test_float_total_order_min_max(ctx):
    """test_float_total_order_min_max (arrow-array/src/arithmetic.rs:863-897) through lt / gt on the device: the all-ones NaN is
    below -inf and below -NaN, the all-but-sign NaN above +inf and above NaN, for Float64 / Float32 / Float16."""

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
        fn = K.lt if op == "lt" else K.gt
        assert fn(HostArray(dt, a).to_device(ctx), HostArray(dt, b).to_device(ctx)).to_pylist() == [True], (dt, op)
        assert fn(HostArray(dt, b).to_device(ctx), HostArray(dt, a).to_device(ctx)).to_pylist() == [False], (dt, op)


def test_neg_reference_vectors(ctx):
    """test_neg (arrow-arith/src/numeric.rs:1151-1198), the inline vectors: neg over Int32 / Int64 / the four Duration units /
    Float32, the overflow texts of i32::MIN / i64::MIN / Duration(i64::MIN), neg_wrapping keeping MIN for the plain integers
    and — `downcast_integer!` does not match a Duration — raising the same overflow for a Duration column."""
    src, out = [1, -5, 2, 693, 3929], [-1, 5, -2, -693, -3929]
    for dt in (A.Int32, A.Int64, A.DurationSecond, A.DurationMillisecond, A.DurationMicrosecond, A.DurationNanosecond):
        r = K.neg(HostArray(dt, np.array(src, dtype=dt.np_dtype)).to_device(ctx))
        assert r.data_type == dt and r.to_pylist() == out, dt
    f = np.array([np.finfo(np.float32).max, np.finfo(np.float32).min, np.inf, 1.3, 0.5], dtype=np.float32)
    r = K.neg(HostArray(A.Float32, f).to_device(ctx))
    assert np.array_equal(host(r).values.view(np.uint32), (-f).view(np.uint32))
    for dt, text in ((A.Int32, "Arithmetic overflow: Overflow happened on: - -2147483648"),
                     (A.Int64, "Arithmetic overflow: Overflow happened on: - -9223372036854775808"),
                     (A.DurationSecond, "Arithmetic overflow: Overflow happened on: - -9223372036854775808")):
        lo = np.iinfo(dt.np_dtype).min
        with pytest.raises(A.array.ArithmeticOverflow) as ei:
            K.neg(HostArray(dt, np.array([lo], dtype=dt.np_dtype)).to_device(ctx))
        assert str(ei.value) == text, str(ei.value)
    for dt in (A.Int32, A.Int64):
        lo = np.iinfo(dt.np_dtype).min
        assert K.neg_wrapping(HostArray(dt, np.array([lo], dtype=dt.np_dtype)).to_device(ctx)).to_pylist() == [lo]
    with pytest.raises(A.array.ArithmeticOverflow) as ei:
        K.neg_wrapping(HostArray(A.DurationSecond, np.array([-2**63], dtype=np.int64)).to_device(ctx))
    assert "Arithmetic overflow: Overflow happened on: - -9223372036854775808" in str(ei.value)
    # neg of an unsigned column is refused, neg_wrapping wraps it (numeric.rs:101-103, :181)
    with pytest.raises(A.array.InvalidArgumentError) as ei:
        K.neg(HostArray(A.UInt8, np.array([1, 2], dtype=np.uint8)).to_device(ctx))
    assert "Invalid arithmetic operation: !UInt8" in str(ei.value)
    assert K.neg_wrapping(HostArray(A.UInt8, np.array([1, 2], dtype=np.uint8)).to_device(ctx)).to_pylist() == [255, 254]


@pytest.mark.parametrize("case", load_golden("cmp"), ids=lambda c: c["name"])
def test_cmp_golden(ctx, case):
    fn = CMP_FN[CMP[case["op"]]]

    def operand(side):
        if side + "_scalar" in case:
            sc = case[side + "_scalar"]
            return A.Scalar.new(sc["value"], orc.TYPES[sc["type"]], ctx)
        return dev(case[side], ctx)

    l, r = operand("lhs"), operand("rhs")
    if "error" in case:
        return expect_err(case, lambda: fn(l, r))
    exp = golden_array(case["expected"])
    check(fn(l, r), exp, case["name"])
    if len(exp) and not (isinstance(l, A.Scalar) and isinstance(r, A.Scalar)):  # the reference's x10 replication
        def x10(side, d):
            if isinstance(d, A.Scalar):
                return d
            h = golden_array(case[side])
            return HostArray(h.data_type, np.tile(h.values, 10), None if h.valid is None else np.tile(h.valid, 10)).to_device(ctx)
        e10 = HostArray(A.Boolean, np.tile(exp.values, 10), None if exp.valid is None else np.tile(exp.valid, 10))
        check(fn(x10("lhs", l), x10("rhs", r)), e10, case["name"] + " x10")


@pytest.mark.parametrize("case", load_golden("cast"), ids=lambda c: c["name"])
def test_cast_golden(ctx, case):
    opts = K.CastOptions(safe=case.get("safe", True))
    if "error" in case:
        return expect_err(case, lambda: K.cast_with_options(dev(case["values"], ctx), orc.TYPES[case["to"]], opts))
    got = K.cast_with_options(dev(case["values"], ctx), orc.TYPES[case["to"]], opts)
    check(got, golden_array(case["expected"]), case["name"])
    if case["to"] == "Utf8":
        check(K.cast(dev(case["values"], ctx), A.LargeUtf8),
              HostArray(A.LargeUtf8, golden_array(case["expected"]).values, golden_array(case["expected"]).valid))


# ------------------------------------------------------------- fuzz: filter
def _seed(name):
    """A seed from a type name that is the SAME in every process (hash(str) is salted per interpreter run, so a red
    run could not be replayed — VERDICT r03 weak #2)."""
    import zlib
    return zlib.crc32(name.encode())


from float_strata import float_strata as _float_strata  # noqa: E402


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
    rng = np.random.default_rng(_seed(dt.name))
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


@pytest.mark.parametrize("n", [1, 777, 8192, 100003])
def test_filter_record_batch_many_columns_one_launch(ctx, oracle, n):
    """filter_record_batch (filter.rs:201-221 applied per column, :476 row count): primitive columns of one width and
    validity shape leave through ONE scatter launch and one host wait (groups of up to 8), everything else column by
    column; each output column — values, null-buffer presence, null count — must equal the oracle's filter()."""
    rng = np.random.default_rng(31 + n)
    nulls = lambda p: rng.random(n) < p
    spec = [(A.Int64, nulls(0.9)), (A.Float64, nulls(0.8)), (A.Int64, None), (A.UInt64, nulls(0.5)), (A.Int32, nulls(0.9)),
            (A.Float32, nulls(0.9)), (A.Int32, None), (A.UInt32, None), (A.Boolean, nulls(0.9)), (A.Utf8, nulls(0.9)),
            (A.Int16, nulls(0.7)), (A.UInt16, nulls(0.7)), (A.Int8, None)] + [(A.Int64, nulls(0.95)) for _ in range(9)]
    cols = []
    for dt, v in spec:
        if dt is A.Utf8:
            cols.append(HostArray.from_pylist([None if (v is not None and not v[i]) else f"s{i % 97}" * (i % 4) for i in range(n)], A.Utf8))
        elif dt is A.Boolean:
            cols.append(HostArray(A.Boolean, rng.random(n) < 0.5, v))
        else:
            cols.append(HostArray(dt, _rand_values(rng, dt, n), v))
    names = [f"c{i}" for i in range(len(cols))]
    for sel, mask_nulls in ((0.1, True), (0.6, False), (1.0, False), (0.0, False)):
        mask = HostArray(A.Boolean, rng.random(n) < sel, (rng.random(n) < 0.95) if mask_nulls else None)
        out = K.filter_record_batch(A.RecordBatch(names, [c.to_device(ctx) for c in cols]), mask.to_device(ctx))
        exp = [oracle.filter(c, mask) for c in cols]
        assert out.num_rows() == len(exp[0]), f"selectivity {sel}"
        for i, (c, e) in enumerate(zip(out.columns, exp)):
            check(c, e, f"column {i} ({spec[i][0]}) at selectivity {sel}")
            assert_same_nulls_presence(host(c), e, f"column {i} at selectivity {sel}")


# --------------------------------------------------------------- fuzz: take
@pytest.mark.parametrize("idt", [A.UInt8, A.Int8, A.UInt16, A.Int16, A.UInt32, A.Int32, A.UInt64, A.Int64], ids=str)
def test_fuzz_take(ctx, oracle, idt):
    rng = np.random.default_rng(_seed(idt.name))
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
    # check_bounds with index nulls only tests `index >= len` (take.rs:183-191): a negative valid index
    # passes the check and then panics in take_native after the u32 reinterpretation
    neg = HostArray(A.Int32, np.array([1, -1, 0], dtype=np.int32), np.array([True, True, False]))
    with pytest.raises(A.Panic) as ei:
        K.take(v, neg.to_device(ctx), K.TakeOptions(True))
    assert str(ei.value) == "Out-of-bounds index 4294967295"
    i = HostArray(A.UInt32, np.array([1, 400, 2], dtype=np.uint32), np.array([True, True, False])).to_device(ctx)
    with pytest.raises(A.Panic) as ei:
        K.take(v, i)
    assert str(ei.value) == "Out-of-bounds index 400"


def test_take_deferred_mode_parity_and_late_oob(ctx, oracle):
    """ah_take in deferred mode only ENQUEUES (no read-back): results are the oracle's (null count
    resolved lazily), and an out-of-bounds index — the reference's panic (take.rs:447,454) — surfaces at the NEXT
    ah_synchronize / ah_array_resolve with the synchronous call's exact text, the first fault in stream order, once."""
    rng = np.random.default_rng(_seed("take-deferred"))
    v4 = HostArray(A.Int32, np.arange(4, dtype=np.int32)).to_device(ctx)
    cases = []
    for it, (vdt, idt) in enumerate([(A.Int64, A.UInt32), (A.Int8, A.Int16), (A.Float64, A.Int64), (A.Boolean, A.UInt8), (A.Int32, A.Int32)]):
        vlen = int(rng.integers(1, 200 if idt in (A.UInt8,) else 5000))
        n = int(rng.integers(1, 9000))
        hv = HostArray(vdt, _rand_values(rng, vdt, vlen), (rng.random(vlen) < 0.8) if it % 2 else None)
        idx = rng.integers(0, min(vlen, np.iinfo(idt.np_dtype).max), n).astype(idt.np_dtype)
        ivalid = (rng.random(n) < 0.85) if it % 3 == 0 else None
        if ivalid is not None:
            idx = np.where(ivalid, idx, np.iinfo(idt.np_dtype).max).astype(idt.np_dtype)
        hi = HostArray(idt, idx, ivalid)
        cases.append((hv, hi, hv.to_device(ctx), hi.to_device(ctx)))
    n0 = len(cases[0][1])
    hi2 = HostArray(A.UInt32, rng.integers(0, n0, 50).astype(np.uint32))  # into the FIRST take's result (n0 rows)
    with ctx.deferred_mode():
        outs = [K.take(dv, di) for _hv, _hi, dv, di in cases]  # enqueued back to back, nothing read back
        chained = K.take(outs[0], hi2.to_device(ctx))  # a deferred result as the next call's values
    for (hv, hi, _dv, _di), got in zip(cases, outs):
        check(got, oracle.take(hv, hi), f"deferred take {hv.data_type} by {hi.data_type}")
    check(chained, oracle.take(oracle.take(cases[0][0], cases[0][1]), hi2), "deferred take of a deferred take")
    # out-of-bounds: nothing at return, the panic at the next synchronisation — the three wordings
    for mk, text in ((lambda: HostArray(A.Int32, np.array([1, -1], dtype=np.int32)), "index out of bounds: the len is 4 but the index is 4294967295"),
                     (lambda: HostArray(A.UInt32, np.array([1, 400, 2], dtype=np.uint32), np.array([True, True, False])), "Out-of-bounds index 400"),
                     (lambda: HostArray(A.Int16, np.array([2, 3, -5, 9], dtype=np.int16)), "index out of bounds: the len is 4 but the index is 4294967291")):
        ctx.set_deferred(True)
        try:
            bad = K.take(v4, mk().to_device(ctx))      # returns: only enqueued
            good = K.take(v4, HostArray(A.UInt32, np.array([3, 0], dtype=np.uint32)).to_device(ctx))
            also_bad = K.take(v4, HostArray(A.UInt32, np.array([77], dtype=np.uint32)).to_device(ctx))  # a later fault: the first one wins
            with pytest.raises(A.Panic) as ei:
                ctx.synchronize()
            assert str(ei.value) == text
            ctx.synchronize()  # raised once; the slot is re-armed
            assert host(good).to_pylist() == [3, 0]
            del bad, also_bad
        finally:
            ctx.set_deferred(False)
    with ctx.deferred_mode():  # Boolean values: BooleanBuffer::value's assertion
        hb = HostArray(A.Boolean, np.array([True, False, True]))
        K.take(hb.to_device(ctx), HostArray(A.UInt32, np.array([0, 5], dtype=np.uint32)).to_device(ctx))
        with pytest.raises(A.Panic) as ei:
            ctx.synchronize()
        assert str(ei.value) == "assertion failed: idx < self.bit_len"
    # round 6: a call that WAITS internally (filter reads its count back) while a deferred take's fault is
    # outstanding must not hand the garbage on silently: it fails, and ah_synchronize still reports the panic in full
    ctx.set_deferred(True)
    try:
        bad = K.take(v4, HostArray(A.UInt32, np.array([1, 400, 2], dtype=np.uint32)).to_device(ctx))
        with pytest.raises((A.Panic, A.array.HipError)) as ei:
            K.filter(bad, HostArray(A.Boolean, np.array([True, False, True])).to_device(ctx))
        assert "deferred ah_take" in str(ei.value) or "assert" in str(ei.value)
        with pytest.raises(A.Panic) as ei:
            ctx.synchronize()
        assert str(ei.value) == "index out of bounds: the len is 4 but the index is 400"
        ctx.synchronize()
        assert host(K.filter(v4, HostArray(A.Boolean, np.array([True, False, True, True])).to_device(ctx))).to_pylist() == [0, 2, 3]
        del bad
    finally:
        ctx.set_deferred(False)
    # check_bounds and string values stay synchronous in deferred mode (errors at return)
    with ctx.deferred_mode():
        with pytest.raises(A.array.ComputeError):
            K.take(v4, HostArray(A.Int32, np.array([1, 9], dtype=np.int32)).to_device(ctx), K.TakeOptions(True))
    # a recorded take: every replay re-arms the check
    ok_idx = HostArray(A.UInt32, np.array([3, 1, 1, 0], dtype=np.uint32)).to_device(ctx)
    with ctx.graph_capture() as g:
        t = K.take(v4, ok_idx)
    g.launch()
    ctx.synchronize()
    assert host(t).to_pylist() == [3, 1, 1, 0]


# -------------------------------------------------------------- fuzz: arith
@pytest.mark.parametrize("dt", [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64,
                                A.Float32, A.Float64], ids=str)
def test_fuzz_arith(ctx, oracle, dt):
    rng = np.random.default_rng(_seed(dt.name))
    is_f = np.dtype(dt.np_dtype).kind == "f"
    for it in range(10):
        n = int(rng.integers(1, 5000))
        if is_f:
            a, b = _rand_values(rng, dt, n), _rand_values(rng, dt, n)
            b[rng.random(n) < 0.02] = 0
            ut = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
            top = ut(1) << ut(a.dtype.itemsize * 8 - 1)
            for x in (a, b):  # +-inf, and NaNs of both signs with payloads (quiet and signalling)
                x[rng.random(n) < 0.01] = np.inf
                x[rng.random(n) < 0.01] = -np.inf
                k = rng.random(n) < 0.01
                expo = ut(0x7FF0000000000000 if a.dtype.itemsize == 8 else 0x7F800000)
                payload = rng.integers(1, 1 << (52 if a.dtype.itemsize == 8 else 23), n).astype(ut)
                sign = np.where(rng.random(n) < 0.5, top, ut(0)).astype(ut)
                x.view(ut)[k] = (expo | payload | sign)[k]
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
            rb, drb = hb, db
            if not is_f and op in (6, 7) and it % 4 != 3:  # mostly avoid /0; every 4th iter keeps zeros
                rb = HostArray(dt, np.where(b == 0, 1, b).astype(dt.np_dtype), bv)
                drb = rb.to_device(ctx)
            try:
                exp = oracle.arith(op, ha, rb)
            except A.ArrowError as ex:  # checked op failed in the oracle: HIP must fail identically
                with pytest.raises(type(ex)) as ei:
                    ARITH_FN[op](da, drb)
                assert ei.value.message == ex.message, f"{dt} op {op} iter {it}"
                continue
            got = ARITH_FN[op](da, drb)
            # floats: correctly rounded on both sides -> 0 ULP (stated bar <= 1 ULP)
            if is_f:
                check_float(got, exp, f"{dt} op {op}", a, np.asarray(rb.values))
            else:
                check_exact(got, exp, f"{dt} op {op} iter {it}")


INT_TYPES = [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64]


def _full_range_ints(rng, dt, n):
    """Uniform over the whole encoding space of the width, with the edge values planted: MIN, MAX, -1, 0, 1, MIN + 1."""
    info = np.iinfo(dt.np_dtype)
    x = rng.integers(info.min, info.max, n, dtype=dt.np_dtype, endpoint=True)
    edges = [info.min, info.max, 0, 1, info.min + 1, info.max - 1] + ([-1, -2] if info.min < 0 else [])
    k = rng.random(n) < 0.08
    x[k] = rng.choice(np.array(edges, dtype=dt.np_dtype), int(k.sum()))
    return x


def _arith_same(oracle, op, ha, hb, da, db, msg, **sc):
    """One op through the oracle and the device: identical result (raw bytes, null slots included) or identical failure
    (error class AND text: the first failing valid row's operands are in it)."""
    try:
        exp = oracle.arith(op, ha, hb, **sc)
    except A.ArrowError as ex:
        with pytest.raises(type(ex)) as ei:
            ARITH_FN[op](da, db)
        assert ei.value.message == ex.message, msg
        return False
    check_exact(ARITH_FN[op](da, db), exp, msg)
    return True


@pytest.mark.parametrize("dt", INT_TYPES, ids=str)
def test_fuzz_arith_full_range_integers(ctx, oracle, dt):
    """the integer ops over FULL-RANGE operands, all eight widths (the 8- and 16-bit kernels run 8-16 rows
    per lane: where a sign-extension slip would hide).  Wrapping ops must match bit for bit; checked ops fail on the first
    failing VALID row with the reference's text (`Overflow happened on: a + b`, arithmetic.rs:147-260); `MIN / -1` overflows,
    `MIN % -1` is 0 (numeric.rs:345-351), a zero divisor under a NULL slot is not an error (try_binary visits valid slots
    only, arity.rs:285-294).  Arrays, sliced arrays, scalars on either side."""
    rng = np.random.default_rng(_seed("fullrange-" + dt.name))
    info = np.iinfo(dt.np_dtype)
    ok_count = fail_count = 0
    for it in range(6):
        n = int(rng.integers(1, 6000))
        a, b = _full_range_ints(rng, dt, n), _full_range_ints(rng, dt, n)
        av = (rng.random(n) < 0.85) if it % 2 == 0 else None
        bv = (rng.random(n) < 0.85) if it % 3 != 1 else None
        if it == 2:  # every zero divisor sits under a NULL slot: div / rem must succeed unless MIN / -1 shows up
            bv = (b != 0) & (rng.random(n) < 0.9)
        if it == 3:  # small second operand: the checked forms mostly succeed at full-range first operands
            b = rng.integers(0, 2, n).astype(dt.np_dtype)
        if it == 4 and info.min < 0:  # MIN / -1 and MIN % -1 as the ONLY special rows
            a = np.where(rng.random(n) < 0.5, info.min, a).astype(dt.np_dtype)
            b = np.full(n, -1, dtype=dt.np_dtype)
        ha, hb = HostArray(dt, a, av), HostArray(dt, b, bv)
        da, db = ha.to_device(ctx), hb.to_device(ctx)
        off = int(rng.integers(0, min(n, 130)))
        for op in range(8):
            r = _arith_same(oracle, op, ha, hb, da, db, f"{dt} full-range op {op} iter {it}")
            ok_count, fail_count = ok_count + r, fail_count + (not r)
            _arith_same(oracle, op, ha.slice(off, n - off), hb.slice(off, n - off), da.slice(off, n - off), db.slice(off, n - off),
                        f"{dt} full-range op {op} iter {it} sliced {off}")
            for sv in (a[int(rng.integers(0, n))], info.min, info.max, dt.np_dtype(1)) + ((dt.np_dtype(-1),) if info.min < 0 else ()):
                hs = HostArray(dt, np.array([sv], dtype=dt.np_dtype))
                ds = A.Scalar(hs.to_device(ctx))
                _arith_same(oracle, op, ha, hs, da, ds, f"{dt} full-range op {op} iter {it} rscalar {sv}", r_scalar=True)
                _arith_same(oracle, op, hs, hb, ds, db, f"{dt} full-range op {op} iter {it} lscalar {sv}", l_scalar=True)
        for wrapping, fn in ((False, K.neg), (True, K.neg_wrapping)):
            try:
                exp = oracle.neg(ha, wrapping=wrapping)
            except A.ArrowError as ex:
                with pytest.raises(type(ex)) as ei:
                    fn(da)
                assert ei.value.message == ex.message, f"{dt} neg wrapping={wrapping}"
                continue
            check_exact(fn(da), exp, f"{dt} full-range neg wrapping={wrapping} iter {it}")
    assert ok_count >= 18 and fail_count >= 6, (ok_count, fail_count)  # the wrapping forms succeeded, the checked forms failed


@pytest.mark.parametrize("dt", [A.Float32, A.Float64], ids=str)
def test_fuzz_arith_float_bit_patterns(ctx, oracle, dt):
    """All eight ops of arrow_arith::numeric on float operands drawn over the WHOLE encoding space (`_float_strata`):
    denormal inputs and results, overflow to +-inf in mul, div around the subnormal boundary, and rem (fmod) across
    exponent gaps of hundreds of binades — arrays and scalars on either side — bit-exact against the oracle under the
    usual "both operands NaN" exclusion.  ArrowNativeTypeOp for floats: arrow-array/src/arithmetic.rs:308-430
    (add == add_wrapping, div / rem never error, `%` is fmod); numeric.rs:1364-1397 holds the reference's own literals."""
    rng = np.random.default_rng(_seed("bits-" + dt.name))
    for it in range(4):
        a, b = _float_strata(rng, dt, int(rng.integers(600, 9000)))
        n = len(a)
        av = (rng.random(n) < 0.9) if it % 2 == 0 else None
        bv = (rng.random(n) < 0.9) if it == 1 else None
        ha, hb = HostArray(dt, a, av), HostArray(dt, b, bv)
        da, db = ha.to_device(ctx), hb.to_device(ctx)
        scalars = [b[int(i)] for i in rng.integers(0, n, 3)] + [dt.np_dtype(0.0), dt.np_dtype(np.inf)]
        for op in range(8):
            check_float(ARITH_FN[op](da, db), oracle.arith(op, ha, hb), f"{dt} bit-pattern op {op} iter {it}", a, b)
            for sv in scalars[:2] if it else scalars:
                hs = HostArray(dt, np.array([sv], dtype=dt.np_dtype))
                ds = A.Scalar(hs.to_device(ctx))
                full = np.full(n, sv, dtype=dt.np_dtype)
                check_float(ARITH_FN[op](da, ds), oracle.arith(op, ha, hs, r_scalar=True), f"{dt} bit-pattern op {op} rscalar {sv!r}", a, full)
                check_float(ARITH_FN[op](ds, db), oracle.arith(op, hs, hb, l_scalar=True), f"{dt} bit-pattern op {op} lscalar {sv!r}", full, b)
        check_float(K.neg(da), oracle.neg(ha), f"{dt} bit-pattern neg")
        check_float(K.neg_wrapping(da), oracle.neg(ha, wrapping=True), f"{dt} bit-pattern neg_wrapping")


def test_cast_chain_is_the_step_by_step_casts(ctx, oracle):
    """ah_cast_chain (round 5): cast(cast(x, t0), t1) ... as one call — byte-exact against the oracle's
    step-by-step casts.  Int64 -> Float64 -> Utf8 / LargeUtf8 takes the fused path (text straight from the Int64 column,
    `v as f64` in registers: values >= 2^53 round, exponent forms appear); other chains run step by step inside the library.
    Reference arms: cast/mod.rs:1664 (Int64 -> Float64), :1549-1553 + cast/string.rs:21-39 (Float64 -> Utf8)."""
    rng = np.random.default_rng(_seed("cast-chain"))
    for it in range(4):
        n = int(rng.integers(1, 70000))
        v = rng.integers(-10**6, 10**6, n).astype(np.int64)
        k = rng.random(n) < 0.05
        v[k] = rng.integers(-2**63, 2**63 - 1, int(k.sum()), dtype=np.int64)  # full range: rounding above 2^53, "1e16"-style output
        edge = np.array([0, 1, -1, 2**53, 2**53 + 1, -(2**53) - 1, 2**63 - 1, -2**63, 10**15, 10**16, 10**17, 999999999999999999], dtype=np.int64)
        v[:min(n, len(edge))] = edge[:min(n, len(edge))]
        h = HostArray(A.Int64, v, (rng.random(n) < 0.9) if it % 2 == 0 else None)
        d = h.to_device(ctx)
        for to in (A.LargeUtf8, A.Utf8):
            exp = oracle.cast(oracle.cast(h, A.Float64), to)
            check_exact(K.cast_chain(d, [A.Float64, to]), exp, f"chain Int64 -> Float64 -> {to} iter {it}")
            check_exact(K.cast_chain(d.slice(7, n - 7), [A.Float64, to]) if n > 7 else K.cast_chain(d, [A.Float64, to]),
                        oracle.cast(oracle.cast(h.slice(7, n - 7), A.Float64), to) if n > 7 else exp, f"chain sliced -> {to}")
        # unfused chains: same answer as the separate calls
        check_exact(K.cast_chain(d, [A.Float64, A.Float32, A.Int32]), oracle.cast(oracle.cast(oracle.cast(h, A.Float64), A.Float32), A.Int32), "three numeric steps")
        check_exact(K.cast_chain(d, [A.Int32, A.LargeUtf8]), oracle.cast(oracle.cast(h, A.Int32), A.LargeUtf8), "Int64 -> Int32 -> LargeUtf8")
        check_exact(K.cast_chain(d, [A.Float64]), oracle.cast(h, A.Float64), "one step")
    with pytest.raises(A.array.CastError):
        K.cast_chain(d, [A.Float64, A.Utf8, A.Utf8View, A.Boolean])


def _f16_strata(rng, n):
    """Float16 operand pairs over the whole encoding space (the sampler that pins the oracle: tests/test_oracle_golden.py)."""
    from test_oracle_golden import f16_strata
    return f16_strata(rng, n)


def check_f16(got_dev, exp_host, msg, lhs=None, rhs=None):
    """Bit equality on the 16-bit patterns, NaNs included, except where BOTH operands are NaN (check_float's rule)."""
    g = host(got_dev)
    assert_same_nulls_presence(g, exp_host, msg)
    gv, ev = np.asarray(g.values).view(np.uint16), np.asarray(exp_host.values).view(np.uint16)
    m = np.ones(len(gv), dtype=bool)
    if lhs is not None and rhs is not None:
        both = np.isnan(np.asarray(lhs)) & np.isnan(np.asarray(rhs))
        assert np.all(np.isnan(gv[both].view(np.float16)) & np.isnan(ev[both].view(np.float16))), f"{msg} NaN op NaN must be NaN"
        m = ~both
    bad = np.nonzero(gv[m] != ev[m])[0]
    assert len(bad) == 0, f"{msg}: {len(bad)} patterns differ, first {bad[:4]}: got {gv[m][bad[:4]]} want {ev[m][bad[:4]]}"
    assert got_dev.null_count() == exp_host.null_count, f"{msg} null_count"


def test_float16_every_encoding_neg_and_casts(ctx, oracle):
    """VERDICT r04 missing #3: Float16 through neg / neg_wrapping and the numeric casts on ALL 65 536 encodings (arrays with and
    without nulls, sliced), both directions, safe and unsafe — bit-exact against the oracle, whose conversions are pinned to
    numpy float16 on the CPU.  Reference arms: numeric.rs:113 (neg), cast/mod.rs:1578-1697 (`(Int64, Float16)`, `(Float16, Float64)`, ...)."""
    h = np.arange(1 << 16, dtype=np.uint32).astype(np.uint16).view(np.float16)
    rng = np.random.default_rng(_seed("f16-casts"))
    for valid in (None, rng.random(len(h)) < 0.9):
        hh = HostArray(A.Float16, h, valid)
        dh = hh.to_device(ctx)
        check_f16(K.neg(dh), oracle.neg(hh), "f16 neg")
        check_f16(K.neg_wrapping(dh), oracle.neg(hh, wrapping=True), "f16 neg_wrapping")
        check_f16(K.neg(dh.slice(77, 4001)), oracle.neg(hh.slice(77, 4001)), "f16 neg sliced")
        for to in (A.Float32, A.Float64, A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64):
            check_exact(K.cast(dh, to), oracle.cast(hh, to), f"cast f16 -> {to}")
            check_exact(K.cast(dh.slice(33, 5000), to), oracle.cast(hh.slice(33, 5000), to), f"cast f16 -> {to} sliced")
        check_f16(K.cast(dh, A.Float16), oracle.cast(hh, A.Float16), "cast f16 -> f16")
    # unsafe mode: the first value that does not fit is reported with its Debug text (f16 prints as its f32 value)
    few = HostArray(A.Float16, np.array([1.5, 300.0, -2.0, np.nan], dtype=np.float16))
    for to in (A.Int8, A.UInt8, A.Int16):
        try:
            exp = oracle.cast(few, to, safe=False)
        except A.ArrowError as ex:
            with pytest.raises(type(ex)) as ei:
                K.cast_with_options(few.to_device(ctx), to, K.CastOptions(safe=False))
            assert ei.value.message == ex.message
        else:
            check_exact(K.cast_with_options(few.to_device(ctx), to, K.CastOptions(safe=False)), exp, f"unsafe cast f16 -> {to}")
    # -> Float16: every f16 value as f32 / f64, its f32 neighbours and the exact midpoints between adjacent f16 values
    fin = h[np.isfinite(h)].astype(np.float32)
    with np.errstate(over="ignore"):
        nxt = np.nextafter(h[np.isfinite(h)], np.float16(np.inf)).astype(np.float32)
    mid = ((fin.astype(np.float64) + nxt.astype(np.float64)) / 2).astype(np.float32)
    nanbits = np.array([0x7FC00000, 0xFFC00000, 0x7F800001, 0xFF800001, 0x7FA00000, 0x7FFFFFFF, 0x7F802000], dtype=np.uint32).view(np.float32)
    cand = np.concatenate([fin, np.nextafter(fin, np.float32(np.inf)), np.nextafter(fin, np.float32(-np.inf)), mid,
                           np.nextafter(mid, np.float32(np.inf)), np.nextafter(mid, np.float32(-np.inf)), nanbits,
                           np.array([65504, 65519.99, 65520, 65536, 1e9, np.inf, -np.inf, 2.0**-24, 2.0**-25, 2.0**-25 * 1.0001, 2.0**-26,
                                     -2.0**-25, 1e-30, 0.0, -0.0], dtype=np.float32)])
    with np.errstate(invalid="ignore"):  # (the signalling NaNs among the candidates)
        cand64 = np.concatenate([cand.astype(np.float64), np.array([1.0 + 2.0**-11 + 2.0**-30, 2049.0000001, 65519.999999])])
    for src_dt, vals in ((A.Float32, cand), (A.Float64, cand64)):
        hs = HostArray(src_dt, vals.astype(src_dt.np_dtype), rng.random(len(vals)) < 0.95)
        check_f16(K.cast(hs.to_device(ctx), A.Float16), oracle.cast(hs, A.Float16), f"cast {src_dt} -> f16")
    for src_dt in INT_TYPES:
        info = np.iinfo(src_dt.np_dtype)
        iv = np.concatenate([rng.integers(max(info.min, -70000), min(info.max, 70000), 4000, endpoint=True),
                             rng.integers(info.min, info.max, 3000, dtype=src_dt.np_dtype, endpoint=True).astype(object),
                             np.array([x for x in (2049, 2051, 65519, 65520, -65520, 16777217, info.min, info.max) if info.min <= x <= info.max], dtype=object)])
        hs = HostArray(src_dt, np.array([int(x) for x in iv], dtype=src_dt.np_dtype), rng.random(len(iv)) < 0.9)
        check_f16(K.cast(hs.to_device(ctx), A.Float16), oracle.cast(hs, A.Float16), f"cast {src_dt} -> f16")


def test_float16_arith_and_compare_bit_patterns(ctx, oracle):
    """Float16 add / sub / mul / div / rem (and the wrapping aliases) and the eight compares on a stratified sample of encoding
    PAIRS: computed as `half` does — to f32, one operation, one rounding back; totalOrder / bit equality on the 16-bit pattern
    (numeric.rs:240, arithmetic.rs:400-430) — arrays, sliced arrays, scalars on either side, bit-exact."""
    rng = np.random.default_rng(_seed("f16-arith"))
    for it in range(3):
        a, b = _f16_strata(rng, int(rng.integers(3000, 40000)))
        n = len(a)
        av = (rng.random(n) < 0.9) if it % 2 == 0 else None
        bv = (rng.random(n) < 0.9) if it == 1 else None
        ha, hb = HostArray(A.Float16, a, av), HostArray(A.Float16, b, bv)
        da, db = ha.to_device(ctx), hb.to_device(ctx)
        off = int(rng.integers(1, 70))
        scalars = [b[int(i)] for i in rng.integers(0, n, 2)] + [np.float16(0.0), np.float16(np.inf), np.float16(-0.0)]
        for op in range(8):
            check_f16(ARITH_FN[op](da, db), oracle.arith(op, ha, hb), f"f16 op {op} iter {it}", a, b)
            check_f16(ARITH_FN[op](da.slice(off, n - off), db.slice(off, n - off)), oracle.arith(op, ha.slice(off, n - off), hb.slice(off, n - off)),
                      f"f16 op {op} sliced", a[off:], b[off:])
            for sv in scalars[:2] if it else scalars:
                hs = HostArray(A.Float16, np.array([sv], dtype=np.float16))
                ds = A.Scalar(hs.to_device(ctx))
                full = np.full(n, sv, dtype=np.float16)
                check_f16(ARITH_FN[op](da, ds), oracle.arith(op, ha, hs, r_scalar=True), f"f16 op {op} rscalar {sv!r}", a, full)
                check_f16(ARITH_FN[op](ds, db), oracle.arith(op, hs, hb, l_scalar=True), f"f16 op {op} lscalar {sv!r}", full, b)
            exp = oracle.compare(op, ha, hb)
            got = CMP_FN[op](da, db)
            check(got, exp, f"f16 cmp op {op}")
            assert_same_nulls_presence(host(got), exp, f"f16 cmp op {op}")
            check(CMP_FN[op](da.slice(off, n - off), db.slice(off, n - off)),
                  oracle.compare(op, ha.slice(off, n - off), hb.slice(off, n - off)), f"f16 cmp op {op} sliced")
            for i in rng.integers(0, n, 2):
                sc = HostArray(A.Float16, b[int(i):int(i) + 1].copy(), None)
                dsc = A.Scalar(sc.to_device(ctx))
                check(CMP_FN[op](da, dsc), oracle.compare(op, ha, sc, r_scalar=True), f"f16 cmp op {op} rscalar")
                check(CMP_FN[op](dsc, db), oracle.compare(op, sc, hb, l_scalar=True), f"f16 cmp op {op} lscalar")


@pytest.mark.parametrize("dt", [A.Float32, A.Float64], ids=str)
def test_fuzz_cmp_float_bit_patterns(ctx, oracle, dt):
    """The eight arrow_ord::cmp ops on the same operand space: totalOrder / bit equality (arithmetic.rs:400-410) must
    hold for EVERY encoding — denormals, NaN payloads of both signs, signed zeros, ties — arrays, sliced arrays and
    scalars on either side."""
    rng = np.random.default_rng(_seed("cmpbits-" + dt.name))
    for it in range(3):
        a, b = _float_strata(rng, dt, int(rng.integers(600, 9000)))
        n = len(a)
        av = (rng.random(n) < 0.8) if it % 2 == 0 else None
        bv = (rng.random(n) < 0.8) if it == 1 else None
        ha, hb = HostArray(dt, a, av), HostArray(dt, b, bv)
        da, db = ha.to_device(ctx), hb.to_device(ctx)
        off = int(rng.integers(1, 70))
        for op in range(8):
            exp = oracle.compare(op, ha, hb)
            got = CMP_FN[op](da, db)
            check(got, exp, f"{dt} bit-pattern cmp op {op}")
            assert_same_nulls_presence(host(got), exp, f"{dt} bit-pattern cmp op {op}")
            check(CMP_FN[op](da.slice(off, n - off), db.slice(off, n - off)),
                  oracle.compare(op, ha.slice(off, n - off), hb.slice(off, n - off)), f"{dt} bit-pattern cmp op {op} sliced")
            for i in rng.integers(0, n, 2):
                sc = HostArray(dt, b[int(i):int(i) + 1].copy(), None)
                dsc = A.Scalar(sc.to_device(ctx))
                check(CMP_FN[op](da, dsc), oracle.compare(op, ha, sc, r_scalar=True), f"{dt} bit-pattern cmp op {op} rscalar")
                check(CMP_FN[op](dsc, db), oracle.compare(op, sc, hb, l_scalar=True), f"{dt} bit-pattern cmp op {op} lscalar")


def test_config0_filter_int32_plumbing(ctx, oracle):
    """BASELINE.json configs[0], by name: filter() on a 2^20-row Int32 PrimitiveArray, Bernoulli(0.5) predicate, no nulls
    (the shape of arrow/benches/filter_kernels.rs:39-45 at the size BASELINE quotes) — through the oracle AND through
    the drop-in boundary.  At 50 % selectivity `IterationStrategy::default_strategy` (filter.rs:346-364) picks
    IndexIterator (selectivity <= 0.8): the oracle reports the strategy it took, the device result must equal its."""
    n = 1 << 20
    rng = np.random.default_rng(20)
    vals = HostArray(A.Int32, rng.integers(-2**31, 2**31 - 1, n, dtype=np.int32))
    mask = HostArray(A.Boolean, rng.random(n) < 0.5)
    exp = oracle.filter(vals, mask)
    k = int(mask.values.sum())
    assert oracle.filter_strategy(mask) == "Indices"
    assert len(exp) == k and 0.49 * n < k < 0.51 * n and exp.valid is None
    assert np.array_equal(exp.values, vals.values[mask.values])  # the oracle against plain numpy on this shape
    got = K.filter(vals.to_device(ctx), mask.to_device(ctx))
    assert got.data_type == A.Int32 and got.length == k and got.nulls() is None
    check_exact(got, exp, "configs[0]")
    # the same call through the raw C ABI (what a Rust host binds): ah_filter(ctx, values, predicate, out)
    dv, dm = vals.to_device(ctx), mask.to_device(ctx)
    out = A._lib.ArrayOut()
    vv, mv = dv.view(), dm.view()
    import ctypes as C_
    ctx.check(ctx.lib.ah_filter(ctx.handle, C_.byref(vv), C_.byref(mv), C_.byref(out)))
    raw = A.Array._from_out(ctx, out, A.Int32)
    check_exact(raw, exp, "configs[0] through ah_filter")
    # FilterBuilder::optimize changes the strategy, never the result (filter.rs:285-303)
    pred = K.FilterBuilder(dm).optimize().build()
    assert pred.count() == k
    check_exact(pred.filter(dv), exp, "configs[0] optimized predicate")


def test_generated_nan_bits_feed_total_order_compare(ctx, oracle):
    """`lt` / `eq` order floats by totalOrder and bit equality
    (arrow-array/src/arithmetic.rs:400-410), so the SIGN of a generated NaN is observable downstream:
    on the x86 reference inf + -inf = 0xFFF8... (negative), hence lt(add(inf, -inf), 0.0) is TRUE.
    The device must agree, end to end and bit for bit (no NaN == NaN allowance)."""
    for dt, ut in ((A.Float64, np.uint64), (A.Float32, np.uint32)):
        npd = dt.np_dtype
        a = np.array([np.inf, 0.0, np.inf, 5.0, np.inf, -0.0, 1.0, -np.inf], dtype=npd)
        b = np.array([-np.inf, np.inf, np.inf, 0.0, 3.0, 0.0, 2.0, np.inf], dtype=npd)
        zero = HostArray(dt, np.zeros(len(a), dtype=npd))
        ha, hb = HostArray(dt, a), HostArray(dt, b)
        da, db, dz = ha.to_device(ctx), hb.to_device(ctx), zero.to_device(ctx)
        for op, fn in ((1, K.add_wrapping), (3, K.sub_wrapping), (5, K.mul_wrapping), (6, K.div), (7, K.rem)):
            exp = oracle.arith(op, ha, hb)
            got = fn(da, db)
            check_float(got, exp, f"{dt} op {op} generated NaNs", a, b)
            ev = np.asarray(exp.values)
            gen = np.isnan(ev)
            if gen.any():  # every NaN here was generated: the x86 default NaN, sign bit set
                assert np.all(ev[gen].view(ut) >> ut(ev.dtype.itemsize * 8 - 1) == 1)
            for cmp_op, cfn in ((2, K.lt), (0, K.eq), (3, K.lt_eq)):
                check_exact(cfn(got, dz), oracle.compare(cmp_op, exp, zero), f"{dt} cmp {cmp_op} after op {op}")
        # the headline case, spelled out
        s = K.add_wrapping(da, db)
        l = host(K.lt(s, dz))
        assert bool(np.asarray(l.values)[0]) is True, "lt(add(inf, -inf), 0.0) must be true as on the x86 reference"


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
    rng = np.random.default_rng(_seed(dt.name) + 1)
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


# ------------------------------------------------ strings through filter / take
def test_fuzz_string_filter_take(ctx, oracle):
    """filter_bytes / take_bytes on device vs the oracle: Utf8 and LargeUtf8, sliced inputs,
    null slots that carry bytes, nullable indices, empty strings, multi-byte UTF-8."""
    from test_oracle_golden import _rand_strings
    rng = np.random.default_rng(27)
    for it in range(30):
        n = int(rng.integers(1, 400)) if it < 24 else int(rng.integers(5000, 40000))
        dt = A.Utf8 if it % 2 else A.LargeUtf8
        vals = _rand_strings(rng, n, 14 if it < 24 else 40)
        valid = (rng.random(n) < 0.8) if it % 3 else None
        h = HostArray(dt, vals, valid)
        d = A.Array.from_strings(vals, valid, dt, ctx)
        sel = [0.0, 1.0, 0.05, 0.5, 0.95][it % 5]
        m = HostArray(A.Boolean, rng.random(n) < sel, (rng.random(n) < 0.9) if it % 4 == 0 else None)
        got = K.filter(d, m.to_device(ctx))
        check(got, oracle.filter(h, m), f"filter {it}")
        off = int(rng.integers(0, min(n, 6)))
        ln = n - off - int(rng.integers(0, min(n - off, 3) + 1)) if n - off > 0 else 0
        if ln > 0:
            ms = HostArray(A.Boolean, m.values[:ln], None if m.valid is None else m.valid[:ln])
            check(K.filter(d.slice(off, ln), ms.to_device(ctx)), oracle.filter(h.slice(off, ln), ms), f"sliced {it}")
        k = int(rng.integers(0, 2 * n))
        idt = [A.UInt32, A.Int32, A.UInt64, A.Int64, A.UInt16][it % 5]
        idx = rng.integers(0, min(n, np.iinfo(idt.np_dtype).max), k).astype(idt.np_dtype)
        ivalid = (rng.random(k) < 0.85) if it % 2 == 0 else None
        if ivalid is not None:
            idx = np.where(ivalid, idx, np.iinfo(idt.np_dtype).max).astype(idt.np_dtype)  # garbage under nulls
        hi = HostArray(idt, idx, ivalid)
        got = K.take(d, hi.to_device(ctx))
        exp = oracle.take(h, hi)
        check(got, exp, f"take {it}")
        if k:
            assert_same_nulls_presence(host(got), exp, f"take {it}")
    d = A.Array.from_strings(["a", "b"], None, A.Utf8, ctx)
    with pytest.raises(A.Panic) as ei:
        K.take(d, HostArray(A.UInt32, np.array([0, 2], dtype=np.uint32)).to_device(ctx))
    assert str(ei.value) == "index out of bounds: the len is 3 but the index is 3"
    with pytest.raises(A.Panic) as ei:
        K.take(d, HostArray(A.UInt32, np.array([7], dtype=np.uint32)).to_device(ctx))
    assert str(ei.value) == "index out of bounds: the len is 3 but the index is 7"
    # cast -> filter -> take chain stays on device: Float64 -> Utf8 -> filter -> take
    f = HostArray(A.Float64, rng.normal(size=5000) * 100, rng.random(5000) < 0.9)
    m = HostArray(A.Boolean, rng.random(5000) < 0.3)
    i = HostArray(A.UInt32, rng.integers(0, 1000, 777).astype(np.uint32))
    got = K.take(K.filter(K.cast(f.to_device(ctx), A.Utf8), m.to_device(ctx)), i.to_device(ctx))
    check(got, oracle.take(oracle.filter(oracle.cast(f, A.Utf8), m), i), "cast->filter->take")


def _string_buffers(arr):
    """(offsets as int64, data bytes) of a device Utf8 / LargeUtf8 array, through plain copies."""
    ow = np.int64 if arr.data_type == A.LargeUtf8 else np.int32
    offs = A.array._copy_dtoh(arr.ctx, arr.offsets.ptr, (arr.length + 1) * np.dtype(ow).itemsize).view(ow).astype(np.int64)
    data = A.array._copy_dtoh(arr.ctx, arr.values.ptr, int(offs[-1])) if offs[-1] else np.zeros(0, np.uint8)
    return offs, data


def test_string_kernels_past_one_scan_workgroup(ctx):
    """The block totals of the string kernels are scanned by ONE launch of chained workgroups (scan_chain.hpp), 1024 totals
    per round: every other string test fits one workgroup.  6 Mi rows = 1 536 filter tiles, 3 Mi indices = 3 072 take rounds,
    6 144+ view blocks -> several chained workgroups.  Checked against results that never touch that scan: filter / take
    commute with the cast (numbers are filtered / taken first, then printed), and the out-of-line offsets of the views must
    be the running sum of the long strings' lengths."""
    n = 6 << 20
    rng = np.random.default_rng(2909)
    vals = np.round(rng.normal(size=n) * 1e9, 3)  # "-1234567890.123": out-of-line in a view
    vals[rng.random(n) < 0.3] = 7.0  # short strings: inline views between the out-of-line ones
    valid = rng.random(n) < 0.9
    for dt in (A.LargeUtf8, A.Utf8):
        x = HostArray(A.Float64, vals, valid).to_device(ctx)
        sx = K.cast(x, dt)
        m = HostArray(A.Boolean, rng.random(n) < 0.1).to_device(ctx)
        got, exp = K.filter(sx, m), K.cast(K.filter(x, m), dt)
        assert got.length == exp.length and got.null_count() == exp.null_count()
        go, gd = _string_buffers(got)
        eo, ed = _string_buffers(exp)
        assert np.array_equal(go, eo), f"{dt} filter offsets"
        # (a null slot keeps its bytes through filter_bytes; the cast wrote none, so the data compare is exact too)
        assert np.array_equal(gd, ed), f"{dt} filter data"
        k = 3 << 20
        idx = rng.integers(0, n, k).astype(np.uint32)
        ivalid = rng.random(k) < 0.95
        hi = HostArray(A.UInt32, idx, ivalid).to_device(ctx)
        got, exp = K.take(sx, hi), K.cast(K.take(x, hi), dt)
        assert got.length == exp.length == k and got.null_count() == exp.null_count()
        go, gd = _string_buffers(got)
        eo, ed = _string_buffers(exp)
        assert np.array_equal(go, eo), f"{dt} take offsets"
        assert np.array_equal(gd, ed), f"{dt} take data"
        assert np.array_equal(got.valid_mask(), exp.valid_mask())
        # -> Utf8View: the long strings' offsets are the exclusive running sum of their lengths, their bytes the source's
        v = K.cast(sx, A.Utf8View)
        so, sd = _string_buffers(sx)
        raw = A.array._copy_dtoh(ctx, v.values.ptr, n * 16).reshape(-1, 16)
        lens = raw[:, 0:4].copy().view(np.uint32).ravel().astype(np.int64)
        assert np.array_equal(lens[valid], (so[1:] - so[:-1])[valid]) and not raw[~valid].any()
        long_rows = valid & (lens > 12)
        ll = lens[long_rows]
        want_off = np.cumsum(ll) - ll
        got_off = raw[:, 12:16].copy().view(np.uint32).ravel().astype(np.int64)[long_rows]
        assert np.array_equal(got_off, want_off), f"{dt} view offsets"
        assert len(v.data_buffers) == 1 and v.data_buffers[0].nbytes == int(ll.sum())
        buf = v.data_buffers[0].to_numpy()
        src = np.repeat(so[:-1][long_rows] - want_off, ll) + np.arange(int(ll.sum()))
        assert np.array_equal(buf, sd[src]), f"{dt} view data"


def test_string_filter_tile_spanning_4gib(ctx):
    """LargeUtf8 filter stages (tile-relative start, local offset) pairs in 4 bytes each; a 4096-row tile whose selected rows
    end 4 GiB or more behind the tile's first offset cannot be encoded, the ranges pass says so and the call reruns with
    8-byte pairs.  64 rows of 80 MiB (5 GiB of text, generated on the device); three selected rows, the last two past 4 GiB."""
    L, rows = 80 << 20, 64
    R = A.array._RawMem
    text = ctx.alloc(rows * L)
    ctx.check(ctx.lib.ah_gen_uniform_small(ctx.handle, text.ptr, 1, rows * L, 77, 0))
    offs = A.array.DeviceBuffer.from_numpy(ctx, (np.arange(rows + 1, dtype=np.int64) * L))
    d = A.Array(ctx, A.LargeUtf8, rows, R(text.ptr, text.nbytes, text), 0, None, 0, 0, R(offs.ptr, offs.nbytes, offs))
    for picks in ([2, 52, 63], [1, 2]):  # the second selection ends before 4 GiB: the narrow pairs serve it
        m = np.zeros(rows, dtype=bool)
        m[picks] = True
        got = K.filter(d, HostArray(A.Boolean, m).to_device(ctx))
        assert got.length == len(picks) and got.null_count() == 0
        go = A.array._copy_dtoh(ctx, got.offsets.ptr, (len(picks) + 1) * 8).view(np.int64)
        assert np.array_equal(go, np.arange(len(picks) + 1, dtype=np.int64) * L)
        for j, r in enumerate(picks):
            a = A.array._copy_dtoh(ctx, got.values.ptr + j * L, L)
            b = A.array._copy_dtoh(ctx, text.ptr + r * L, L)
            assert np.array_equal(a, b), f"row {r}"


def test_string_take_offset_overflow(ctx):
    """take_bytes: i32 offsets past i32::MAX -> ArrowError::OffsetOverflowError(capacity) (take.rs:521)."""
    big = "x" * (1 << 20)
    d = A.Array.from_strings([big, "y"], None, A.Utf8, ctx)
    idx = HostArray(A.UInt32, np.zeros(2049, dtype=np.uint32)).to_device(ctx)
    with pytest.raises(A.array.OffsetOverflowError) as ei:
        K.take(d, idx)
    assert ei.value.message == str(2049 << 20) and str(ei.value) == f"Offset overflow error: {2049 << 20}"
    dl = A.Array.from_strings([big, "y"], None, A.LargeUtf8, ctx)
    assert K.take(dl, idx).values.nbytes == 2049 << 20


# ------------------------------------------------------------ boolean kernels
from test_oracle_golden import BOOL_BIN, BOOL_UN  # noqa: E402
BOOL_FN = {"nullif": K.nullif, "and": K.and_, "or": K.or_, "and_not": K.and_not, "and_kleene": K.and_kleene, "or_kleene": K.or_kleene,
           "not": K.not_, "is_null": K.is_null, "is_not_null": K.is_not_null}


@pytest.mark.parametrize("case", load_golden("boolean"), ids=lambda c: c["name"])
def test_boolean_golden(ctx, case):
    l = dev(case["lhs"], ctx)
    fn = BOOL_FN[case["op"]]
    if case["op"] in BOOL_UN:
        got = fn(l)
    else:
        r = dev(case["rhs"], ctx)
        if "error" in case:
            return expect_err(case, lambda: fn(l, r))
        got = fn(l, r)
    check(got, golden_array(case["expected"]), case["name"])
    if case.get("no_null_buffer"):
        assert got.nulls() is None


def test_fuzz_boolean(ctx, oracle):
    rng = np.random.default_rng(99)
    for it in range(12):
        n = int(rng.integers(1, 9000))
        a = HostArray(A.Boolean, rng.random(n) < 0.5, (rng.random(n) < 0.8) if it % 2 == 0 else None)
        b = HostArray(A.Boolean, rng.random(n) < 0.5, (rng.random(n) < 0.8) if it % 3 == 0 else None)
        da, db = a.to_device(ctx), b.to_device(ctx)
        off = int(rng.integers(0, min(n, 9)))
        for name, op in BOOL_BIN.items():
            exp = oracle.boolean_binary(op, a, b)
            got = BOOL_FN[name](da, db)
            check(got, exp, f"{name} iter {it}")
            assert_same_nulls_presence(host(got), exp, name)
            check(BOOL_FN[name](da.slice(off, n - off), db.slice(off, n - off)),
                  oracle.boolean_binary(op, a.slice(off, n - off), b.slice(off, n - off)), f"{name} sliced")
        for name, op in BOOL_UN.items():
            src = a if name == "not" else HostArray(A.Int32, rng.integers(0, 9, n).astype(np.int32), a.valid)
            exp = oracle.boolean_unary(op, src)
            got = BOOL_FN[name](src.to_device(ctx))
            check(got, exp, name)
            assert_same_nulls_presence(host(got), exp, name)
    for it in range(6):  # nullif: zero-copy values, fresh nulls
        n = int(rng.integers(1, 5000))
        v = HostArray(A.Int64, rng.integers(-9, 9, n), (rng.random(n) < 0.8) if it % 2 else None)
        c = HostArray(A.Boolean, rng.random(n) < 0.3, (rng.random(n) < 0.8) if it % 3 else None)
        dv = v.to_device(ctx)
        got = K.nullif(dv, c.to_device(ctx))
        assert got.values.ptr == dv.values.ptr  # shared values buffer
        exp = oracle.nullif(v, c)
        check(got, exp, "nullif")
        assert_same_nulls_presence(host(got), exp, "nullif")
    # predicate built on device feeds filter directly: lt(col, 3) AND is_not_null(col)
    col = HostArray(A.Int32, rng.integers(0, 9, 5000).astype(np.int32), rng.random(5000) < 0.9)
    dcol = col.to_device(ctx)
    pred = K.and_(K.lt(dcol, A.Scalar.new(3, A.Int32, ctx)), K.is_not_null(dcol))
    hp = oracle.boolean_binary(0, oracle.compare(2, col, HostArray(A.Int32, np.array([3], dtype=np.int32)), r_scalar=True),
                               oracle.boolean_unary(12, col))
    check(K.filter(dcol, pred), oracle.filter(col, hp), "lt -> and -> filter")
    check(K.prep_null_mask_filter(pred), HostArray(A.Boolean, hp.values & (hp.valid if hp.valid is not None else True)))


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


# ------------------------------------------------------- cast -> Utf8 (Ryu)
def _f64_corpus(rng, n):
    specials = np.array([0.0, -0.0, 1.0, 1.5, 2.5, 3.2234, 123.564532, -556132.25, 1e15, 1e16, 1e17, 1e21, 1e22,
                         1e23, 0.1, 0.3, 1e-5, 1e-6, 9.999e-6, 5e-324, 1.7976931348623157e308,
                         2.2250738585072014e-308, 9007199254740993.0, 2.0**63, -(2.0**63), np.nan, np.inf, -np.inf,
                         123456.0, 1e100, 4.35, 299792458.0, 1e-300, -1.2345678901234567e-300])
    bits = rng.integers(0, 2**64, n, dtype=np.uint64).view(np.float64)
    ints = rng.integers(-10**6, 10**6, n).astype(np.float64)
    big = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64).astype(np.float64)
    pow10 = 10.0 ** rng.integers(-320, 308, n // 4) * rng.integers(1, 10, n // 4)
    short = np.round(rng.normal(size=n) * 1000, 3)
    return np.concatenate([specials, bits, ints, big, pow10, short])


def test_cast_f64_to_utf8_vs_oracle(ctx, oracle):
    """Byte-exact strings: device Ryu vs the oracle (libstdc++ shortest digits + ryu layout,
    itself pinned against CPython repr in the CPU suite)."""
    rng = np.random.default_rng(77)
    vals = _f64_corpus(rng, 30000)
    h = HostArray(A.Float64, vals, rng.random(len(vals)) < 0.9)
    for to in (A.Utf8, A.LargeUtf8):
        exp = oracle.cast(h, to)
        got = K.cast(h.to_device(ctx), to)
        check(got, exp, str(to))
        assert_same_nulls_presence(host(got), exp)
    nonull = HostArray(A.Float64, vals[:5000])
    got = K.cast(nonull.to_device(ctx), A.Utf8)
    assert got.nulls() is None  # builder materialises no null buffer without nulls
    check(got, oracle.cast(nonull, A.Utf8))
    # sliced input (unaligned validity)
    d = h.to_device(ctx).slice(3, 20001)
    check(K.cast(d, A.Utf8), oracle.cast(h.slice(3, 20001), A.Utf8), "sliced")


@pytest.mark.parametrize("src", [A.Float64, A.Float32])
def test_cast_to_utf8_sparse_and_dense_tiles(ctx, oracle, src):
    """The side queue's tile decision: a 512-row tile with fewer than 16 shortest-digit (Ryu) rows hands them to the
    dense side passes (tile record of at most 15 {row, length} entries), one with 16 or more keeps them in-kernel.
    Tiles with 0..20 such rows, next to each other inside one 2048-row length-pass workgroup, with nulls, sliced."""
    rng = np.random.default_rng(5150)
    tiles, T = 84, 512
    n = tiles * T + 77  # ragged last tile
    vals = rng.integers(-10**6, 10**6, n).astype(np.float64)
    for j in range(tiles):
        g = j % 21
        rows = j * T + rng.choice(T, g, replace=False)
        kind = rng.integers(0, 4, g)
        gen = np.where(kind == 0, rng.integers(2**53, 2**62, g).astype(np.float64) * rng.choice([-1.0, 1.0], g),
                       np.where(kind == 1, rng.normal(size=g) * 1e-7,
                                np.where(kind == 2, rng.normal(size=g) * 1e300 if src is A.Float64 else rng.normal(size=g) * 1e30,
                                         rng.normal(size=g) * 123.456)))
        vals[rows] = gen
    vals = vals.astype(src.np_dtype)
    h = HostArray(src, vals, rng.random(n) < 0.9)
    d = h.to_device(ctx)
    for to in (A.Utf8, A.LargeUtf8):
        check(K.cast(d, to), oracle.cast(h, to), f"{src}->{to}")
        for off in (5, 64, 511):
            check(K.cast(d.slice(off, n - off - 3), to), oracle.cast(h.slice(off, n - off - 3), to), f"{src}->{to} sliced at {off}")
    allvalid = HostArray(src, vals)
    check(K.cast(allvalid.to_device(ctx), A.LargeUtf8), oracle.cast(allvalid, A.LargeUtf8), "no nulls")


def test_cast_big_integer_doubles_to_utf8(ctx, oracle):
    """Integer-valued doubles in [2^53, 2^64) — what `Int64 as f64` yields beyond 2^53, the cast chain's general rows — take an
    exact 64-bit form of Ryu's digit-removal loop instead of the 128-bit multiplication (csrc/cast_string.hip: d2d_big_int).
    Same digits as Ryu for every class that exercises its rules: random mantissas at every exponent, neighbours of powers
    of ten (long runs of removable digits, "1e19"), halfway cases (last removed digit 5 with zeros below: round half even),
    even / odd mantissas (inclusive / exclusive interval ends), exact powers of two (which stay on the general path)."""
    rng = np.random.default_rng(2653)
    parts = []
    for s in range(1, 12):  # value = m2 << s, m2 a 53-bit mantissa
        m2 = rng.integers(1 << 52, 1 << 53, 30000, dtype=np.uint64)
        parts.append((m2 << np.uint64(s)).astype(np.float64))
    for k in range(15, 20):  # around 10^k and 5 * 10^(k-1): doubles next to them, both sides
        for base in (10 ** k, 5 * 10 ** (k - 1), 25 * 10 ** (k - 2), 125 * 10 ** (k - 3)):
            if base >= 2 ** 64:
                continue
            x = np.float64(base)
            near = [x]
            lo = hi = x
            for _ in range(40):
                lo, hi = np.nextafter(lo, 0.0), np.nextafter(hi, np.inf)
                near += [lo, hi]
            parts.append(np.array([v for v in near if 2.0 ** 53 <= v < 2.0 ** 64], dtype=np.float64))
    # multiples of 5, 50, 500 … (ties after removing 1, 2, 3 … digits) and of 10^j (trailing zeros of the lower bound)
    for j in range(1, 6):
        m = rng.integers((1 << 53) // 10 ** j, (1 << 62) // 10 ** j, 20000, dtype=np.uint64) * np.uint64(10 ** j)
        parts.append(m.astype(np.float64))
        parts.append((m + np.uint64(5 * 10 ** (j - 1))).astype(np.float64))
    parts.append(np.array([2.0 ** e for e in range(53, 64)]))
    vals = np.concatenate(parts)
    vals = np.concatenate([vals, -vals[::7]])
    assert ((np.abs(vals) >= 2.0 ** 53) & (np.abs(vals) < 2.0 ** 64)).all()
    h = HostArray(A.Float64, vals, rng.random(len(vals)) < 0.97)
    check(K.cast(h.to_device(ctx), A.LargeUtf8), oracle.cast(h, A.LargeUtf8), "big integer doubles")
    # the chain's own route: Int64 -> Float64 -> LargeUtf8 in one call
    iv = np.concatenate([rng.integers(-2**63, 2**63 - 1, 200000, dtype=np.int64), rng.integers(-10**6, 10**6, 50000, dtype=np.int64)])
    hi = HostArray(A.Int64, iv, rng.random(len(iv)) < 0.9)
    check(K.cast_chain(hi.to_device(ctx), [A.Float64, A.LargeUtf8]), oracle.cast(oracle.cast(hi, A.Float64), A.LargeUtf8), "chain, full-range Int64")


def test_cast_f32_and_ints_to_utf8(ctx, oracle):
    rng = np.random.default_rng(78)
    f32 = np.concatenate([rng.integers(0, 2**32, 40000, dtype=np.uint64).astype(np.uint32).view(np.float32),
                          np.array([0.0, -0.0, 1.0, 1.5, 1e10, 1e13, 1e14, 3e9, 1e-5, 1e-6, 1e-7, 3.4028235e38, 1e-45,
                                    np.nan, np.inf, -np.inf, 16777216.0, 0.3], dtype=np.float32)])
    h = HostArray(A.Float32, f32, rng.random(len(f32)) < 0.95)
    check(K.cast(h.to_device(ctx), A.Utf8), oracle.cast(h, A.Utf8), "f32")
    for dt in [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64]:
        vals = _rand_values(rng, dt, 5000)
        info = np.iinfo(dt.np_dtype)
        vals[:4] = [info.min, info.max, 0, info.max // 3]
        vals[100:2000] = (vals[100:2000] % 100).astype(dt.np_dtype)
        h = HostArray(dt, vals, rng.random(5000) < 0.9)
        check(K.cast(h.to_device(ctx), A.Utf8), oracle.cast(h, A.Utf8), str(dt))


def _check_views(got, exp_strings_host, msg):
    """A Utf8View result against the oracle's Utf8 strings of the same cast (value_to_string_view and value_to_string
    share the formatter, cast/string.rs:21 vs :41) + the layout rules of a view array (byte_view.rs): length, <= 12
    bytes inline and zero padded, longer ones {4-byte prefix, buffer 0, offset} into the one data buffer."""
    assert got.data_type == A.Utf8View and got.length == len(exp_strings_host)
    ev = exp_strings_host.valid if exp_strings_host.valid is not None else np.ones(len(exp_strings_host), dtype=bool)
    assert np.array_equal(got.valid_mask(), ev), f"{msg}: validity"
    assert (got.validity is None) == (exp_strings_host.null_count == 0), f"{msg}: null buffer presence"
    vals = got.values_numpy()  # decodes the views through the data buffer
    for i, (g, e, v) in enumerate(zip(vals, exp_strings_host.values, ev)):
        if v and g != e:
            raise AssertionError(f"{msg}: row {i}: got {g!r} expected {e!r}")
    raw = A.array._copy_dtoh(got.ctx, got.values.ptr, got.length * 16).reshape(-1, 16)
    lens = raw[:, :4].copy().view(np.uint32).ravel()
    inline = ev & (lens <= 12)
    pad = np.arange(12)[None, :] >= lens[:, None]
    assert not (raw[:, 4:][inline] * pad[inline]).any(), f"{msg}: inline views must be zero padded"
    assert not raw[~ev].any(), f"{msg}: null rows are all-zero views (append_null)"
    long_rows = ev & (lens > 12)
    nb = sum(len(e.encode()) for e, v, ln in zip(exp_strings_host.values, ev, lens) if v and ln > 12)
    if long_rows.any():
        assert len(got.data_buffers) == 1 and got.data_buffers[0].nbytes == nb, f"{msg}: one exact-size data buffer"
        assert not raw[long_rows][:, 8:12].any(), f"{msg}: buffer index 0"
    else:
        assert got.data_buffers == [], f"{msg}: no data buffer when every string is inline"


def test_cast_to_utf8view(ctx, oracle):
    """The `-> Utf8View` arms (VERDICT r03 missing #3): numbers (cast/mod.rs:1546 -> value_to_string_view, cast/string.rs:41)
    and Utf8 / LargeUtf8 (:1302 / :1432).  The reference's own goldens go through THIS arm: test_cast_float_to_utf8view
    (mod.rs:4857-4876: [1.5, 2.5, None] for Float32 and Float64) and test_cast_int_to_utf8view (:4827-4853:
    [None, 8, 9, 10] for every integer type)."""
    for dt in (A.Float32, A.Float64):
        h = HostArray(dt, np.array([1.5, 2.5, 0.0], dtype=dt.np_dtype), np.array([True, True, False]))
        assert K.can_cast_types(dt, A.Utf8View)
        got = K.cast(h.to_device(ctx), A.Utf8View)
        assert got.to_pylist() == ["1.5", "2.5", None]
    for dt in (A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64):
        h = HostArray(dt, np.array([0, 8, 9, 10], dtype=dt.np_dtype), np.array([False, True, True, True]))
        assert K.cast(h.to_device(ctx), A.Utf8View).to_pylist() == [None, "8", "9", "10"]
    rng = np.random.default_rng(4857)
    # Float64 over the whole corpus of the Utf8 tests (exponent forms are > 12 bytes: the out-of-line path), nulls, sliced
    vals = _f64_corpus(rng, 20000)
    h = HostArray(A.Float64, vals, rng.random(len(vals)) < 0.9)
    _check_views(K.cast(h.to_device(ctx), A.Utf8View), oracle.cast(h, A.Utf8), "f64")
    _check_views(K.cast(h.to_device(ctx).slice(5, 9000), A.Utf8View), oracle.cast(h.slice(5, 9000), A.Utf8), "f64 sliced")
    nonull = HostArray(A.Float64, vals[:4000])
    _check_views(K.cast(nonull.to_device(ctx), A.Utf8View), oracle.cast(nonull, A.Utf8), "f64 no nulls")
    # every string inline: small integers as doubles
    small = HostArray(A.Float64, rng.integers(-999, 999, 5000).astype(np.float64), rng.random(5000) < 0.8)
    _check_views(K.cast(small.to_device(ctx), A.Utf8View), oracle.cast(small, A.Utf8), "f64 all inline")
    f32 = rng.integers(0, 2**32, 20000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    h32 = HostArray(A.Float32, f32, rng.random(len(f32)) < 0.95)
    _check_views(K.cast(h32.to_device(ctx), A.Utf8View), oracle.cast(h32, A.Utf8), "f32 bit patterns")
    for dt in (A.Int64, A.UInt64, A.Int8):  # full-range 64-bit integers print 19-20 digits: out of line
        iv = _rand_values(rng, dt, 7000)
        hi = HostArray(dt, iv, rng.random(7000) < 0.9)
        _check_views(K.cast(hi.to_device(ctx), A.Utf8View), oracle.cast(hi, A.Utf8), str(dt))
    # Utf8 / LargeUtf8 -> Utf8View: empty strings, exactly 12 and 13 bytes, multi-byte UTF-8, null slots that carry bytes
    words = ["", "a", "twelve bytes", "thirteen byte", "ünïcödé ✓ text that is long", "x" * 40, "1234567890ab", "1234567890abc"]
    strs = [words[i] for i in rng.integers(0, len(words), 6000)]
    for dt in (A.Utf8, A.LargeUtf8):
        hs = HostArray(dt, strs, rng.random(6000) < 0.85)
        d = A.Array.from_strings(strs, hs.valid, dt, ctx)
        _check_views(K.cast(d, A.Utf8View), hs, f"{dt} -> Utf8View")
        _check_views(K.cast(d.slice(7, 3000), A.Utf8View), hs.slice(7, 3000), f"{dt} sliced -> Utf8View")
    assert K.cast(A.Array.from_strings([], None, A.Utf8, ctx), A.Utf8View).length == 0


def test_cast_chain_config4_shape(ctx, oracle):
    """Config 4: Int64 -> Float64 -> Utf8 on the SURVEY §8d distribution (uniform [-1e6,1e6] with
    1% full range), 10% nulls — 2M rows on both sides."""
    n = 1 << 21
    small = oracle.gen_i64(n, 42, -10**6, 10**6)
    full = oracle.gen_i64(n, 43, -2**63, 2**63 - 1)
    pick = oracle.gen_bits(n, 44, 0.01)
    vals = np.where(pick, full, small)
    valid = oracle.gen_bits(n, 45, 0.9)
    vals[~valid] = 0
    h = HostArray(A.Int64, vals, valid)
    f = K.cast(h.to_device(ctx), A.Float64)
    ef = oracle.cast(h, A.Float64)
    check_exact(f, ef, "stage 1")
    check(K.cast(f, A.LargeUtf8), oracle.cast(ef, A.LargeUtf8), "stage 2")


def test_cast_utf8_offset_overflow(ctx):
    """i32 offsets past 2^31 bytes: the reference panics "byte array offset overflow"
    (generic_bytes_builder.rs:86-87); LargeUtf8 succeeds."""
    n = 1 << 27
    buf = ctx.alloc(n * 8)
    ctx.check(ctx.lib.ah_gen_uniform_f64(ctx.handle, buf.ptr, n, 5, -1e-300, 1e-300, 0))
    R = A.array._RawMem
    arr = A.Array(ctx, A.Float64, n, R(buf.ptr, n * 8, buf), 0)
    with pytest.raises(A.Panic) as ei:
        K.cast(arr, A.Utf8)
    assert str(ei.value) == "byte array offset overflow"
    big = K.cast(arr, A.LargeUtf8)
    assert big.length == n and big.values.nbytes > 2**31


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


@pytest.mark.parametrize("case", orc.load_golden("concat"), ids=lambda c: c["name"])
def test_concat_reference_goldens(ctx, case):
    if "error" in case:
        return expect_err(case, lambda: K.concat([orc.golden_array(p).to_device(ctx) for p in case["pieces"]]))
    got = K.concat([orc.golden_array(p).to_device(ctx) for p in case["pieces"]])
    orc.assert_logical_eq(HostArray.from_device(got), orc.golden_array(case["expected"]), case["name"])


def test_concat_batches(ctx):
    """concat_batches (arrow-select/src/concat.rs:607) and its tests :704, :1548, :1591, :1609."""
    i32 = lambda v: HostArray.from_pylist(v, A.Int32).to_device(ctx)  # noqa: E731
    utf = lambda v: HostArray.from_pylist(v, A.Utf8).to_device(ctx)  # noqa: E731
    b1 = A.RecordBatch(["a", "b"], [i32([1, 2]), utf(["a", "b"])])
    b2 = A.RecordBatch(["a", "b"], [i32([3, 4]), utf(["c", "d"])])
    out = K.concat_batches(["a", "b"], [b1, b2])                      # concat_record_batches :1548
    assert out.names == ["a", "b"] and out.num_columns() == 2 and out.num_rows() == 4
    assert out.column(0).to_pylist() == [1, 2, 3, 4] and out.column(1).to_pylist() == ["a", "b", "c", "d"]
    e1, e2 = A.RecordBatch([], [], 100), A.RecordBatch([], [], 100)
    assert K.concat_batches([], [e1, e2]).num_rows() == 200              # test_concat_batches_no_columns :704
    assert K.concat_batches(["a", "b"], []).num_rows() == 0              # empty input -> RecordBatch::new_empty
    c = A.RecordBatch(["c"], [i32([3, 4])])                              # names differ: the given schema wins :1591
    out = K.concat_batches(["a"], [A.RecordBatch(["a"], [i32([1, 2])]), c])
    assert out.names == ["a"] and out.column(0).to_pylist() == [1, 2, 3, 4]
    with pytest.raises(A.array.InvalidArgumentError) as ei:              # :1609
        K.concat_batches(["a"], [A.RecordBatch(["a"], [i32([1, 2])]), A.RecordBatch(["a"], [utf(["foo", "bar"])])])
    assert str(ei.value) == ("Invalid argument error: It is not possible to concatenate arrays of different data types "
                             "(Int32, Utf8).")


@pytest.mark.parametrize("dt", [A.Utf8, A.LargeUtf8], ids=repr)
def test_concat_strings(ctx, oracle, dt):
    """concat_bytes (concat.rs:355) -> GenericByteBuilder::append_array: offsets shifted per piece, bytes cut to the
    referenced range (sliced pieces), validity only when some piece has nulls."""
    rng = np.random.default_rng(41)
    words = ["", "a", "bc", "longer string " * 3, "ünï"]
    for trial in range(4):
        pieces, dev = [], []
        for k in range(6):
            n = int(rng.integers(0, 900))
            strs = [words[rng.integers(0, len(words))] + str(rng.integers(0, 1000)) for _ in range(n)]
            valid = (rng.random(n) < 0.8) if (k % 2 == 0 and trial != 3) else None
            h = HostArray(dt, strs, valid)
            off = int(rng.integers(0, n // 2 + 1))
            ln = int(rng.integers(0, n - off + 1))
            pieces.append(h.slice(off, ln))
            dev.append(h.to_device(ctx).slice(off, ln))
        exp = oracle.concat(pieces)
        got = K.concat(dev)
        orc.assert_logical_eq(HostArray.from_device(got), exp, f"{dt} trial {trial}")
        assert (got.nulls() is None) == (exp.valid is None) and got.null_count() == exp.null_count


def test_concat_utf8_offset_overflow(ctx):
    """append_array (generic_bytes_builder.rs:185-189): OffsetOverflowError(total bytes) once i32 offsets overflow."""
    n = 1 << 20
    one = A.Array.from_strings(["x" * 1100] * n, ctx=ctx)  # 1.15e9 bytes per piece
    with pytest.raises(A.array.OffsetOverflowError) as e:
        K.concat([one, one])
    assert str(e.value) == f"Offset overflow error: {2 * 1100 * n}"
    fits = K.concat([one.slice(0, n // 2), one.slice(n // 2, n // 2)])
    assert fits.length == n and fits.values.nbytes == 1100 * n


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


# ------------------------------------------------------------ BatchCoalescer
def _check_batches(co, model, tag="", presence_cols=None):
    while True:
        b = co.next_completed_batch()
        if b is None:
            break
        assert model.completed, f"{tag}: device produced an extra batch"
        exp = model.completed.popleft()
        assert b.num_rows() == len(exp[0]), f"{tag}: batch rows {b.num_rows()} != {len(exp[0])}"
        for ci, (c, e) in enumerate(zip(b.columns, exp)):
            check(c, e, tag)
            if presence_cols is not None and ci not in presence_cols:
                continue  # generic columns: a single buffered piece keeps its own (possibly all-valid) buffer
            if not (e.valid is not None and e.null_count == 0):  # bypassed inputs keep their own buffers
                assert_same_nulls_presence(host(c), e, tag)
    assert not model.completed, f"{tag}: device is missing {len(model.completed)} batch(es)"


def test_batch_coalescer_doc_examples(ctx, oracle):
    """coalesce.rs:79-107 (push_batch), :218-228 (with filter), :245-256 (with indices)."""
    from coalesce_model import ModelCoalescer
    i32 = lambda xs: HostArray.from_pylist(xs, A.Int32)
    co = K.BatchCoalescer.new(["a"], [A.Int32], 4, ctx)
    co.push_batch(A.RecordBatch(["a"], [i32([1, 2, 3]).to_device(ctx)]))
    assert co.next_completed_batch() is None
    co.push_batch(A.RecordBatch(["a"], [i32([4, 5]).to_device(ctx)]))
    b = co.next_completed_batch()
    assert b.columns[0].to_pylist() == [1, 2, 3, 4] and co.next_completed_batch() is None
    co.finish_buffered_batch()
    assert co.next_completed_batch().columns[0].to_pylist() == [5] and co.is_empty()

    co = K.BatchCoalescer.new(["a"], [A.Int32], 1000, ctx)
    f = HostArray.from_pylist([True, False, True], A.Boolean).to_device(ctx)
    co.push_batch_with_filter(A.RecordBatch(["a"], [i32([1, 2, 3]).to_device(ctx)]), f)
    co.push_batch_with_filter(A.RecordBatch(["a"], [i32([4, 5, 6]).to_device(ctx)]), f)
    co.finish_buffered_batch()
    assert co.next_completed_batch().columns[0].to_pylist() == [1, 3, 4, 6]

    co = K.BatchCoalescer.new(["a"], [A.Int32], 1000, ctx)
    co.push_batch(A.RecordBatch(["a"], [i32([0, 0, 0]).to_device(ctx)]))
    idx = HostArray.from_pylist([0, 1, 4, 2, 5, 3], A.UInt64).to_device(ctx)
    co.push_batch_with_indices(A.RecordBatch(["a"], [i32([1, 1, 4, 5, 1, 4]).to_device(ctx)]), idx)
    co.finish_buffered_batch()
    assert co.next_completed_batch().columns[0].to_pylist() == [0, 0, 0, 1, 1, 1, 4, 4, 5]


@pytest.mark.parametrize("case", __import__("coalesce_reference_cases").SCENARIOS, ids=lambda c: c[0])
def test_batch_coalescer_reference_scenarios(ctx, case):
    """Every deterministic scenario of the reference's own BatchCoalescer tests (arrow-select/src/coalesce.rs mod tests,
    transcribed in tests/coalesce_reference_cases.py): output batch sizes in order, buffered rows after each step, has_completed,
    the three large-batch bypass rules of set_biggest_coalesce_batch_size, and the output rows = the pushed rows."""
    from coalesce_reference_cases import run_scenario
    name, source, target, limit, steps, tail = case
    co = K.BatchCoalescer.new(["c0"], [A.Int32], target, ctx).with_biggest_coalesce_batch_size(limit)

    class Adapter:
        def push(self, n):
            co.push_batch(A.RecordBatch(["c0"], [HostArray(A.Int32, np.arange(n, dtype=np.int32)).to_device(ctx)], n))

        def drain(self):
            out = []
            while True:
                b = co.next_completed_batch()
                if b is None:
                    return out
                vals = host(b.columns[0]).values[:b.num_rows()]
                out.append([int(x) for x in vals])

        def buffered(self):
            return co.get_buffered_rows()

        def has_completed(self):
            return co.has_completed_batch()

        def finish(self):
            co.finish_buffered_batch()

    run_scenario(Adapter(), steps, tail)
    assert co.is_empty()


@pytest.mark.parametrize("kind", ["utf8", "uint32_nullable"])
def test_batch_coalescer_utf8_split_reference(ctx, kind):
    """test_utf8_split (coalesce.rs:1035-1043) and the nullable uint32_batch its siblings use (:2004-2009): 3000 + 1040 rows,
    every third one NULL (strings "value{i}"), target 1024 -> 1024, 1024, 1024, 968 rows; the output is the input, row for row."""
    def batch(n):
        if kind == "utf8":
            return HostArray.from_pylist([None if i % 3 == 0 else f"value{i}" for i in range(n)], A.Utf8)
        return HostArray.from_pylist([None if i % 3 == 0 else i for i in range(n)], A.UInt32)
    dt = A.Utf8 if kind == "utf8" else A.UInt32
    co = K.BatchCoalescer.new(["c0"], [dt], 1024, ctx)
    pushed = []
    for n in (3000, 1040):
        h = batch(n)
        pushed += h.to_pylist()
        co.push_batch(A.RecordBatch(["c0"], [h.to_device(ctx)], n))
    co.finish_buffered_batch()
    sizes, got = [], []
    while True:
        b = co.next_completed_batch()
        if b is None:
            break
        sizes.append(b.num_rows())
        got += host(b.columns[0]).to_pylist()
    assert sizes == [1024, 1024, 1024, 968] and got == pushed


def test_batch_coalescer_push_batch_with_indices_reference(ctx):
    """test_coalasce_push_batch_with_indices (coalesce.rs:2635-2662): 0 .. 2333 pushed plainly, then 2333 .. 23333 REVERSED
    pushed through the reversing indices: one batch 0 .. 23333."""
    mid, total = 2333, 23333
    co = K.BatchCoalescer.new(["c0"], [A.UInt32], total, ctx)
    co.push_batch(A.RecordBatch(["c0"], [HostArray(A.UInt32, np.arange(mid, dtype=np.uint32)).to_device(ctx)]))
    b2 = HostArray(A.UInt32, np.arange(mid, total, dtype=np.uint32)[::-1].copy())
    idx = HostArray(A.UInt64, np.arange(total - mid, dtype=np.uint64)[::-1].copy())
    co.push_batch_with_indices(A.RecordBatch(["c0"], [b2.to_device(ctx)]), idx.to_device(ctx))
    co.finish_buffered_batch()
    out = co.next_completed_batch()
    assert out.num_rows() == total and np.array_equal(host(out.columns[0]).values, np.arange(total, dtype=np.uint32))
    assert host(out.columns[0]).valid is None and co.next_completed_batch() is None


@pytest.mark.parametrize("limit", [None, 500])
def test_batch_coalescer_fuzz(ctx, oracle, limit):
    """Random push / push_with_filter / push_with_indices sequences: the device coalescer must
    emit the same batches (sizes, order, contents, null-buffer presence) as the model of
    coalesce.rs, including the large-batch bypass cases 1-3 (coalesce.rs:296-420)."""
    from coalesce_model import ModelCoalescer
    rng = np.random.default_rng(1234 + (limit or 0))
    dts = [A.Int64, A.Float64, A.Int32]
    for trial in range(6):
        target = int(rng.choice([7, 64, 1000, 4096]))
        co = K.BatchCoalescer.new(["a", "b", "c"], dts, target, ctx).with_biggest_coalesce_batch_size(limit)
        model = ModelCoalescer(oracle, dts, target)
        model.limit = limit
        for step in range(25):
            n = int(rng.integers(0, 3 * target + 5)) if step % 5 else int(rng.integers(0, 12 * target))
            cols = [HostArray(dt, _rand_values(rng, dt, n), (rng.random(n) < 0.85) if k != 2 else None)
                    for k, dt in enumerate(dts)]
            dcols = [c.to_device(ctx) for c in cols]
            kind = step % 3
            if kind == 0:
                co.push_batch(A.RecordBatch(["a", "b", "c"], dcols))
                model.push(cols)
            elif kind == 1:
                sel = float(rng.choice([0.0, 0.03, 0.3, 1.0]))
                flen = n - int(rng.integers(0, min(n, 3) + 1))
                f = HostArray(A.Boolean, rng.random(flen) < sel, (rng.random(flen) < 0.9) if step % 2 else None)
                co.push_batch_with_filter(A.RecordBatch(["a", "b", "c"], dcols), f.to_device(ctx))
                model.push_with_filter(cols, f)
            elif n:
                idx = HostArray(A.UInt32, rng.integers(0, n, int(rng.integers(0, 2 * n))).astype(np.uint32))
                co.push_batch_with_indices(A.RecordBatch(["a", "b", "c"], dcols), idx.to_device(ctx))
                model.push_with_indices(cols, idx)
            assert co.get_buffered_rows() == model.buffered, f"trial {trial} step {step}"
            _check_batches(co, model, f"trial {trial} step {step}")
        co.finish_buffered_batch()
        model.finish()
        _check_batches(co, model, f"trial {trial} final")
        assert co.is_empty()


@pytest.mark.parametrize("dts", [(A.Int64, A.Float64), (A.Int32, A.Float32, A.UInt32), (A.Int16,),
                                 (A.Int64, A.Float64, A.UInt64, A.Int64, A.Float64, A.UInt64, A.Int64, A.Float64)])
def test_batch_coalescer_same_shape_columns_one_launch(ctx, oracle, dts):
    """All columns nullable and of one width: ONE scatter launch per output batch a filtered push lands in
    (ah_filter_apply_into_acc_cols), windows of the filtered stream when the push straddles output batches — also
    several of them (target far below the selected rows).  Same batches as the model of coalesce.rs:229."""
    from coalesce_model import ModelCoalescer
    rng = np.random.default_rng(99 + len(dts))
    names = [f"c{k}" for k in range(len(dts))]
    for trial, limit in enumerate([None, None, 5000, None]):
        target = int(rng.choice([13, 64, 1000, 4096, 20000]))
        co = K.BatchCoalescer.new(names, list(dts), target, ctx).with_biggest_coalesce_batch_size(limit)
        model = ModelCoalescer(oracle, list(dts), target)
        model.limit = limit
        for step in range(20):
            n = int(rng.integers(0, 6 * target + 9)) if step % 4 else int(rng.integers(4096, 70000))
            cols = [HostArray(dt, _rand_values(rng, dt, n), rng.random(n) < float(rng.choice([0.5, 0.9]))) for dt in dts]
            sel = float(rng.choice([0.02, 0.3, 0.97]))
            flen = n - int(rng.integers(0, min(n, 3) + 1))
            f = HostArray(A.Boolean, rng.random(flen) < sel, (rng.random(flen) < 0.9) if step % 2 else None)
            off = int(rng.integers(0, 70)) if n > 200 else 0  # sliced inputs: bit offsets in mask and validity
            dcols = [c.to_device(ctx) for c in cols]
            df = f.to_device(ctx)
            if off and flen > off:
                dcols = [c.slice(off, n - off) for c in dcols]
                df = df.slice(off, flen - off)
                cols = [c.slice(off, n - off) for c in cols]
                f = f.slice(off, flen - off)
            co.push_batch_with_filter(A.RecordBatch(names, dcols), df)
            model.push_with_filter(cols, f)
            assert co.get_buffered_rows() == model.buffered, f"trial {trial} step {step}"
            _check_batches(co, model, f"same-shape trial {trial} step {step}")
        co.finish_buffered_batch()
        model.finish()
        _check_batches(co, model, f"same-shape trial {trial} final")
        assert co.is_empty()


def test_batch_coalescer_grouped_pushes_equal_single_pushes(ctx, oracle):
    """ah_coalescer_push_batches_with_filters (n filtered pushes, one wait for the n counts) must produce exactly the
    output batches of n single pushes — compared with the model of coalesce.rs, bypass limit included."""
    from coalesce_model import ModelCoalescer
    rng = np.random.default_rng(4242)
    dts = [A.Int64, A.Float64]
    for trial, limit in enumerate([None, 300, None, 64]):
        target = int(rng.choice([50, 512, 4096]))
        co = K.BatchCoalescer.new(["a", "b"], dts, target, ctx).with_biggest_coalesce_batch_size(limit)
        model = ModelCoalescer(oracle, dts, target)
        model.limit = limit
        pending = []
        for step in range(40):
            n = int(rng.integers(0, 3 * target + 5))
            cols = [HostArray(dt, _rand_values(rng, dt, n), (rng.random(n) < 0.85) if k == 0 else None) for k, dt in enumerate(dts)]
            flen = n - int(rng.integers(0, min(n, 3) + 1))
            f = HostArray(A.Boolean, rng.random(flen) < float(rng.choice([0.0, 0.05, 0.4, 1.0])),
                          (rng.random(flen) < 0.9) if step % 3 == 0 else None)
            pending.append((A.RecordBatch(["a", "b"], [c.to_device(ctx) for c in cols]), f.to_device(ctx)))
            model.push_with_filter(cols, f)
            if len(pending) >= int(rng.integers(1, 8)) or step == 39:
                co.push_batches_with_filters(pending)
                pending = []
                assert co.get_buffered_rows() == model.buffered, f"trial {trial} step {step}"
                _check_batches(co, model, f"grouped trial {trial} step {step}")
        co.finish_buffered_batch()
        model.finish()
        _check_batches(co, model, f"grouped trial {trial} final")
        assert co.is_empty()


@pytest.mark.parametrize("schema", ["i64_f64", "i32x3", "mixed_generic", "limit"])
def test_batch_coalescer_pipelined_pushes_equal_the_model(ctx, oracle, schema):
    """push_batches_with_filters_begin / _end (round 4): the counts of group g + 1 are enqueued BEFORE group g is appended,
    two groups in flight.  The output batches must be exactly those of the coalesce.rs model fed the same pairs in the
    same order — fused multi-scatter shapes, generic columns, a bypass limit, groups of one (the pipelined single push),
    empty predicates and predicates shorter than their batch."""
    from coalesce_model import ModelCoalescer
    import zlib
    rng = np.random.default_rng(zlib.crc32(("pipe-" + schema).encode()))
    dts = {"i64_f64": [A.Int64, A.Float64], "i32x3": [A.Int32, A.Float32, A.UInt32], "mixed_generic": [A.Utf8, A.Int64, A.Boolean],
           "limit": [A.Int64, A.Float64]}[schema]
    names = [f"c{i}" for i in range(len(dts))]
    limit = 300 if schema == "limit" else None

    def col(dt, n):
        valid = rng.random(n) < 0.85
        if dt in (A.Utf8, A.LargeUtf8):
            return HostArray(dt, ["v" * int(rng.integers(0, 5)) + str(rng.integers(0, 999)) for _ in range(n)], valid)
        return HostArray(dt, _rand_values(rng, dt, n), valid)

    for trial in range(3):
        target = int(rng.choice([60, 1000, 30_000]))
        co = K.BatchCoalescer.new(names, dts, target, ctx).with_biggest_coalesce_batch_size(limit)
        model = ModelCoalescer(oracle, dts, target)
        model.limit = limit
        groups, hosts = [], []
        for _ in range(10):
            g, hg = [], []
            for _ in range(int(rng.integers(1, 9))):
                n = int(rng.integers(1, 4000))
                cols = [col(dt, n) for dt in dts]
                flen = n - int(rng.integers(0, min(n, 3) + 1))
                f = HostArray(A.Boolean, rng.random(flen) < float(rng.choice([0.0, 0.03, 0.4, 1.0])), (rng.random(flen) < 0.9) if rng.random() < 0.3 else None)
                g.append((A.RecordBatch(names, [c.to_device(ctx) for c in cols]), f.to_device(ctx)))
                hg.append((cols, f))
            groups.append(g)
            hosts.append(hg)
        pending, pending_host = co.push_batches_with_filters_begin(groups[0]), hosts[0]
        for gi in range(1, len(groups) + 1):
            nxt = co.push_batches_with_filters_begin(groups[gi]) if gi < len(groups) else None
            pending.end()
            for cols, f in pending_host:
                model.push_with_filter(cols, f)
            assert co.get_buffered_rows() == model.buffered, f"{schema} trial {trial} group {gi - 1}"
            _check_batches(co, model, f"pipelined {schema} trial {trial} group {gi - 1}", presence_cols=[] if schema == "mixed_generic" else None)
            pending, pending_host = nxt, (hosts[gi] if gi < len(groups) else None)
        co.finish_buffered_batch()
        model.finish()
        _check_batches(co, model, f"pipelined {schema} trial {trial} final", presence_cols=[] if schema == "mixed_generic" else None)
        assert co.is_empty()


@pytest.mark.parametrize("form", ["pipelined", "one_call"])
@pytest.mark.parametrize("seed", range(4))
def test_batch_coalescer_cut_batches_walk_only_their_tiles(ctx, oracle, seed, form):
    """A pushed batch that an output-batch boundary cuts is appended by one scatter segment per output batch; each segment
    launches only the tiles its window of the filtered stream can lie in, bounded on the host from the predicate's quantile
    prefixes (csrc/filter.hip window_tiles; recorded by the pipelined pushes).  Batches of 0.2-2.5 M rows (4 ... 39 count
    groups, so the bounds really cut), targets that cut every batch several times, and selections that are uniform, empty
    at one end, or packed into a narrow band — a bound one tile short loses rows, and the coalesce.rs model notices."""
    from coalesce_model import ModelCoalescer
    rng = np.random.default_rng(9100 + seed)
    dts = [[A.Int64, A.Float64], [A.Int32, A.Float32, A.UInt32], [A.Int64], [A.Int16, A.Int16]][seed]
    names = [f"c{i}" for i in range(len(dts))]
    target = [70_001, 333_333, 20_000, 150_000][seed]
    co = K.BatchCoalescer.new(names, dts, target, ctx)
    model = ModelCoalescer(oracle, dts, target)

    def predicate(n):
        shape = int(rng.integers(0, 5))
        m = np.zeros(n, dtype=bool)
        if shape == 0:
            m = rng.random(n) < float(rng.choice([0.05, 0.3, 0.8]))
        elif shape == 1:  # nothing selected in the first part
            cut = int(rng.integers(0, n))
            m[cut:] = rng.random(n - cut) < 0.5
        elif shape == 2:  # nothing selected in the last part
            cut = int(rng.integers(1, n + 1))
            m[:cut] = rng.random(cut) < 0.5
        elif shape == 3:  # a narrow dense band
            a = int(rng.integers(0, n))
            m[a:min(n, a + int(rng.integers(1, 200_000)))] = True
        else:  # bands at count-group boundaries (65 536 rows)
            for g in rng.integers(0, max(1, n // 65536), size=3):
                lo = int(g) * 65536
                m[max(0, lo - 3):min(n, lo + 5)] = True
        return m

    groups, hosts = [], []
    for _ in range(4):
        g, hg = [], []
        for _ in range(int(rng.integers(1, 5))):
            n = int(rng.integers(200_000, 2_500_000))
            cols = [HostArray(dt, _rand_values(rng, dt, n), rng.random(n) < 0.85) for dt in dts]
            f = HostArray(A.Boolean, predicate(n), (rng.random(n) < 0.95) if rng.random() < 0.3 else None)
            g.append((A.RecordBatch(names, [c.to_device(ctx) for c in cols]), f.to_device(ctx)))
            hg.append((cols, f))
        groups.append(g)
        hosts.append(hg)
    if form == "one_call":  # ah_coalescer_push_batches_with_filters: the counts through the coalescer's own words as well
        for gi, g in enumerate(groups):
            co.push_batches_with_filters(g)
            for cols, f in hosts[gi]:
                model.push_with_filter(cols, f)
            assert co.get_buffered_rows() == model.buffered, f"seed {seed} group {gi}"
            _check_batches(co, model, f"cut batches (one call) seed {seed} group {gi}")
    else:
        pending, pending_host = co.push_batches_with_filters_begin(groups[0]), hosts[0]
        for gi in range(1, len(groups) + 1):
            nxt = co.push_batches_with_filters_begin(groups[gi]) if gi < len(groups) else None
            pending.end()
            for cols, f in pending_host:
                model.push_with_filter(cols, f)
            assert co.get_buffered_rows() == model.buffered, f"seed {seed} group {gi - 1}"
            _check_batches(co, model, f"cut batches seed {seed} group {gi - 1}")
            pending, pending_host = nxt, (hosts[gi] if gi < len(groups) else None)
    co.finish_buffered_batch()
    model.finish()
    _check_batches(co, model, f"cut batches seed {seed} final")
    assert co.is_empty()


@pytest.mark.parametrize("seed", range(6))
def test_batch_coalescer_grouped_pushes_one_launch_per_window(ctx, oracle, seed):
    """Same-width, all-nullable columns and no bypass limit: the grouped push scatters the batches that land in one
    output window with ONE multi-batch launch (csrc/filter.hip filter_scatter_multi_kernel: up to 8 (batch, window)
    segments, windows that straddle output batches, more than 8 batches per window, an all-selected batch in between
    that takes the copy path).  The output batches must be exactly those of the coalesce.rs model."""
    from coalesce_model import ModelCoalescer
    rng = np.random.default_rng(5150 + seed)
    dts = [[A.Int64, A.Float64], [A.Int32, A.Float32, A.UInt32], [A.Int64], [A.Int16, A.Int16]][seed % 4]
    names = [f"c{i}" for i in range(len(dts))]
    target = [3000, 200, 70_000, 1000, 5, 40_000][seed]
    co = K.BatchCoalescer.new(names, dts, target, ctx)
    model = ModelCoalescer(oracle, dts, target)
    pending = []
    for step in range(60):
        n = int(rng.integers(1, [2000, 5000, 20_000][seed % 3]))
        cols = [HostArray(dt, _rand_values(rng, dt, n), rng.random(n) < 0.85) for dt in dts]
        flen = n - int(rng.integers(0, min(n, 3) + 1))
        psel = float(rng.choice([0.0, 0.02, 0.3, 0.9, 1.0]))
        f = HostArray(A.Boolean, rng.random(flen) < psel, (rng.random(flen) < 0.9) if step % 4 == 0 else None)
        pending.append((A.RecordBatch(names, [c.to_device(ctx) for c in cols]), f.to_device(ctx)))
        model.push_with_filter(cols, f)
        if len(pending) >= int(rng.integers(2, 20)) or step == 59:
            co.push_batches_with_filters(pending)
            pending = []
            assert co.get_buffered_rows() == model.buffered, f"seed {seed} step {step}"
            _check_batches(co, model, f"multi seed {seed} step {step}")
    co.finish_buffered_batch()
    model.finish()
    _check_batches(co, model, f"multi seed {seed} final")
    assert co.is_empty()


def _expected_stream(host_cols, keep, target):
    """What BatchCoalescer must emit for a stream of filtered pushes (coalesce.rs:229-330): the selected rows of the whole
    stream, in order, cut into `target`-row batches — an independent numpy statement of the same thing the coalesce.rs model
    computes push by push.  -> list of batches, each a list of HostArray (validity None iff the batch holds no null)."""
    sel = [HostArray(c.data_type, np.asarray(c.values)[keep], None if c.valid is None else np.asarray(c.valid)[keep]) for c in host_cols]
    k = int(keep.sum())
    out = []
    for lo in range(0, k, target):
        n = min(target, k - lo)
        batch = []
        for c in sel:
            piece = c.slice(lo, n)
            nulls = 0 if piece.valid is None else int((~piece.valid).sum())
            batch.append(HostArray(piece.data_type, piece.values, piece.valid if nulls else None))
        out.append(batch)
    return out


def _drain_and_check(co, expected, tag):
    """every completed batch (bulk fetch) against the head of `expected`; -> batches consumed"""
    got = co.next_completed_batches()
    assert len(got) <= len(expected), f"{tag}: device produced {len(got)} batches, expected at most {len(expected)}"
    for j, b in enumerate(got):
        exp = expected[j]
        assert b.num_rows() == len(exp[0]), f"{tag}: batch {j} rows {b.num_rows()} != {len(exp[0])}"
        for c, e in zip(b.columns, exp):
            check(c, e, f"{tag} batch {j}")
            assert_same_nulls_presence(host(c), e, f"{tag} batch {j}")
    del expected[:len(got)]
    return len(got)


@pytest.mark.parametrize("shape", ["i64_f64_8192x1024", "mixed_widths_ragged", "sparse_0.001", "dense_0.6_no_validity"])
def test_batch_coalescer_slab_push_reference_operating_point(ctx, shape):
    """the reference's own operating point — 8192-row input batches, target 8192 (coalesce.rs:172-173,
    arrow/benches/coalesce_kernels.rs:34) — through the slab push: ANY number of batches per call, one count + one scatter
    launch per destination, output batches carved from one slab.  1 024 batches in one call, then pipelined groups of 256
    (begin / end, two in flight), then single pushes and grouped pushes alternating (the slab continues the in-progress batch
    of the round-4 paths and vice versa).  Every output batch against an independent numpy statement of the stream."""
    rng = np.random.default_rng(_seed("slab-" + shape))
    if shape == "i64_f64_8192x1024":
        dts, target, sizes, p_sel = [A.Int64, A.Float64], 8192, [8192] * 1024, 0.1
    elif shape == "mixed_widths_ragged":
        dts, target, p_sel = [A.Int32, A.Int64, A.Int8, A.Float64, A.Int16], 4096, 0.13
        sizes = [int(x) for x in rng.integers(0, 20000, 400)] + [70000, 1, 0, 131072 + 5]
    elif shape == "sparse_0.001":
        dts, target, sizes, p_sel = [A.Int64, A.Float64], 8192, [8192] * 700 + [65536 * 2 + 17] * 3, 0.001
    else:
        dts, target, sizes, p_sel = [A.Float64, A.Int64], 1 << 16, [8192] * 300, 0.6
    total = sum(sizes)
    names = [f"c{k}" for k in range(len(dts))]
    with_valid = shape != "dense_0.6_no_validity"
    hcols = [HostArray(dt, _rand_values(rng, dt, total), (rng.random(total) < 0.9) if (with_valid or k == 1) and k != 2 else None)
             for k, dt in enumerate(dts)]
    fbits = rng.random(total) < p_sel
    fvalid = (rng.random(total) < 0.95) if shape != "sparse_0.001" else None
    hf = HostArray(A.Boolean, fbits, fvalid)
    keep = fbits & (fvalid if fvalid is not None else True)
    dcols = [c.to_device(ctx) for c in hcols]
    df = hf.to_device(ctx)
    pairs, off = [], 0
    for n in sizes:  # zero-copy device slices: ragged sizes give every bit offset
        pairs.append((A.RecordBatch(names, [c.slice(off, n) for c in dcols], n), df.slice(off, n)))
        off += n
    expected_all = _expected_stream(hcols, keep, target)

    # (a) everything in ONE call
    co = K.BatchCoalescer.new(names, dts, target, ctx)
    expected = [list(b) for b in expected_all]
    co.push_batches_with_filters(pairs)
    assert co.get_buffered_rows() == int(keep.sum()) % target
    _drain_and_check(co, expected, f"{shape} one call")
    co.finish_buffered_batch()
    _drain_and_check(co, expected, f"{shape} one call, tail")
    assert not expected and co.is_empty()

    # (b) pipelined groups: begin of group g + 1 before end of group g
    co = K.BatchCoalescer.new(names, dts, target, ctx)
    expected = [list(b) for b in expected_all]
    gsz = 256
    groups = [pairs[i:i + gsz] for i in range(0, len(pairs), gsz)]
    pending = co.push_batches_with_filters_begin(groups[0])
    for g in groups[1:]:
        nxt = co.push_batches_with_filters_begin(g)
        pending.end()
        pending = nxt
        _drain_and_check(co, expected, f"{shape} pipelined")
    pending.end()
    co.finish_buffered_batch()
    _drain_and_check(co, expected, f"{shape} pipelined, tail")
    assert not expected and co.is_empty()

    # (c) single pushes (round-4 paths: speculative scatter into the in-progress buffers) and grouped pushes alternating,
    # an aborted group in between (its batches are NOT appended)
    co = K.BatchCoalescer.new(names, dts, target, ctx)
    i, step, kept = 0, 0, np.zeros(total, dtype=bool)
    starts = np.concatenate([[0], np.cumsum(sizes)])
    while i < len(pairs):
        if step % 3 == 0:
            co.push_batch_with_filter(*pairs[i])
            kept[starts[i]:starts[i + 1]] = True
            i += 1
        elif step % 7 == 4:
            h = co.push_batches_with_filters_begin(pairs[i:i + 5])
            h.abort()
            i += 5
        else:
            m = int(rng.integers(2, 90))
            co.push_batches_with_filters(pairs[i:i + m])
            kept[starts[i]:starts[min(i + m, len(pairs))]] = True
            i += m
        step += 1
    co.finish_buffered_batch()
    expected = _expected_stream(hcols, keep & kept, target)
    _drain_and_check(co, expected, f"{shape} alternating")
    assert not expected and co.is_empty()


def test_batch_coalescer_slab_cap_bounds_retention(ctx):
    """every output batch of a slab push is a slice of a pool block that lives until its LAST slice is released, so
    one kept batch used to pin the push's whole output; the reference allocates each batch on its own (coalesce.rs:560-600).
    Slabs are capped (AH_COALESCE_SLAB_BYTES; here 1 MiB so that a small push spans dozens of slabs, with the in-progress batch
    topped up first and a tail left over): every batch is still the reference's, and after releasing all batches but one the
    context holds one slab's worth of output, not the push's."""
    import gc
    import os
    rng = np.random.default_rng(77)
    dts, target, sizes = [A.Int64, A.Float64], 8192, [8192] * 512
    total = sum(sizes)
    hcols = [HostArray(dt, _rand_values(rng, dt, total), rng.random(total) < 0.9) for dt in dts]
    hf = HostArray(A.Boolean, rng.random(total) < 0.5)
    keep = np.asarray(hf.values)
    dcols, df = [c.to_device(ctx) for c in hcols], hf.to_device(ctx)
    names = ["a", "b"]
    pairs, off = [], 0
    for n in sizes:
        pairs.append((A.RecordBatch(names, [c.slice(off, n) for c in dcols], n), df.slice(off, n)))
        off += n
    os.environ["AH_COALESCE_SLAB_BYTES"] = str(1 << 20)
    try:
        co = K.BatchCoalescer.new(names, dts, target, ctx)
        expected = _expected_stream(hcols, keep, target)
        co.push_batches_with_filters(pairs[:3])   # leaves an in-progress batch behind: the next push tops it up first
        co.push_batches_with_filters(pairs[3:])
        got = co.next_completed_batches()
        co.finish_buffered_batch()
        got += co.next_completed_batches()
        assert len(got) == len(expected)
        for j, (b, exp) in enumerate(zip(got, expected)):
            for c, e in zip(b.columns, exp):
                check(c, e, f"capped slabs batch {j}")
        out_bytes = int(keep.sum()) * 16
        assert out_bytes > 30 << 20
        kept = got[len(got) // 2]
        del got, b, c, co
        gc.collect()
        base = ctx.memory_stats()["live_bytes"]
        del kept
        gc.collect()
        held_by_one_batch = base - ctx.memory_stats()["live_bytes"]
        # one slab: at most the cap (rounded up to whole output batches and the pool's 1 MiB granule), never the whole push
        assert 0 < held_by_one_batch <= 3 << 20, held_by_one_batch
    finally:
        os.environ.pop("AH_COALESCE_SLAB_BYTES", None)


def test_batch_coalescer_more_than_64_pairs_when_the_slab_path_declines(ctx):
    """`push_batches_with_filters_begin` with more than 64 pairs raised whenever the slab path declined for a reason
    Python could not see (deferred context, two pushes already in flight, AH_COALESCE_SLAB=0).  The library now counts groups of
    64 itself: 100 pairs in deferred mode, and three begins in flight, give the reference's stream."""
    rng = np.random.default_rng(78)
    dts, target, sizes = [A.Int64, A.Int32], 8192, [int(x) for x in rng.integers(100, 9000, 300)]
    total = sum(sizes)
    hcols = [HostArray(dt, _rand_values(rng, dt, total), rng.random(total) < 0.9) for dt in dts]
    hf = HostArray(A.Boolean, rng.random(total) < 0.3, rng.random(total) < 0.95)
    keep = np.asarray(hf.values) & np.asarray(hf.valid)
    dcols, df = [c.to_device(ctx) for c in hcols], hf.to_device(ctx)
    names = ["a", "b"]
    pairs, off = [], 0
    for n in sizes:
        pairs.append((A.RecordBatch(names, [c.slice(off, n) for c in dcols], n), df.slice(off, n)))
        off += n
    # (a) a deferred context, 100 pairs per begin
    co = K.BatchCoalescer.new(names, dts, target, ctx)
    expected = _expected_stream(hcols, keep, target)
    with ctx.deferred_mode():
        for g in range(0, 300, 100):
            co.push_batches_with_filters_begin(pairs[g:g + 100]).end()
    ctx.synchronize()
    _drain_and_check(co, expected, "deferred, 100 pairs")
    co.finish_buffered_batch()
    _drain_and_check(co, expected, "deferred, 100 pairs, tail")
    assert not expected
    # (b) three begins in flight (the slab path takes two), ended in order
    co = K.BatchCoalescer.new(names, dts, target, ctx)
    expected = _expected_stream(hcols, keep, target)
    h = [co.push_batches_with_filters_begin(pairs[g:g + 100]) for g in range(0, 300, 100)]
    for x in h:
        x.end()
    _drain_and_check(co, expected, "three begins in flight")
    co.finish_buffered_batch()
    _drain_and_check(co, expected, "three begins in flight, tail")
    assert not expected
    # (c) a begun push that is never ended, collected AFTER its coalescer: aborted by the coalescer's own finalizer
    import gc
    co = K.BatchCoalescer.new(names, dts, target, ctx)
    pend = co.push_batches_with_filters_begin(pairs[:70])
    pend.co = None  # (the pending push normally keeps its coalescer alive: drop that edge to force the order)
    del co
    gc.collect()
    del pend
    gc.collect()
    assert len(K.filter(dcols[0], df)) == int(keep.sum())  # the context is still healthy


@pytest.mark.parametrize("grouped", [False, True])
def test_batch_coalescer_owned_large_batch_does_not_alias_its_input(ctx, grouped):
    """Round 6, found by the guard-page run: a filtered push whose predicate is SHORTER than its batch and selects every one of
    its rows takes filter's `All` strategy — a borrowed slice of the caller's buffers (filter.rs:546) — and with a bypass limit
    below the row count that slice went into the completed queue as if the coalescer owned it (large-batch cases 1 / 2,
    coalesce.rs:330-360): the caller, told the batch was NOT bypassed, may release its input, and the completed batch then read
    released memory.  Here the input is released and its pool blocks are recycled with other bytes before the batch is fetched."""
    import gc
    rng = np.random.default_rng(79)
    n, flen = 12199, 12197
    vals = [rng.integers(-2**62, 2**62, n, dtype=np.int64), rng.normal(size=n)]
    valid = rng.random(n) < 0.85
    co = K.BatchCoalescer.new(["a", "b"], [A.Int64, A.Float64], 512, ctx).with_biggest_coalesce_batch_size(64)
    cols = [HostArray(A.Int64, vals[0], valid).to_device(ctx), HostArray(A.Float64, vals[1]).to_device(ctx)]
    f = HostArray(A.Boolean, np.ones(flen, dtype=bool)).to_device(ctx)
    rb = A.RecordBatch(["a", "b"], cols)
    if grouped:
        co.push_batches_with_filters([(rb, f)])
    else:
        co.push_batch_with_filter(rb, f)
    del rb, cols, f
    gc.collect()
    junk = [HostArray(A.Int64, np.full(n, 0x5A5A5A5A5A5A5A5A, dtype=np.int64), np.zeros(n, dtype=bool)).to_device(ctx) for _ in range(4)]
    got = co.next_completed_batch()
    assert got is not None and got.num_rows() == flen
    check(got.columns[0], HostArray(A.Int64, vals[0][:flen], valid[:flen]), "owned large batch, Int64")
    check(got.columns[1], HostArray(A.Float64, vals[1][:flen]), "owned large batch, Float64")
    del junk


def test_batch_coalescer_generic_columns(ctx, oracle):
    """Boolean / Utf8 / LargeUtf8 columns go through GenericInProgressArray (coalesce/generic.rs): buffered
    slices and filtered arrays, `concat` on finish — next to a primitive column on the fused path; same batch
    sequence as the model."""
    from coalesce_model import ModelCoalescer
    rng = np.random.default_rng(77)
    dts = [A.Utf8, A.Int64, A.Boolean, A.LargeUtf8]
    names = ["s", "i", "b", "ls"]
    words = ["", "x", "yz", "a longer value "]

    def col(dt, n, k):
        valid = (rng.random(n) < 0.85) if k % 2 == 0 else None
        if dt in (A.Utf8, A.LargeUtf8):
            return HostArray(dt, [words[rng.integers(0, 4)] + str(rng.integers(0, 99)) for _ in range(n)], valid)
        return HostArray(dt, _rand_values(rng, dt, n), valid)

    for trial in range(4):
        target = int(rng.choice([5, 64, 777]))
        co = K.BatchCoalescer.new(names, dts, target, ctx)
        model = ModelCoalescer(oracle, dts, target)
        for step in range(18):
            n = int(rng.integers(0, 3 * target + 5))
            cols = [col(dt, n, k) for k, dt in enumerate(dts)]
            dcols = [c.to_device(ctx) for c in cols]
            if step % 3 == 0:
                co.push_batch(A.RecordBatch(names, dcols))
                model.push(cols)
            elif step % 3 == 1:
                f = HostArray(A.Boolean, rng.random(n) < float(rng.choice([0.0, 0.1, 0.6, 1.0])))
                co.push_batch_with_filter(A.RecordBatch(names, dcols), f.to_device(ctx))
                model.push_with_filter(cols, f)
            elif n:
                idx = HostArray(A.UInt32, rng.integers(0, n, int(rng.integers(0, 2 * n))).astype(np.uint32))
                co.push_batch_with_indices(A.RecordBatch(names, dcols), idx.to_device(ctx))
                model.push_with_indices(cols, idx)
            assert co.get_buffered_rows() == model.buffered
            _check_batches(co, model, f"generic trial {trial} step {step}", presence_cols=[1])
        co.finish_buffered_batch()
        model.finish()
        _check_batches(co, model, f"generic trial {trial} final", presence_cols=[1])


@pytest.mark.parametrize("schema", ["bool", "utf8", "large_utf8", "mixed", "bool_i64"])
def test_batch_coalescer_grouped_pushes_generic_schemas(ctx, oracle, schema):
    """a grouped push (n > 1, no bypass limit) of a schema whose columns are ALL Boolean, or ALL
    Utf8 / LargeUtf8, all nullable, has "one width" (0 / -1) and used to be sent to the multi-batch scatter, which has no
    in-progress buffers for generic columns.  Such schemas take the per-batch path; the output batches are those of the
    coalesce.rs model, and the coalescer stays usable."""
    from coalesce_model import ModelCoalescer
    import zlib
    rng = np.random.default_rng(zlib.crc32(schema.encode()))
    dts = {"bool": [A.Boolean, A.Boolean], "utf8": [A.Utf8], "large_utf8": [A.LargeUtf8, A.LargeUtf8],
           "mixed": [A.Utf8, A.Boolean, A.LargeUtf8], "bool_i64": [A.Boolean, A.Int64]}[schema]
    names = [f"c{i}" for i in range(len(dts))]
    words = ["", "q", "rs", "a somewhat longer value "]

    def col(dt, n):
        valid = rng.random(n) < 0.8  # every column nullable: the shape the fused path keys on
        if dt in (A.Utf8, A.LargeUtf8):
            return HostArray(dt, [words[rng.integers(0, 4)] + str(rng.integers(0, 99)) for _ in range(n)], valid)
        return HostArray(dt, _rand_values(rng, dt, n), valid)

    for trial in range(3):
        target = int(rng.choice([7, 100, 1500]))
        co = K.BatchCoalescer.new(names, dts, target, ctx)
        model = ModelCoalescer(oracle, dts, target)
        pending = []
        for step in range(24):
            n = int(rng.integers(1, 2 * target + 9))
            cols = [col(dt, n) for dt in dts]
            f = HostArray(A.Boolean, rng.random(n) < float(rng.choice([0.0, 0.1, 0.6, 1.0])), (rng.random(n) < 0.9) if step % 4 == 0 else None)
            pending.append((A.RecordBatch(names, [c.to_device(ctx) for c in cols]), f.to_device(ctx)))
            model.push_with_filter(cols, f)
            if len(pending) >= int(rng.integers(2, 7)) or step == 23:
                co.push_batches_with_filters(pending)
                pending = []
                assert co.get_buffered_rows() == model.buffered, f"{schema} trial {trial} step {step}"
                _check_batches(co, model, f"grouped generic {schema} trial {trial} step {step}", presence_cols=[])
        co.finish_buffered_batch()
        model.finish()
        _check_batches(co, model, f"grouped generic {schema} trial {trial} final", presence_cols=[])
        assert co.is_empty()


@pytest.mark.parametrize("dt", [A.Utf8View, A.BinaryView], ids=str)
def test_batch_coalescer_view_columns(ctx, oracle, dt):
    """Utf8View / BinaryView columns through the native coalescer (InProgressByteViewArray, coalesce/byte_view.rs:39;
    VERDICT r03 missing #4): inputs with MANY data buffers each (so buffer indices really are shifted), pushed plain,
    filtered (single, grouped, pipelined) and by indices, straddling output batches, next to an Int64 column, with and
    without a bypass limit — every output batch must hold exactly the rows of the coalesce.rs model, as logical values
    (views are decoded through the buffer list the mirror attaches: a wrong index shift reads the wrong bytes)."""
    from coalesce_model import ModelCoalescer
    import zlib
    rng = np.random.default_rng(zlib.crc32(dt.name.encode()))
    dts = [dt, A.Int64]
    names = ["v", "i"]

    def make(n):
        items = _view_items(rng, n)
        # the MODEL's column is LargeUtf8 (the CPU oracle has no view layout; the logical values are the same strings)
        host_v = HostArray(A.LargeUtf8, [x if x is not None else "" for x in items],
                           np.array([x is not None for x in items]) if any(x is None for x in items) else None)
        dev_items = items if dt == A.Utf8View else [None if x is None else x.encode() for x in items]
        dev_v = A.Array.from_string_views(dev_items, dt, ctx=ctx, block_size=int(rng.choice([64, 256, 8192])))
        host_i = HostArray(A.Int64, rng.integers(-99, 99, n), rng.random(n) < 0.9)
        return [host_v, host_i], A.RecordBatch(names, [dev_v, host_i.to_device(ctx)], n)

    def same(co, model, tag):
        while True:
            b = co.next_completed_batch()
            if b is None:
                break
            assert model.completed, f"{tag}: extra batch"
            exp = model.completed.popleft()
            assert b.num_rows() == len(exp[0]), tag
            got_v = [x.decode() if isinstance(x, bytes) else x for x in b.columns[0].to_pylist()]
            want_v = [x if (exp[0].valid is None or exp[0].valid[k]) else None for k, x in enumerate(exp[0].values)]
            assert got_v == want_v, f"{tag}: view column"
            check(b.columns[1], exp[1], tag)
        assert not model.completed, f"{tag}: missing batches"

    for trial, limit in enumerate([None, None, 400]):
        target = int(rng.choice([37, 500, 2000]))
        co = K.BatchCoalescer.new(names, dts, target, ctx).with_biggest_coalesce_batch_size(limit)
        model = ModelCoalescer(oracle, [A.LargeUtf8, A.Int64], target)
        model.limit = limit
        for step in range(28):
            n = int(rng.integers(1, 3 * target + 5))
            hosts, batch = make(n)
            kind = step % 5
            if kind == 0:
                co.push_batch(batch)
                model.push(hosts)
            elif kind == 1:
                f = HostArray(A.Boolean, rng.random(n) < float(rng.choice([0.0, 0.1, 0.6, 1.0])), (rng.random(n) < 0.9) if step % 2 else None)
                co.push_batch_with_filter(batch, f.to_device(ctx))
                model.push_with_filter(hosts, f)
            elif kind == 2:
                idx = HostArray(A.UInt32, rng.integers(0, n, int(rng.integers(0, 2 * n))).astype(np.uint32))
                co.push_batch_with_indices(batch, idx.to_device(ctx))
                model.push_with_indices(hosts, idx)
            else:  # grouped (3) / pipelined grouped (4): two or three filtered batches in one call
                pairs, hp = [], []
                for _ in range(int(rng.integers(2, 4))):
                    m_ = int(rng.integers(1, 2 * target))
                    h2, b2 = make(m_)
                    f = HostArray(A.Boolean, rng.random(m_) < float(rng.choice([0.05, 0.5, 1.0])))
                    pairs.append((b2, f.to_device(ctx)))
                    hp.append((h2, f))
                if kind == 3:
                    co.push_batches_with_filters(pairs)
                else:
                    co.push_batches_with_filters_begin(pairs).end()
                for h2, f in hp:
                    model.push_with_filter(h2, f)
            assert co.get_buffered_rows() == model.buffered, f"{dt} trial {trial} step {step}"
            same(co, model, f"{dt} trial {trial} step {step} kind {kind}")
        co.finish_buffered_batch()
        model.finish()
        same(co, model, f"{dt} trial {trial} final")
        assert co.is_empty()


# ------------------------------------------------- RCCL reassembly, 1 rank
def test_communicator_world1_nccl(ctx, oracle, tmp_path):
    """Plumbing of the RCCL path with a single rank (the box has one GPU): count exchange,
    zero-copy hand-off of arrow_hip buffers to torch, bitmap merge at a bit offset."""
    import os
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from arrow_rs_amd import distributed as D
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    # rendezvous through a FILE store: "bind port 0, close, listen on it again" lost the port to another process once in
    # six full-suite runs (EADDRINUSE), and with `-x` one such flake ends the driver's whole GPU run
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rendezvous", rank=0, world_size=1)
    try:
        comm = D.Communicator(ctx, dist)
        rng = np.random.default_rng(4)
        n = 100_001
        h = HostArray(A.Int64, rng.integers(-2**62, 2**62, n, dtype=np.int64), rng.random(n) < 0.9)
        m = HostArray(A.Boolean, rng.random(n) < 0.3)
        f = K.filter(h.to_device(ctx), m.to_device(ctx))
        g = comm.all_gatherv(f)
        check(g, oracle.filter(h, m), "all_gatherv world=1")
        nonull = HostArray(A.Float64, rng.normal(size=1000))
        g2 = comm.all_gatherv(nonull.to_device(ctx))
        assert g2.nulls() is None
        check(g2, nonull)
    finally:
        dist.destroy_process_group()


def test_cpp_host_mirror(ctx, tmp_path):
    """The C++ host mirror (include/arrow_hip.hpp) over the same C ABI, written like the
    reference's own unit tests (tests/cpp/test_host_mirror.cpp)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_host_mirror")
    libdir = os.path.join(root, "arrow-rs_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "test_host_mirror.cpp"), "-L" + libdir,
                           "-larrow_hip", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "CPP_HOST_MIRROR_OK" in out.stdout, out.stdout + out.stderr


# -------------------------------------- boundary: allocator hook, streams, threads
def test_output_allocator_hook(ctx, oracle):
    """ah_context_set_allocator: results land in memory the HOST owns (the Rust shim's
    Buffer::from_custom_allocation hook, arrow-buffer/src/buffer/immutable.rs:170-176)."""
    import ctypes as C
    from arrow_rs_amd import _lib as L
    c2 = A.Context(0)
    arena = c2.alloc(64 << 20)
    state = {"top": 0, "live": {}, "allocs": 0, "frees": 0}

    @L.ALLOC_FN
    def alloc(user, nbytes):
        off = (state["top"] + 255) & ~255
        if off + nbytes > arena.nbytes:
            return None
        state["top"] = off + nbytes
        state["live"][arena.ptr + off] = nbytes
        state["allocs"] += 1
        return arena.ptr + off

    @L.FREE_FN
    def free(user, ptr, nbytes):
        assert state["live"].pop(ptr) == nbytes
        state["frees"] += 1

    c2.lib.ah_context_set_allocator(c2.handle, alloc, free, None)
    try:
        rng = np.random.default_rng(8)
        n = 200_000
        h = HostArray(A.Int64, rng.integers(-9, 9, n), rng.random(n) < 0.9)
        m = HostArray(A.Boolean, rng.random(n) < 0.25)
        got = K.filter(h.to_device(c2), m.to_device(c2))
        assert arena.ptr <= got.values.ptr < arena.ptr + arena.nbytes      # values came from the hook
        assert arena.ptr <= got.validity.ptr < arena.ptr + arena.nbytes
        check(got, oracle.filter(h, m), "hooked filter")
        s = K.add_wrapping(got, got)
        assert arena.ptr <= s.values.ptr < arena.ptr + arena.nbytes
        check(s, oracle.arith(1, oracle.filter(h, m), oracle.filter(h, m)), "hooked add")
        live_before = len(state["live"])
        del got, s
        import gc
        gc.collect()
        assert len(state["live"]) < live_before and state["frees"] > 0     # release went back through the hook
    finally:
        c2.lib.ah_context_set_allocator(c2.handle, L.ALLOC_FN(0), L.FREE_FN(0), None)


def test_user_stream_and_two_contexts_in_threads(ctx, oracle):
    """Kernels follow ah_context_set_stream; two contexts used from two threads at once give the
    same answers (the reference's kernels are re-entrant pure functions, array/mod.rs:100)."""
    import threading
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(18)
    n = 300_000
    h = HostArray(A.Int64, rng.integers(-2**40, 2**40, n), rng.random(n) < 0.9)
    m = HostArray(A.Boolean, rng.random(n) < 0.4)
    exp = oracle.filter(h, m)
    c2 = A.Context(0)
    stream = torch.cuda.Stream(device=0)
    c2.lib.ah_context_set_stream(c2.handle, stream.cuda_stream)
    assert c2.lib.ah_context_stream(c2.handle) == stream.cuda_stream
    check(K.filter(h.to_device(c2), m.to_device(c2)), exp, "user stream")
    c2.lib.ah_context_set_stream(c2.handle, None)
    errs = []

    def work(c):
        try:
            dv, dm = h.to_device(c), m.to_device(c)
            for _ in range(20):
                check(K.filter(dv, dm), exp, "threaded")
                check(K.lt(dv, dv), oracle.compare(2, h, h), "threaded lt")
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)

    ths = [threading.Thread(target=work, args=(c,)) for c in (ctx, c2)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs


def test_one_context_shared_by_many_threads(ctx, oracle):
    """Every entry point locks its context (include/arrow_hip.h): six threads hammering ONE
    context — the pool, the read-back mailbox, the self-cleaning scratch and the error string are all per context —
    must give the oracle's answers, including the error texts of calls that fail."""
    import threading
    rng = np.random.default_rng(31)
    n = 200_000
    h = HostArray(A.Int64, rng.integers(-2**40, 2**40, n), rng.random(n) < 0.9)
    m = HostArray(A.Boolean, rng.random(n) < 0.3)
    idx = HostArray(A.UInt32, rng.integers(0, n, n // 7).astype(np.uint32))
    bad = HostArray(A.UInt32, np.array([1, n + 5], dtype=np.uint32))
    dv, dm, di, db = h.to_device(ctx), m.to_device(ctx), idx.to_device(ctx), bad.to_device(ctx)
    exp_f, exp_t = oracle.filter(h, m), oracle.take(h, idx)
    exp_c, exp_a = oracle.cast(h, A.Float64), oracle.arith(1, h, h)
    errs = []

    def work(k):
        try:
            for it in range(15):
                check(K.filter(dv, dm), exp_f, f"thread {k} filter")
                check(K.take(dv, di), exp_t, f"thread {k} take")
                check_exact(K.cast(dv, A.Float64), exp_c, f"thread {k} cast")
                check_exact(K.add_wrapping(dv, dv), exp_a, f"thread {k} add")
                if (it + k) % 5 == 0:
                    with pytest.raises(A.Panic) as ei:
                        K.take(dv, db)
                    assert str(ei.value) == f"index out of bounds: the len is {n} but the index is {n + 5}"
        except BaseException as ex:  # noqa: BLE001
            errs.append(repr(ex)[:300])

    ths = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs


@pytest.mark.skipif(__import__("os").environ.get("AH_DEBUG_GUARD") == "1", reason="the guard-page allocator neither rounds nor caches: the pool accounting asserted here does not apply")
def test_context_memory_stats(ctx):
    """ah_context_stats: the pooled allocator's accounting (MemoryPool::used / TrackingMemoryPool for HBM,
    arrow-buffer/src/pool.rs:73-93).  Live bytes follow results as they are produced and released, the high-water mark
    keeps the peak until it is reset, cached bytes are what ah_pool_trim hands back to HIP."""
    import gc
    gc.collect()
    n = 3_000_000
    a = HostArray(A.Int64, np.arange(n, dtype=np.int64)).to_device(ctx)
    s0 = ctx.memory_stats(reset_peaks=True)
    r = K.add_wrapping(a, a)  # one 24 MB result (rounded up to whole MiB by the pool)
    s1 = ctx.memory_stats()
    assert n * 8 <= s1["live_bytes"] - s0["live_bytes"] <= n * 8 + (2 << 20)
    assert s1["high_water_bytes"] >= s1["live_bytes"] and s1["alloc_calls"] > s0["alloc_calls"]
    assert s1["allocated_bytes_total"] - s0["allocated_bytes_total"] >= n * 8
    del r
    gc.collect()
    s2 = ctx.memory_stats()
    assert s2["live_bytes"] <= s0["live_bytes"] + (1 << 20) and s2["free_calls"] > s1["free_calls"]
    assert s2["high_water_bytes"] == s1["high_water_bytes"]  # the peak stays
    assert s2["cached_bytes"] >= n * 8  # the released block sits on a free list
    r2 = K.add_wrapping(a, a)  # ... and serves the next allocation of that size
    s3 = ctx.memory_stats(reset_peaks=True)
    assert s3["pool_hits"] > s2["pool_hits"] and s3["device_malloc_calls"] == s2["device_malloc_calls"]
    s4 = ctx.memory_stats()
    assert s4["high_water_bytes"] == s4["live_bytes"]  # reset: the peak restarts from what is live now
    # bytes over the host boundary: the upload of `a` was counted, a download is
    assert s4["host_to_device_bytes"] >= n * 8
    before = s4["device_to_host_bytes"]
    r2.values_numpy()
    assert ctx.memory_stats()["device_to_host_bytes"] - before == n * 8
    del r2


def test_bench_json_contract(ctx):
    """bench.py prints ONE JSON line with the contract keys (small --rows so it runs in seconds)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rows", "2000000", "--steps", "2",
                          "--warmup", "1", "--cpu-sample-rows", "1048576"], capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads(lines[0])
    for k in ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "Mrows/s" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["scaling"] == "weak" and d["data"] == "synthetic"
    for k in ["bound", "achieved", "peak", "unit", "frac", "traffic"]:
        assert k in d["roofline"], k
    for k in ["value", "unit", "cores", "kind", "sample"]:
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] == "port" and d["value"] > 0
    # round 6: every per-config fraction of the invocation sits INSIDE `roofline` (the driver's record keeps that object whole)
    bc = d["roofline"]["by_config"]
    assert "filter" in bc
    for k, c in bc.items():  # (avg_launch_ms is printed with 4 decimals: at this test's 2 M rows a launch is ~0.01 ms, hence the relative term)
        again = c["alg_bytes"] / (c["avg_launch_ms"] * 1e-3) / 1e9 / 8000.0
        assert c["avg_launch_ms"] > 0 and abs(c["frac"] - again) <= 0.05 * again + 2e-3, (k, c)
    # round 4: the stdout line is the COMPACT one and it comes last; the full-detail object is on stderr and in the file
    assert len(lines[0]) < 8000 and out.stdout.rstrip().splitlines()[-1] == lines[0]
    det = [l for l in out.stderr.splitlines() if l.startswith("BENCH_DETAIL {")]
    assert len(det) == 1 and json.loads(det[0][len("BENCH_DETAIL "):])["value"] == d["value"]
    # the reference's own criterion shapes (filter_kernels.rs:39-120, take_kernels.rs:32-80, arithmetic / comparison
    # kernels at 65 536 rows): synchronous us, amortised us inside a batch of 64 deferred calls, 1-core oracle us
    rs = d["reference_bench_shapes"]
    assert rs["columns"] == ["sync_us", "batched_us", "graph_us", "cpu_1core_us"] and rs["batch"] == 64
    names = set(rs["shapes"])
    for want in ("filter i32 (kept 1/2)", "filter i32 high selectivity (kept 1023/1024)", "filter i32 low selectivity (kept 1/1024)",
                 "filter context i32 w NULLs (kept 1/2)", "take i32 512", "take i32 1024", "add(0) f32", "lt f32"):
        assert want in names, want
    for name, (sync_us, batched_us, graph_us, cpu_us) in rs["shapes"].items():
        assert sync_us > 0 and cpu_us > 0 and batched_us is not None and batched_us > 0, name  # (round 5: take has the deferred form too)
        if name.startswith(("filter context", "add", "lt", "take")):  # fixed output shape: recordable into a hipGraph
            assert graph_us is not None and 0 < graph_us < 2 * sync_us, (name, graph_us, sync_us)  # (a timing: generous — it only has to be sane)
    assert set(rs["one_cpu_core_wins"]) <= names  # the honest crossover statement, whatever it is on this box
    # round 5: 4-byte and narrower operands, and the coalescer at the reference's batch sizes, in the same line
    if "configs_narrow" in d:
        for k in ("filter_i32", "take_i32", "add_wrapping_f32", "lt_f32", "lt_i32_scalar", "cast_i32_f64", "cast_i32_i64", "filter_i16", "filter_i8"):
            c = d["configs_narrow"][k]
            assert "error" not in c and c["avg_launch_ms"] > 0 and abs(c["frac"] - c["alg_bytes"] / (c["avg_launch_ms"] * 1e-3) / 1e9 / 8000.0) < 2e-3, (k, c)
    if "coalesce_by_batch_rows" in d:
        pts = d["coalesce_by_batch_rows"]["points"]
        for k in ("8192x8192", "65536x2^20", "16777216x2^20"):
            ms, mrows, frac, pushes, out_batches, launches = pts[k]
            assert ms > 0 and out_batches > 0 and launches < 400, (k, pts[k])


def test_filter_tile_and_chunk_boundaries(ctx, oracle):
    """Sizes around the 4096-row tile, the 1024-row count chunk and the 2048-row LDS stage, with
    selection patterns that put exactly CAP / CAP+1 / all rows of a tile into the output, for every
    value width, with and without validity, at odd slice offsets."""
    rng = np.random.default_rng(123)
    sizes = [1, 63, 64, 65, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193, 12289, 100003]
    dts = [A.UInt8, A.Int16, A.Int32, A.Int64, A.Decimal128(10, 2), A.Boolean]
    for n in sizes:
        patterns = {
            "all": np.ones(n, dtype=bool),
            "none": np.zeros(n, dtype=bool),
            "first_cap": np.arange(n) < 2048,
            "cap_plus_one": np.arange(n) < 2049,
            "every_other": (np.arange(n) % 2) == 0,
            "last_only": np.arange(n) == n - 1,
            "dense_tail": np.arange(n) >= max(0, n - 3000),
            "random": rng.random(n) < 0.31,
        }
        dt = dts[sizes.index(n) % len(dts)]
        if dt.physical == A._lib.AH_FIXED16:
            raw = np.zeros(n + 3, dtype=dt.np_dtype)
            raw["lo"] = rng.integers(0, 2**63, n + 3, dtype=np.uint64)
            vals = raw
        else:
            vals = _rand_values(rng, dt, n + 3)
        for with_valid in (False, True):
            full = HostArray(dt, vals, (rng.random(n + 3) < 0.7) if with_valid else None)
            dfull = full.to_device(ctx)
            for off in (0, 3):
                h, d = full.slice(off, n), dfull.slice(off, n)
                for name, pat in patterns.items():
                    m = HostArray(A.Boolean, pat)
                    got = K.filter(d, m.to_device(ctx))
                    check(got, oracle.filter(h, m), f"n={n} {dt} {name} valid={with_valid} off={off}")


def test_bitmap_ops_large_offsets(ctx, oracle):
    """Bit offsets beyond one word on every bitmap input (sliced far into a parent buffer)."""
    rng = np.random.default_rng(321)
    n, off = 70_000, 1000 + 37
    a = HostArray(A.Int32, rng.integers(-5, 5, n + off).astype(np.int32), rng.random(n + off) < 0.8)
    b = HostArray(A.Int32, rng.integers(-5, 5, n + off + 9).astype(np.int32), rng.random(n + off + 9) < 0.8)
    m = HostArray(A.Boolean, rng.random(n + off + 5) < 0.4, rng.random(n + off + 5) < 0.9)
    da, db, dm = a.to_device(ctx).slice(off, n), b.to_device(ctx).slice(off + 9, n), m.to_device(ctx).slice(off + 5, n)
    ha, hb, hm = a.slice(off, n), b.slice(off + 9, n), m.slice(off + 5, n)
    check_exact(K.add_wrapping(da, db), oracle.arith(1, ha, hb), "add sliced")
    check(K.lt(da, db), oracle.compare(2, ha, hb), "lt sliced")
    check(K.filter(da, dm), oracle.filter(ha, hm), "filter sliced")
    check(K.and_kleene(K.lt(da, db), dm), oracle.boolean_binary(3, oracle.compare(2, ha, hb), hm), "kleene sliced")
    check_exact(K.cast(da, A.Float64), oracle.cast(ha, A.Float64), "cast sliced")
    idx = HostArray(A.UInt32, rng.integers(0, n, 5000).astype(np.uint32), rng.random(5000) < 0.9)
    check_exact(K.take(da, idx.to_device(ctx)), oracle.take(ha, idx), "take sliced")


@pytest.mark.skipif(__import__("os").environ.get("AH_DEBUG_GUARD") == "1", reason="the gloo TEST transport hands device pointers to writev(): hipMalloc memory is host-visible on these boxes, a guard mapping (hipMemMap, device access only) is not (EFAULT in gloo's tcp pair)")
@pytest.mark.parametrize("world", [2, 3])
def test_communicator_two_ranks_one_gpu(ctx, world):
    """N>1 on the device path: `world` processes share GPU 0 over gloo (device tensors); sharded
    device filter + Communicator.all_gatherv must equal the oracle's un-sharded filter on every rank."""
    import os
    import subprocess
    import sys
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", GLOO_SOCKET_IFNAME="lo")
    rdzv = os.path.join(tempfile.mkdtemp(prefix="ah_rdzv_"), "store")  # a file store: no port to lose (see test_communicator_world1_nccl)
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "dist_gpu_worker.py"), str(r), str(world), rdzv],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, o[-3000:]


# ------------------------------------------------------------------ view arrays
def _view_items(rng, n):
    words = ["", "a", "twelve bytes", "thirteen byte", "x" * 40, "näïve ünïcödé strings", "0123456789" * 7]
    return [None if rng.random() < 0.15 else words[rng.integers(0, len(words))] + str(rng.integers(0, 10**6))
            for _ in range(n)]


@pytest.mark.parametrize("dt", [A.Utf8View, A.BinaryView], ids=repr)
def test_byte_view_filter_take(ctx, dt):
    """filter_byte_view (filter.rs:931-944) / take_byte_view (take.rs:630-640): views are filtered / gathered as
    16-byte natives, the data-buffer list is shared with the input, nulls follow filter_nulls / take_nulls."""
    # the reference's own inline vectors first: _test_filter_byte_view (filter.rs:1278-1328, both predicates) and
    # _test_byte_view (take.rs:1780-1815) — ["hello", "world", null, "large payload over 12 bytes", "lulu"]
    conv = (lambda x: x) if dt == A.Utf8View else (lambda x: None if x is None else x.encode())
    ref = [conv(x) for x in ["hello", "world", None, "large payload over 12 bytes", "lulu"]]
    ra = A.Array.from_string_views(ref, dt, ctx=ctx)
    f1 = K.filter(ra, A.Array.from_numpy(np.array([True, False, True, True, False]), ctx=ctx))
    assert f1.to_pylist() == [conv(x) for x in ["hello", None, "large payload over 12 bytes"]] and f1.length == 3
    f2 = K.filter(ra, A.Array.from_numpy(np.array([True, False, False, False, True]), ctx=ctx))
    assert f2.to_pylist() == [conv(x) for x in ["hello", "lulu"]] and f2.length == 2
    ti = A.Array.from_numpy(np.array([3, 0, 1, 3, 4, 2], dtype=np.uint32), np.array([True, False, True, True, True, True]), ctx=ctx)
    t1 = K.take(ra, ti)
    assert t1.to_pylist() == [conv(x) for x in ["large payload over 12 bytes", None, "world", "large payload over 12 bytes", "lulu", None]]
    rng = np.random.default_rng(12)
    n = 20_000
    items = _view_items(rng, n)
    if dt == A.BinaryView:
        items = [None if s is None else s.encode() + b"\xff\x00" for s in items]
    arr = A.Array.from_string_views(items, dt, ctx=ctx, bit_offset=3, block_size=4096)
    assert len(arr.data_buffers) > 10 and arr.to_pylist() == items
    mask = rng.random(n) < 0.3
    mvalid = rng.random(n) < 0.9
    pred = A.Array.from_numpy(mask, mvalid, ctx=ctx, bit_offset=1)
    f = K.filter(arr, pred)
    want = [it for it, m, v in zip(items, mask, mvalid) if m and v]
    assert f.data_type == dt and f.to_pylist() == want and f.null_count() == sum(x is None for x in want)
    assert f.data_buffers is arr.data_buffers  # `array.data_buffers().to_vec()`: shared, not copied
    idx = rng.integers(0, n, 7000).astype(np.uint32)
    ivalid = rng.random(7000) < 0.9
    t = K.take(arr, A.Array.from_numpy(idx, ivalid, ctx=ctx))
    assert t.to_pylist() == [items[i] if v else None for i, v in zip(idx, ivalid)]
    assert t.data_buffers is arr.data_buffers
    s = arr.slice(17, 5000)
    assert K.filter(s, pred.slice(17, 5000)).to_pylist() == [it for it, m, v in zip(items[17:5017], mask[17:5017], mvalid[17:5017]) if m and v]
    allf = K.filter(arr, A.Array.from_numpy(np.ones(n, bool), ctx=ctx))  # All fast path: zero-copy slice
    assert allf.to_pylist() == items
    nf = K.nullif(arr, pred)
    assert nf.to_pylist() == [None if (m and v) else it for it, m, v in zip(items, mask, mvalid)]
    with pytest.raises(A.array.NotYetImplemented):
        K.concat([arr, arr])


def test_bench_two_ranks_one_gpu(ctx):
    """bench.py's N>1 path end to end (torch.distributed.run, sharded generation, take on the side context
    overlapping the all-gatherv, max-over-ranks timing, one JSON line from rank 0).  The box has one GPU, so
    both ranks share it over gloo; everything but the transport is what the driver runs at N=2."""
    import json
    import os
    import sys
    from launch import run_with_port
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AH_BENCH_BACKEND="gloo", AH_BENCH_SHARED_GPU="1", GLOO_SOCKET_IFNAME="lo")
    r = run_with_port(lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                                    "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                                    "--warmup", "1", "--rows", "50000000", "--no-cpu-baseline"], timeout=600, env=env, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["steps"] == 2
    assert d["config"]["reassemble"] == "allgatherv", d["config"]
    assert d["local_value"] >= d["value"] * 0.5
    assert d["roofline"]["kernel"] in ("take_gather", "filter_scatter") and d["roofline"]["achieved"] > 0
    ex = d["exchange"]  # per-link accounting of the all-gatherv: ~5e6 selected rows x 8 B + validity bits
    assert ex["peers"] == 1 and 30e6 < ex["bytes_to_each_peer"] < 50e6 and ex["per_link_GBps"] > 0
    assert abs(ex["bytes_received"] - ex["bytes_to_each_peer"]) < 2e6  # the other shard has the same density


# ------------------------------------------------------------------ string compare
@pytest.mark.parametrize("case", orc.load_golden("cmp_utf8"), ids=lambda c: c["name"])
def test_cmp_utf8_reference_goldens(ctx, case):
    """comparison.rs:1246-1432 through the device kernel."""
    fn = getattr(K, case["op"])
    for dt in (A.Utf8, A.LargeUtf8):
        l = A.Array.from_strings(list(case["lhs"]), data_type=dt, ctx=ctx)
        if "rhs_scalar" in case:
            r = A.Scalar(A.Array.from_strings([case["rhs_scalar"]], data_type=dt, ctx=ctx))
        else:
            r = A.Array.from_strings(list(case["rhs"]), data_type=dt, ctx=ctx)
        got = fn(l, r)
        assert got.nulls() is None and got.values_numpy().tolist() == case["expected"]


def test_utf8_eq_scalar_on_slice(ctx, oracle):
    """test_utf8_eq_scalar_on_slice (arrow-ord/src/comparison.rs:1147-1164): ["hi", null, "hello", "world", ""].slice(1, 4) == "hello"
    and == "" — a sliced string column with a null against a scalar, on the device and on the oracle."""
    for dt in (A.Utf8, A.LargeUtf8):
        h = HostArray.from_pylist(["hi", None, "hello", "world", ""], dt)
        d = h.to_device(ctx).slice(1, 4)
        for text, exp in (("hello", [None, True, False, False]), ("", [None, False, False, True])):
            sc = HostArray.from_pylist([text], dt)
            assert K.eq(d, A.Scalar(sc.to_device(ctx))).to_pylist() == exp
            assert oracle.compare(0, h.slice(1, 4), sc, r_scalar=True).to_pylist() == exp


def test_cmp_strings_fuzz(ctx, oracle):
    """All eight operators, array/array, array/scalar, scalar/array, nulls, long common prefixes (the 8-byte word
    loop and its byte tail), embedded zero bytes, empty strings; then predicate -> filter on the string column."""
    rng = np.random.default_rng(21)
    stems = ["", "a", "ab", "abcdefgh", "abcdefghi", "abcdefgh\x00", "abcdefghijklmnopqrstuvwxy", "abcdefghijklmnopqrstuvwxz",
             "b", "\x7f", "zzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzz", "zzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzy"]
    for dt in (A.Utf8, A.LargeUtf8):
        for n in (1, 64, 65, 1000, 20_001):
            ls = [stems[rng.integers(0, len(stems))] for _ in range(n)]
            rs = [stems[rng.integers(0, len(stems))] for _ in range(n)]
            for nullable in (False, True):
                lv = (rng.random(n) < 0.8) if nullable else None
                rv = (rng.random(n) < 0.8) if nullable else None
                hl, hr = HostArray(dt, ls, lv), HostArray(dt, rs, rv)
                dl, dr = hl.to_device(ctx), hr.to_device(ctx)
                sc = HostArray(dt, [stems[rng.integers(0, len(stems))]])
                dsc = A.Scalar(sc.to_device(ctx))
                for op, name in enumerate(["eq", "neq", "lt", "lt_eq", "gt", "gt_eq", "distinct", "not_distinct"]):
                    fn = getattr(K, name)
                    check(fn(dl, dr), oracle.compare(op, hl, hr), f"{dt} {name} n={n}")
                    check(fn(dl, dsc), oracle.compare(op, hl, sc, r_scalar=True), f"{dt} {name} scalar rhs")
                    check(fn(dsc, dr), oracle.compare(op, sc, hr, l_scalar=True), f"{dt} {name} scalar lhs")
    # sliced inputs + the pipeline the kernel exists for
    n = 50_000
    strs = [f"key-{int(x):05d}" for x in rng.integers(0, 1000, n)]
    h = HostArray(A.Utf8, strs, rng.random(n) < 0.9)
    d = h.to_device(ctx)
    pivot = HostArray(A.Utf8, ["key-00500"])
    pred = K.lt(d.slice(100, 40_000), A.Scalar(pivot.to_device(ctx)))
    check(pred, oracle.compare(2, h.slice(100, 40_000), pivot, r_scalar=True), "sliced lt scalar")
    got = K.filter(d.slice(100, 40_000), pred)
    check(got, oracle.filter(h.slice(100, 40_000), oracle.compare(2, h.slice(100, 40_000), pivot, r_scalar=True)), "lt -> filter")
    with pytest.raises(A.array.InvalidArgumentError, match="Invalid comparison operation: Utf8 < LargeUtf8"):
        K.lt(d, A.Array.from_strings(strs, data_type=A.LargeUtf8, ctx=ctx))


# ------------------------------------------------------------------ Boolean <-> numeric casts
def test_cast_bool_numeric(ctx, oracle):
    """cast/mod.rs:1243-1290: numeric -> Boolean is `value != 0` (NaN true, -0.0 false; a null buffer only when
    there are nulls), Boolean -> numeric is 1 / 0 with a null buffer always present."""
    rng = np.random.default_rng(19)
    nums = [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64, A.Float32, A.Float64]
    for n in (1, 63, 64, 65, 5000, 100_003):
        for dt in nums:
            v = _rand_values(rng, dt, n)
            v[rng.integers(0, n, max(1, n // 3))] = 0
            if np.dtype(dt.np_dtype).kind == "f":
                v[rng.integers(0, n, max(1, n // 7))] = np.nan
                v[rng.integers(0, n, max(1, n // 7))] = -0.0
            for valid in (None, rng.random(n) < 0.8):
                h = HostArray(dt, v, valid)
                got = K.cast(h.to_device(ctx, bit_offset=3), A.Boolean)
                check_exact(got, oracle.cast(h, A.Boolean), f"{dt}->Boolean n={n}")
                b = HostArray(A.Boolean, rng.random(n) < 0.5, valid)
                got = K.cast(b.to_device(ctx, bit_offset=5), dt)
                check_exact(got, oracle.cast(b, dt), f"Boolean->{dt} n={n}")
                assert got.nulls() is not None
    assert K.can_cast_types(A.Boolean, A.Int32) and K.can_cast_types(A.Float64, A.Boolean)
    # reference goldens: cast/mod.rs `test_cast_bool_to_i32` / `test_cast_i32_to_bool`-style literals
    b = A.Array.from_pylist([True, False, None], A.Boolean, ctx=ctx)
    assert K.cast(b, A.Int32).to_pylist() == [1, 0, None] and K.cast(b, A.Float64).to_pylist() == [1.0, 0.0, None]
    i = A.Array.from_pylist([0, 1, 2, None, -3], A.Int32, ctx=ctx)
    assert K.cast(i, A.Boolean).to_pylist() == [False, True, True, None, True]


# ------------------------------------------------------------------ bitwise kernels
def test_bitwise_kernels(ctx, oracle):
    """arrow-arith/src/bitwise.rs and its tests (:209-380): and/or/xor/and_not/not, shifts with counts taken modulo
    the bit width (negative and oversized counts included), `_scalar` forms, `binary`/`unary` null rules."""
    i32 = lambda xs: A.Array.from_pylist(xs, A.Int32, ctx=ctx)  # noqa: E731
    u64 = lambda xs: A.Array.from_pylist(xs, A.UInt64, ctx=ctx)  # noqa: E731
    # reference literals
    assert K.bitwise_and(u64([1, 2, None, 4]), u64([5, 6, 7, 8])).to_pylist() == [1, 2, None, 0]          # :213
    assert K.bitwise_and(i32([1, -2, None, 4]), i32([5, -6, 8, 9])).to_pylist() == [1, -6, None, 0]       # :220
    assert K.bitwise_shift_left(u64([1, 2, None, 4, 8]), u64([5, 10, 15, None, 20])).to_pylist() == [32, 2048, None, None, 8388608]  # :230
    assert K.bitwise_shift_left_scalar(u64([1, 2, None, 4, 8]), A.Scalar.new(2, A.UInt64, ctx)).to_pylist() == [4, 8, None, 16, 32]  # :242
    assert K.bitwise_shift_right(u64([32, 2048, None, 32768, 8388608]), u64([5, 10, 15, None, 20])).to_pylist() == [1, 2, None, None, 8]  # :250
    assert K.bitwise_or(u64([1, 2, None, 4]), u64([7, 5, 8, 13])).to_pylist() == [7, 7, None, 13]          # :280
    assert K.bitwise_not(u64([0, 2, None, 4])).to_pylist() == [2**64 - 1, 2**64 - 3, None, 2**64 - 5]       # :300
    assert K.bitwise_not(i32([0, -2, None, 4])).to_pylist() == [-1, 1, None, -5]
    assert K.bitwise_and_not(u64([8, 2, None, 4]), u64([7, 5, 8, 13])).to_pylist() == [8, 2, None, 0]      # :312
    assert K.bitwise_xor(u64([1, 2, None, 4]), u64([7, 5, 8, 13])).to_pylist() == [6, 7, None, 9]          # :340
    with pytest.raises(A.array.InvalidArgumentError):
        K.bitwise_and(A.Array.from_numpy(np.zeros(3), ctx=ctx), A.Array.from_numpy(np.zeros(3), ctx=ctx))
    rng = np.random.default_rng(23)
    for dt in (A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64):
        for n in (1, 65, 10_001):
            a = HostArray(dt, _rand_values(rng, dt, n), rng.random(n) < 0.8)
            b = HostArray(dt, _rand_values(rng, dt, n), rng.random(n) < 0.8 if n % 2 else None)
            sc = HostArray(dt, _rand_values(rng, dt, 1))
            da, db = a.to_device(ctx, bit_offset=3), b.to_device(ctx, bit_offset=1)
            for op, fn in ((8, K.bitwise_and), (9, K.bitwise_or), (10, K.bitwise_xor), (11, K.bitwise_shift_left),
                           (12, K.bitwise_shift_right), (13, K.bitwise_and_not)):
                check_exact(fn(da, db), oracle.bitwise(op, a, b), f"{dt} op {op}")
                check_exact(fn(da, A.Scalar(sc.to_device(ctx))), oracle.bitwise(op, a, sc, r_scalar=True), f"{dt} op {op} scalar")
            check_exact(K.bitwise_not(da), oracle.bitwise(14, a), f"{dt} not")


def test_filter_take_beyond_2_pow_32_rows(ctx):
    """64-bit row indexing: an Int8 column of 2^32 + 4099 rows (4.3 GB) filtered at 1 % and taken with UInt64
    indices that point past 2^32.  Checked through the count, the head and the tail of the result."""
    import ctypes as C
    n = (1 << 32) + 4099
    vb = ctx.alloc(((n + 3) // 4) * 4)
    ctx.check(ctx.lib.ah_gen_uniform_i32(ctx.handle, vb.ptr, (n + 3) // 4, 7, 0))
    vals = A.Array(ctx, A.Int8, n, A.array._RawMem(vb.ptr, n, vb))
    mb = ctx.alloc(((n + 63) // 64) * 8)
    ctx.check(ctx.lib.ah_gen_bernoulli_bits(ctx.handle, mb.ptr, n, 9, 0.01, 0))
    mask = A.Array(ctx, A.Boolean, n, A.array._RawMem(mb.ptr, mb.nbytes, mb))
    f = K.filter(vals, mask)
    cnt = C.c_int64()
    ctx.check(ctx.lib.ah_count_set_bits(ctx.handle, mb.ptr, 0, n, C.byref(cnt)))
    assert f.length == cnt.value and abs(cnt.value / n - 0.01) < 1e-4
    # head and tail windows re-derived on the host from the raw device bytes
    w = 1 << 20
    for lo in (0, n - w):
        hv = vb.to_numpy(np.int8, byte_offset=lo, count=w)
        first_bit = lo  # lo is a multiple of 8 in both cases? n - w is not: fetch covering bytes
        b0 = first_bit // 8
        nb = (lo + w + 7) // 8 - b0
        hm = A.unpack_bits(mb.to_numpy(np.uint8, byte_offset=b0, count=nb), first_bit - b0 * 8, w)
        want = hv[hm]
        got = f.slice(0, len(want)).values_numpy() if lo == 0 else f.slice(f.length - len(want), len(want)).values_numpy()
        assert np.array_equal(got, want), lo
    idx = np.array([0, (1 << 32) - 1, 1 << 32, (1 << 32) + 4098, 123456789012 % n], dtype=np.uint64)
    t = K.take(vals, A.Array.from_numpy(idx, ctx=ctx))
    want = np.array([vb.to_numpy(np.int8, byte_offset=int(i), count=1)[0] for i in idx], dtype=np.int8)
    assert np.array_equal(t.values_numpy(), want)
    window = vb.to_numpy(np.int8, byte_offset=(1 << 32) - 5, count=4104)
    got_sum = K.aggregate.sum(vals.slice((1 << 32) - 5, 4104))
    assert int(got_sum) == int(np.array([int(window.astype(np.int64).sum()) & 0xFF], dtype=np.uint8).view(np.int8)[0])
    assert int(K.aggregate.max(vals.slice((1 << 32) - 5, 4104))) == int(window.max())
    # elementwise kernels across the 2^32 boundary: tails re-derived on the host
    tail = 1 << 16
    hv = vb.to_numpy(np.int8, byte_offset=n - tail, count=tail)
    add = K.add_wrapping(vals, vals)
    assert np.array_equal(add.slice(n - tail, tail).values_numpy(), (hv.astype(np.int16) * 2).astype(np.int8))
    assert int(K.aggregate.sum(K.cast(K.lt(vals, A.Scalar.new(0, A.Int8, ctx)).slice(n - tail, tail), A.Int32))) == int((hv < 0).sum())
    c16 = K.cast(vals, A.Int16)
    assert c16.length == n and np.array_equal(c16.slice(n - tail, tail).values_numpy(), hv.astype(np.int16))
    with pytest.raises(A.array.InvalidArgumentError, match="UInt32 indices"):
        K.sort_to_indices(vals)
