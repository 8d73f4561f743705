"""Utf8 / LargeUtf8 -> numeric casts on device (arrow_cast parse_string, arrow-cast/src/cast/string.rs:66-120):
the HIP kernels through the C ABI against the CPU oracle (itself pinned to the reference's literals in
tests/golden/cast.json and to Python's float / int / exact rationals in tests/test_parse_cpu.py), plus round trips
at bench scale: cast(cast(x, LargeUtf8), T) == x bit for bit (Ryu's shortest digits must parse back exactly)."""
import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K
from orc import HostArray, assert_logical_eq, assert_same_nulls_presence

pytestmark = pytest.mark.gpu

NUMERIC = [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64, A.Float32, A.Float64]


def host(a):
    return HostArray.from_device(a)


def _fuzz_texts(rng, n):
    digits = list("0123456789")
    out = []
    for _ in range(n):
        k = int(rng.integers(0, 12))
        if k == 0:
            t = str(int(rng.integers(-300, 300)))
        elif k == 1:
            t = str(int(rng.integers(-2**63, 2**63 - 1, dtype=np.int64)))
        elif k == 2:
            t = repr(float(np.frombuffer(rng.bytes(8), dtype=np.float64)[0]))
        elif k == 3:
            t = repr(float(np.frombuffer(rng.bytes(4), dtype=np.float32)[0]))
        elif k == 4:  # long digit strings: > 19 significant digits, > 32 bytes (memory path), the exact slow path
            t = "".join(rng.choice(digits, int(rng.integers(1, 60)))) + "." + "".join(rng.choice(digits, int(rng.integers(0, 60))))
        elif k == 5:
            t = f"{int(rng.integers(0, 10**18))}e{int(rng.integers(-345, 312))}"
        elif k == 6:
            t = str(rng.choice(["nan", "NaN", "inf", "-inf", "Infinity", "+INF", "-nan", "infinit", "nanx", "", "+", "-", ".", "e5",
                                "1e", "1e+", "0x10", "1_0", "seven", "9.1", "1 2", "--1", "١٢", "1.5\x0b"]))
        elif k == 7:
            t = str(int(rng.choice([127, 128, -128, -129, 255, 256, 32767, 32768, -32768, -32769, 65535, 65536, 2**31 - 1, 2**31,
                                    -2**31, -2**31 - 1, 2**32 - 1, 2**32, 2**63 - 1, 2**63, -2**63, -2**63 - 1, 2**64 - 1, 2**64])))
        elif k == 8:
            t = str(rng.choice([" ", "\t", "\n", "\r\n", "\x0c", ""])) + str(int(rng.integers(-99, 99))) + str(rng.choice(["", " ", "\n\t", ".", ".5", "e2"]))
        elif k == 9:
            t = str(rng.choice(["+", "-", ""])) + "0" * int(rng.integers(0, 25)) + str(int(rng.integers(0, 1000)))
        elif k == 10:  # exact ties between adjacent doubles / floats, written out in full
            if rng.random() < 0.5:
                b = int(rng.integers(0, 2**62))
                lo = np.frombuffer(np.uint64(b).tobytes(), dtype=np.float64)[0]
                hi = np.nextafter(lo, np.inf)
            else:
                b = int(rng.integers(0, 2**31 - 2**23))
                lo = float(np.frombuffer(np.uint32(b).tobytes(), dtype=np.float32)[0])
                hi = float(np.nextafter(np.float32(lo), np.float32(np.inf)))
            from fractions import Fraction
            mid = (Fraction(float(lo)) + Fraction(float(hi))) / 2
            # exact decimal expansion of a dyadic rational
            num, den = mid.numerator, mid.denominator
            k2 = den.bit_length() - 1
            t = str(num * 5**k2)
            t = (t[:-k2] or "0") + "." + t[-k2:].rjust(k2, "0") if k2 else t
            if len(t) > 900:
                t = "1.5"
            if rng.random() < 0.5:
                t += str(rng.choice(["1", "0", "9"]))
        else:
            t = f"{rng.normal() * 10.0 ** float(rng.integers(-30, 30)):.{int(rng.integers(0, 30))}g}"
        out.append(t)
    return out


@pytest.mark.parametrize("to", NUMERIC, ids=str)
@pytest.mark.parametrize("dt", [A.Utf8, A.LargeUtf8], ids=str)
def test_parse_fuzz_vs_oracle(ctx, oracle, dt, to):
    rng = np.random.default_rng(100 + NUMERIC.index(to))
    for it in range(3):
        n = int(rng.integers(1, 3000))
        rows = _fuzz_texts(rng, n)
        valid = (rng.random(n) < 0.85) if it != 1 else None
        hv = HostArray(dt, rows, valid)
        dv = hv.to_device(ctx)
        exp = oracle.cast(hv, to)
        got = K.cast(dv, to)
        assert_logical_eq(host(got), exp, f"{dt}->{to} iter {it}")
        assert_same_nulls_presence(host(got), exp, f"{dt}->{to}")
        assert got.validity is not None  # from_trusted_len_iter: always a null buffer
        # null / unparsable slots hold 0 (PrimitiveArray::from_trusted_len_iter writes the default)
        g = np.asarray(host(got).values)
        assert not g[~exp.valid].view(np.uint8).any()
        off = int(rng.integers(0, min(n, 130)))
        sl = K.cast(dv.slice(off, n - off), to)
        assert_logical_eq(host(sl), oracle.cast(hv.slice(off, n - off), to), f"sliced {dt}->{to}")


@pytest.mark.parametrize("to", [A.Int32, A.UInt64, A.Float32, A.Float64], ids=str)
def test_parse_unsafe_mode(ctx, oracle, to):
    opts = K.CastOptions(safe=False)
    good = HostArray(A.Utf8, ["1", " 22", "", "4\n", "+5"], np.array([True, True, False, True, True]))
    got = K.cast_with_options(good.to_device(ctx), to, opts)
    exp = oracle.cast(good, to, safe=False)
    assert_logical_eq(host(got), exp, "unsafe ok")
    assert_same_nulls_presence(host(got), exp, "unsafe ok")
    no_nulls = HostArray(A.LargeUtf8, ["1", "2", "3"])
    got = K.cast_with_options(no_nulls.to_device(ctx), to, opts)
    assert got.validity is None and got.to_pylist() == oracle.cast(no_nulls, to, safe=False).to_pylist()
    # the FIRST offending valid row is reported, with its text; rows under nulls are never parsed
    bad = HostArray(A.Utf8, ["1", "zzz", "x y", "4", "1e", "seven"], np.array([True, False, True, True, True, True]))
    with pytest.raises(A.array.CastError) as ei:
        K.cast_with_options(bad.to_device(ctx), to, opts)
    with pytest.raises(A.array.CastError) as eo:
        oracle.cast(bad, to, safe=False)
    assert ei.value.message == eo.value.message == f"Cannot cast string 'x y' to value of {to} type"
    many = ["7"] * 100_000
    many[77_777] = "7.5.1"
    many[99_000] = "nope"
    with pytest.raises(A.array.CastError) as ei:
        K.cast_with_options(HostArray(A.Utf8, many).to_device(ctx), to, opts)
    assert ei.value.message == f"Cannot cast string '7.5.1' to value of {to} type"


def test_parse_empty_and_can_cast(ctx):
    for dt in (A.Utf8, A.LargeUtf8):
        for to in NUMERIC:
            assert K.can_cast_types(dt, to)
            e = K.cast(A.Array.from_strings([], None, dt, ctx), to)
            assert e.length == 0 and e.data_type == to
    assert not K.can_cast_types(A.Utf8View, A.Int32)


def test_parse_slow_path_rows_on_device(ctx, oracle):
    """> 19 significant digits whose truncation straddles a rounding boundary: decided by the big-integer kernel."""
    rows = ["9007199254740993.0000000000000000000000001", "9007199254740993.00000000000000000000000000",
            "9007199254740992.99999999999999999999999999", "2.4703282292062327208051355972788608e-324",
            "2.4703282292062327208051355972788609e-324", "1.79769313486231580793728971405301e308",
            "1.79769313486231580793728971405304e308", "16777217.000000000000000000000000000000001", "16777217.0000000000000000000000000000000",
            "7.0064923216240853546186479164495806564013097093825788587853e-46", "7.0064923216240853546186479164495806564013097093825788587854e-46",
            "0." + "0" * 400 + "1" + "5" * 400, "1" + "0" * 300 + "." + "0" * 500 + "1", "-4.35" + "0" * 40 + "1e-320"]
    hv = HostArray(A.LargeUtf8, rows * 50)
    dv = hv.to_device(ctx)
    for to in (A.Float64, A.Float32):
        ctx.profile(True)
        ctx.profile_reset()
        got = K.cast(dv, to)
        assert_logical_eq(host(got), oracle.cast(hv, to), f"slow path {to}")
        assert ctx.profile_get("cast_parse_slow")[1] == 1  # the exact kernel really ran
        ctx.profile(False)


def test_parse_round_trip_at_scale(ctx):
    """2^26 rows: Int64 -> LargeUtf8 -> Int64 and Int64 -> Float64 -> LargeUtf8 -> Float64 give the input back bit for
    bit (the bench's config-4 column: small integers + 1 % full-range, 10 % nulls)."""
    import bench
    n = 1 << 26
    src = bench.gen_i64_column(A, ctx, n, 42, 0.9, 0, -10**6, 10**6)
    text = K.cast(src, A.LargeUtf8)
    back = K.cast(text, A.Int64)
    assert back.null_count() == src.null_count()
    ne = K.neq(back, src)
    assert K.aggregate.bool_or(ne) in (False, None)
    assert K.aggregate.sum(back) == K.aggregate.sum(src)
    del text, back, ne
    wide = bench.gen_i64_column(A, ctx, n, 43, 0.9, 0)  # full range: exponent forms, 17 significant digits
    f = K.cast(wide, A.Float64)
    text = K.cast(f, A.LargeUtf8)
    back = K.cast(text, A.Float64)
    assert back.null_count() == f.null_count()
    assert K.aggregate.bool_or(K.neq(back, f)) in (False, None)  # float `neq` is bit inequality under totalOrder
    f32 = K.cast(wide, A.Float32)
    back32 = K.cast(K.cast(f32, A.LargeUtf8), A.Float32)
    assert K.aggregate.bool_or(K.neq(back32, f32)) in (False, None)
