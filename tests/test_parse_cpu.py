"""Utf8 -> numeric parsing, CPU side (no GPU):

* the product's algorithm header (arrow-rs_amd/csrc/parse_num.hpp: Eisel-Lemire + the big-integer slow path + the
  integer parser) compiled for the HOST and fuzzed against glibc strtod / strtof / wide integer arithmetic
  (tests/cpp/parse_num_host_test.cpp) — the same source the device kernels compile;
* the oracle's restatement of `Parser::parse` (oracle/oracle.cpp parse_string) against independent implementations:
  Python's float() (David Gay's correctly rounded dtoa) for Float64, exact rational arithmetic for Float32, and
  Python's int() for the integer types.
"""
import os
import subprocess
from fractions import Fraction

import numpy as np
import pytest

import arrow_rs_amd as A
from orc import HostArray

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parse_num_header_on_host(tmp_path):
    exe = str(tmp_path / "parse_num_host_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "parse_num_host_test.cpp")], check=True)
    r = subprocess.run([exe, "300000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok "), r.stdout
    cases, slow = int(r.stdout.split()[1]), int(r.stdout.split()[3])
    assert cases > 2_000_000 and slow > 50_000  # the exact path is really exercised


def _texts(rng, n):
    out = []
    for _ in range(n):
        kind = rng.integers(0, 6)
        if kind == 0:
            t = repr(float(np.frombuffer(rng.bytes(8), dtype=np.float64)[0]))
            if "n" in t:  # nan / inf
                t = "1.0"
        elif kind == 1:
            t = str(int(rng.integers(-10**18, 10**18)))
        elif kind == 2:
            ni, nf = int(rng.integers(0, 30)), int(rng.integers(0, 30))
            t = "".join(rng.choice(list("0123456789"), ni)) + "." + "".join(rng.choice(list("0123456789"), nf))
            if ni + nf == 0:
                t = "0."
        elif kind == 3:
            t = f"{int(rng.integers(0, 10**17))}e{int(rng.integers(-340, 310))}"
        elif kind == 4:
            t = f"{rng.random():.{int(rng.integers(0, 25))}e}".replace("e", "E")
        else:
            t = f"-{rng.random() * 10.0 ** float(rng.integers(-50, 50)):.{int(rng.integers(1, 40))}f}"
        out.append(t)
    return out


def _nearest_f32_bits(x: Fraction, neg: bool) -> int:
    """Correctly rounded (nearest, ties to even) binary32 of an exact rational, by exact arithmetic only."""
    sign = 0x80000000 if neg else 0
    x = abs(x)
    if x == 0:
        return sign
    e = x.numerator.bit_length() - x.denominator.bit_length()
    while Fraction(2) ** e > x:
        e -= 1
    while Fraction(2) ** (e + 1) <= x:
        e += 1
    e = max(e, -126)                       # subnormals share the smallest normal's quantum
    n = x / Fraction(2) ** (e - 23)
    fl = n.numerator // n.denominator
    rem = n - fl
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and fl & 1):
        fl += 1
    if fl == 1 << 24:
        fl, e = 1 << 23, e + 1
    if e > 127:
        return sign | 0x7F800000
    if fl < 1 << 23:
        return sign | fl
    return sign | ((e + 127) << 23) | (fl - (1 << 23))


def test_oracle_float_parse_vs_python(oracle):
    rng = np.random.default_rng(7)
    texts = _texts(rng, 4000)
    got64 = oracle.cast(HostArray(A.Utf8, texts), A.Float64)
    assert got64.valid is not None and got64.valid.all(), [t for t, v in zip(texts, got64.valid) if not v][:5]
    exp64 = np.array([float(t) for t in texts], dtype=np.float64)
    assert np.array_equal(np.asarray(got64.values).view(np.uint64), exp64.view(np.uint64))
    sub = texts[:600]
    got32 = oracle.cast(HostArray(A.LargeUtf8, sub), A.Float32)
    bits = np.asarray(got32.values).view(np.uint32)
    for t, b in zip(sub, bits):
        assert int(b) == _nearest_f32_bits(Fraction(t), t.startswith("-")), t


@pytest.mark.parametrize("dt", [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64], ids=str)
def test_oracle_int_parse_vs_python(oracle, dt):
    rng = np.random.default_rng(8)
    info = np.iinfo(dt.np_dtype)
    texts = []
    for _ in range(3000):
        k = rng.integers(0, 5)
        if k == 0:
            v = int(rng.integers(info.min, info.max, endpoint=True, dtype=dt.np_dtype))
        elif k == 1:
            v = int(info.max) + int(rng.integers(-3, 4))
        elif k == 2:
            v = int(info.min) + int(rng.integers(-3, 4))
        else:
            v = int(rng.integers(-10**20, 10**20, dtype=np.int64) if False else rng.integers(-2**62, 2**62)) * int(rng.integers(1, 9))
        t = str(v)
        if rng.random() < 0.1:
            t = "+" + t if v >= 0 else t
        if rng.random() < 0.1:
            t = "000" + t if v >= 0 else "-000" + t[1:]
        if rng.random() < 0.1:
            t = " \t" + t
        if rng.random() < 0.1:
            t = t + "\n "
        texts.append(t)
    texts += ["", "+", "-", "-0", "+0", "1 2", "1x", "x1", "1.0", "1e3", "\x0b1", "١٢"]
    got = oracle.cast(HostArray(A.Utf8, texts), dt)
    for t, v, ok in zip(texts, np.asarray(got.values), got.valid):
        s = t.strip(" \t\n\x0c\r")
        body = s[1:] if s[:1] in "+-" else s
        want_ok = len(body) > 0 and body.isascii() and body.isdigit() and info.min <= int(s) <= info.max
        assert bool(ok) == want_ok, (t, dt)
        if want_ok:
            assert int(v) == int(s), (t, dt)


def test_oracle_parse_nulls_and_unsafe(oracle):
    hv = HostArray(A.Utf8, ["1", "x", "", "4"], np.array([True, True, False, True]))
    safe = oracle.cast(hv, A.Int64)
    assert safe.to_pylist() == [1, None, None, 4] and safe.valid is not None
    assert np.asarray(safe.values).tolist() == [1, 0, 0, 4]  # failed / null slots hold 0
    with pytest.raises(A.array.CastError) as ei:
        oracle.cast(hv, A.Int64, safe=False)
    assert ei.value.message == "Cannot cast string 'x' to value of Int64 type"
    ok = HostArray(A.Utf8, ["1", "7", "", "4"], np.array([True, True, False, True]))
    un = oracle.cast(ok, A.Float32, safe=False)  # a null row is never parsed; the input nulls are cloned
    assert un.to_pylist() == [1.0, 7.0, None, 4.0]
    nn = oracle.cast(HostArray(A.Utf8, ["1", "2"]), A.UInt8, safe=False)
    assert nn.valid is None and nn.to_pylist() == [1, 2]
