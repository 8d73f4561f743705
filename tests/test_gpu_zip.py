"""arrow_select::zip on the device vs the oracle and the reference's own tests (arrow-select/src/zip.rs:870-1064)."""
import zlib

import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K
import orc
from orc import HostArray, golden_array, load_golden
from test_oracle_golden import ERR

pytestmark = pytest.mark.gpu


class _RawDatum:
    """a `Datum` that claims to be a scalar whatever its length (Scalar::new asserts len == 1, scalar.rs:117;
    the reference's length check is only reachable through such a custom Datum)"""

    def __init__(self, array):
        self.array = array

    def get(self):
        return self.array, True


def _datum(h, scalar, ctx, off=0):
    d = h.to_device(ctx, off)
    if not scalar:
        return d
    return A.Scalar(d) if d.length == 1 else _RawDatum(d)


@pytest.mark.parametrize("case", load_golden("zip"), ids=lambda c: c["name"])
@pytest.mark.parametrize("bit_offset", [0, 3])
def test_reference_goldens(ctx, case, bit_offset):
    m = golden_array(case["mask"]).to_device(ctx, bit_offset)
    t = _datum(golden_array(case["truthy"]), case.get("truthy_scalar", False), ctx, bit_offset)
    f = _datum(golden_array(case["falsy"]), case.get("falsy_scalar", False), ctx, bit_offset)
    if "error" in case:
        with pytest.raises(ERR[case["error"]]) as e:
            K.zip(m, t, f)
        assert e.value.message == case["message"]
        return
    orc.assert_logical_eq(HostArray.from_device(K.zip(m, t, f)), golden_array(case["expected"]), case["name"])


@pytest.mark.parametrize("dt", [A.Int8, A.Int16, A.Int32, A.Int64, A.Float64, A.Boolean, A.Decimal128(20, 2)], ids=repr)
def test_zip_fuzz(ctx, oracle, dt):
    rng = np.random.default_rng(zlib.crc32(dt.name.encode()))

    def vals(n):
        if dt == A.Boolean:
            return rng.random(n) < 0.5
        if dt.physical == A._lib.AH_FIXED16:
            v = np.zeros(n, dtype=dt.np_dtype)
            v["lo"] = rng.integers(0, 2**63, n, dtype=np.uint64)
            v["hi"] = rng.integers(-2**40, 2**40, n)
            return v
        npdt = np.dtype(dt.np_dtype)
        if npdt.kind == "f":
            return rng.standard_normal(n)
        info = np.iinfo(npdt)
        return rng.integers(info.min, info.max, n, dtype=npdt, endpoint=True)

    for n in (1, 63, 64, 65, 1000, 70_001):
        for p_mask in (0.0, 0.02, 0.5, 1.0):
            mask = HostArray(A.Boolean, rng.random(n) < p_mask, (rng.random(n) < 0.9) if n % 2 else None)
            for ts, fs in ((False, False), (True, False), (False, True), (True, True)):
                for nullable in (False, True):
                    tn, fn = (1 if ts else n), (1 if fs else n)
                    t = HostArray(dt, vals(tn), (rng.random(tn) < 0.7) if nullable else None)
                    f = HostArray(dt, vals(fn), (rng.random(fn) < 0.7) if nullable and n % 3 else None)
                    want = oracle.zip(mask, t, f, ts, fs)
                    got = K.zip(mask.to_device(ctx, 5), _datum(t, ts, ctx, 1), _datum(f, fs, ctx, 2))
                    orc.assert_logical_eq(HostArray.from_device(got), want, f"{dt} n={n} p={p_mask} {ts} {fs} {nullable}")
                    assert (got.nulls() is None) == (want.valid is None) and got.null_count() == want.null_count
    s = HostArray(dt, vals(500), rng.random(500) < 0.8)
    m = HostArray(A.Boolean, rng.random(500) < 0.5)
    got = K.zip(m.to_device(ctx).slice(100, 300), s.to_device(ctx, 3).slice(7, 300), s.to_device(ctx).slice(200, 300))
    orc.assert_logical_eq(HostArray.from_device(got), oracle.zip(m.slice(100, 300), s.slice(7, 300), s.slice(200, 300)), "sliced")
    with pytest.raises(A.array.NotYetImplemented):
        K.zip(m.to_device(ctx), A.Array.from_strings(["a"] * 500, ctx=ctx), A.Array.from_strings(["b"] * 500, ctx=ctx))


def test_case_when_pipeline(ctx, oracle):
    """CASE WHEN a < 0 THEN -a ELSE b END, entirely in HBM."""
    n = 1 << 22
    a = HostArray(A.Int64, oracle.gen_i64(n, 1, -1000, 1000), oracle.gen_bits(n, 2, 0.9))
    b = HostArray(A.Int64, oracle.gen_i64(n, 3, -1000, 1000))
    da, db = a.to_device(ctx), b.to_device(ctx)
    got = K.zip(K.lt(da, A.Scalar.new(0, A.Int64, ctx)), K.neg_wrapping(da), db)
    cond = (a.values < 0) & a.valid  # null condition -> ELSE
    want = np.where(cond, -a.values, b.values)
    assert np.array_equal(got.values_numpy()[got.valid_mask()], want[got.valid_mask()])
    assert got.null_count() == 0 or np.array_equal(~got.valid_mask(), cond & ~a.valid)
