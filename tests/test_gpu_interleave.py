"""arrow_select::interleave on the device vs the oracle and the reference's tests (arrow-select/src/interleave.rs:940-965)."""
import zlib

import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K
import orc
from orc import HostArray

pytestmark = pytest.mark.gpu


def test_reference_cases(ctx):
    i32 = lambda xs: A.Array.from_pylist(xs, A.Int32, ctx=ctx)  # noqa: E731
    a, b, c = i32([1, 2, 3, 4]), i32([5, 6, 7]), i32([8, 9, 10])
    v = K.interleave([a, b, c], [(0, 3), (0, 3), (2, 2), (2, 0), (1, 1)])          # test_primitive :941
    assert v.to_pylist() == [4, 4, 10, 8, 6] and v.nulls() is None
    b2 = i32([1, 4, None])
    v = K.interleave([a, b2], [(0, 1), (1, 2), (1, 2), (0, 3), (0, 2)])             # test_primitive_nulls :951
    assert v.to_pylist() == [2, None, None, 4, 3]
    e = K.interleave([a], [])                                                       # test_primitive_empty :960
    assert e.length == 0 and e.data_type == A.Int32
    with pytest.raises(A.array.InvalidArgumentError, match="interleave requires input of at least one array"):
        K.interleave([], [(0, 0)])
    with pytest.raises(A.array.InvalidArgumentError, match=r"different data types \(Int32 and Int64\)"):
        K.interleave([a, A.Array.from_pylist([1], A.Int64, ctx=ctx)], [(0, 0)])
    with pytest.raises(A.Panic, match="index out of bounds: the len is 3 but the index is 3"):
        K.interleave([a, b], [(0, 0), (1, 3)])
    with pytest.raises(A.Panic, match="index out of bounds: the len is 2 but the index is 2"):
        K.interleave([a, b], [(2, 0)])


@pytest.mark.parametrize("dt", [A.Int8, A.Int32, A.Int64, A.Float64, A.Boolean, A.Decimal128(20, 2)], ids=repr)
def test_interleave_fuzz(ctx, oracle, dt):
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

    for k in (1, 2, 7):
        for nullable in (False, True):
            hs = []
            for j in range(k):
                n = int(rng.integers(1, 5000))
                hs.append(HostArray(dt, vals(n), (rng.random(n) < 0.8) if (nullable and j % 2 == 0) else None))
            for m in (1, 64, 65, 30_001):
                pairs = [(int(a), int(rng.integers(0, len(hs[a])))) for a in rng.integers(0, k, m)]
                want = oracle.interleave(hs, pairs)
                got = K.interleave([h.to_device(ctx, bit_offset=j % 5) for j, h in enumerate(hs)], pairs)
                orc.assert_logical_eq(HostArray.from_device(got), want, f"{dt} k={k} m={m} nullable={nullable}")
                assert (got.nulls() is None) == (want.valid is None) and got.null_count() == want.null_count


def test_merge_two_sorted_runs(ctx, oracle):
    """The use the kernel exists for: merge step = interleave of sorted runs by (run, row) pairs computed elsewhere;
    also interleave_record_batch (:912)."""
    rng = np.random.default_rng(2)
    a = np.sort(rng.integers(0, 10**6, 50_000))
    b = np.sort(rng.integers(0, 10**6, 70_000))
    order = np.argsort(np.concatenate([a, b]), kind="stable")
    pairs = [(0, int(i)) if i < len(a) else (1, int(i - len(a))) for i in order]
    da, db = A.Array.from_numpy(a, ctx=ctx), A.Array.from_numpy(b, ctx=ctx)
    merged = K.interleave([da, db], pairs)
    assert np.array_equal(merged.values_numpy(), np.sort(np.concatenate([a, b])))
    rb = K.interleave_record_batch([A.RecordBatch(["k", "k2"], [da, da]), A.RecordBatch(["k", "k2"], [db, db])], pairs)
    assert rb.num_rows() == 120_000 and np.array_equal(rb.columns[1].values_numpy(), merged.values_numpy())
