"""The one-launch filter path for query-engine-sized batches (csrc/filter_small.hip: predicates of at most 2^20 rows,
fixed-width columns): count + prefix + scatter in ONE kernel, validity words written once by their owner tile with a
forward gather, completion through a packed ticket.  Every case runs BOTH paths — the one-launch kernel and, with
AH_FILTER_SMALL=0, the general two-pass path — against the oracle (arrow-select/src/filter.rs:201,225,512-532,731-788)."""
import os

import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K
from orc import HostArray, assert_logical_eq, assert_same_nulls_presence

pytestmark = pytest.mark.gpu

DTYPES = [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.Float32, A.Float64, A.Decimal128(20, 2), A.Decimal256(40, 3)]


class _general_path:
    def __enter__(self):
        os.environ["AH_FILTER_SMALL"] = "0"

    def __exit__(self, *exc):
        os.environ.pop("AH_FILTER_SMALL", None)


def _values(rng, dt, n, p_valid):
    npdt = np.dtype(dt.np_dtype)
    if npdt.fields is not None or npdt.kind == "V":  # 16 / 32-byte natives: random bytes
        raw = rng.integers(0, 256, n * npdt.itemsize, dtype=np.uint8).view(npdt)
        v = raw
    elif np.issubdtype(npdt, np.floating):
        v = rng.standard_normal(n).astype(npdt)
    else:
        info = np.iinfo(npdt)
        v = rng.integers(info.min, info.max, n, dtype=npdt, endpoint=True)
    return HostArray(dt, v, None if p_valid is None else rng.random(n) < p_valid)


def _mask(rng, n, kind):
    if kind == "none":
        m = np.zeros(n, dtype=bool)
    elif kind == "all":
        m = np.ones(n, dtype=bool)
    elif kind == "one":
        m = np.zeros(n, dtype=bool)
        if n:
            m[int(rng.integers(0, n))] = True
    elif kind == "runs":  # long selected runs and long gaps: tiles with 0 rows, tiles with every row
        m = np.zeros(n, dtype=bool)
        pos = 0
        while pos < n:
            ln = int(rng.integers(1, 9000))
            if rng.random() < 0.5:
                m[pos:pos + ln] = True
            pos += ln
    else:
        m = rng.random(n) < float(kind)
    return m


SIZES = [1, 2, 63, 64, 65, 1000, 4095, 4096, 4097, 8192, 10_000, 65_536, 100_003, 262_144, 1_000_000, 1 << 20]


@pytest.mark.parametrize("seed", range(48))
def test_small_filter_fuzz_both_paths(ctx, oracle, seed):
    rng = np.random.default_rng(9100 + seed)
    n = SIZES[seed % len(SIZES)]
    dt = DTYPES[(seed // 2) % len(DTYPES)]
    h = _values(rng, dt, n, [None, 0.9, 0.3, 1.0][seed % 4])
    kind = ["0.1", "0.5", "0.9", "0.01", "runs", "one", "none", "all", "0.003"][seed % 9]
    mv = _mask(rng, n, kind)
    mask = HostArray(A.Boolean, mv, (rng.random(n) < 0.9) if seed % 5 == 0 else None)
    exp = oracle.filter(h, mask)
    bo = int(rng.integers(0, 64)) if seed % 3 == 0 else 0
    dv, dm = h.to_device(ctx, bit_offset=bo), mask.to_device(ctx, bit_offset=(bo * 7) % 64)
    for label, cm in (("one-launch", None), ("general", _general_path())):
        if cm:
            with cm:
                got = K.filter(dv, dm)
        else:
            got = K.filter(dv, dm)
        g = HostArray.from_device(got)
        assert_logical_eq(g, exp, f"{label} seed {seed} n {n} {dt} mask {kind}")
        assert_same_nulls_presence(g, exp, f"{label} seed {seed}")


@pytest.mark.parametrize("seed", range(12))
def test_small_filter_sliced_and_short_predicate(ctx, oracle, seed):
    """values sliced at an odd element (unaligned for 16-byte loads), validity at a bit offset, predicate shorter than values"""
    rng = np.random.default_rng(9300 + seed)
    n = [5000, 70_001, 300_000][seed % 3]
    dt = [A.Int64, A.Int32, A.Int16, A.Float64][seed % 4]
    h = _values(rng, dt, n + 5, 0.8)
    off = 1 + seed % 3
    hs = h.slice(off, n)
    plen = n - (seed % 2) * 777
    mask = HostArray(A.Boolean, rng.random(plen) < 0.2)
    exp = oracle.filter(hs, mask)
    d = h.to_device(ctx).slice(off, n)
    dm = mask.to_device(ctx)
    assert_logical_eq(HostArray.from_device(K.filter(d, dm)), exp, f"sliced seed {seed}")
    with _general_path():
        assert_logical_eq(HostArray.from_device(K.filter(d, dm)), exp, f"sliced general seed {seed}")


@pytest.mark.parametrize("seed", range(10))
def test_small_filter_record_batch_mixed_columns(ctx, oracle, seed):
    """filter_record_batch: columns of several widths / nullability shapes leave in one launch per shape and ONE wait;
    a Boolean or Utf8 column sends the whole batch down the general path — same results either way"""
    rng = np.random.default_rng(9500 + seed)
    n = [8192, 8191, 65_536, 1000, 500_000][seed % 5]
    dts = [A.Int64, A.Float64, A.Int32, A.Int64, A.Int8, A.Float32, A.Decimal128(10, 1), A.Int16, A.Int64, A.Float64][:3 + seed]
    cols = [_values(rng, dt, n, [0.9, None, 0.5, 1.0][i % 4]) for i, dt in enumerate(dts)]
    if seed % 4 == 3:
        cols.append(HostArray(A.Boolean, rng.random(n) < 0.5, rng.random(n) < 0.9))
    mask = HostArray(A.Boolean, _mask(rng, n, ["0.1", "0.5", "runs", "0.9"][seed % 4]), (rng.random(n) < 0.95) if seed % 2 else None)
    rb = A.RecordBatch([f"c{i}" for i in range(len(cols))], [c.to_device(ctx) for c in cols], n)
    dm = mask.to_device(ctx)
    for label, cm in (("one-launch", None), ("general", _general_path())):
        if cm:
            with cm:
                out = K.filter_record_batch(rb, dm)
        else:
            out = K.filter_record_batch(rb, dm)
        exp_rows = int(np.count_nonzero(mask.values & (mask.valid if mask.valid is not None else True)))
        assert out.num_rows() == exp_rows
        for i, c in enumerate(cols):
            e = oracle.filter(c, mask)
            g = HostArray.from_device(out.columns[i])
            assert_logical_eq(g, e, f"{label} seed {seed} column {i} {c.data_type}")
            assert_same_nulls_presence(g, e, f"{label} seed {seed} column {i}")


def test_small_filter_many_calls_leave_tickets_clean(ctx, oracle):
    """the completion tickets live in the context's self-cleaning scratch: 300 back-to-back calls of changing shapes"""
    rng = np.random.default_rng(5)
    for it in range(300):
        n = int(rng.integers(1, 40_000))
        ncols = int(rng.integers(1, 6))
        cols = [_values(rng, [A.Int64, A.Int32, A.Float64][i % 3], n, [0.9, None][(i + it) % 2]) for i in range(ncols)]
        mask = HostArray(A.Boolean, rng.random(n) < rng.random())
        rb = A.RecordBatch([f"c{i}" for i in range(ncols)], [c.to_device(ctx) for c in cols], n)
        out = K.filter_record_batch(rb, mask.to_device(ctx))
        for i, c in enumerate(cols):
            assert_logical_eq(HostArray.from_device(out.columns[i]), oracle.filter(c, mask), f"iteration {it} column {i}")
