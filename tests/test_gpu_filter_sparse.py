"""The wave-per-tile scatter for sparse selections (csrc/filter.hip: filter_scatter_sparse_kernel, taken when
K * 64 <= len; AH_FILTER_SPARSE=0 / 1 forces the choice).  Every case runs the sparse kernel AND the tiled kernel on the
same device arrays and compares both with the oracle (arrow-select/src/filter.rs:201,225,512-532,731-788):
values, validity, null count, null-buffer presence.  Sizes above 2^20 rows, or the predicate-object / record-batch /
coalescer entry points, so that the one-launch small-batch path is not what answers."""
import os

import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K
from orc import HostArray, assert_logical_eq, assert_same_nulls_presence
from test_gpu_filter_small import DTYPES, _values, _mask

pytestmark = pytest.mark.gpu


class _force:
    def __init__(self, value):
        self.value = value

    def __enter__(self):
        os.environ["AH_FILTER_SPARSE"] = self.value

    def __exit__(self, *exc):
        os.environ.pop("AH_FILTER_SPARSE", None)


def _profiled(ctx, fn):
    ctx.profile(True)
    ctx.profile_reset()
    try:
        return fn()
    finally:
        ctx.profile(False)


@pytest.mark.parametrize("seed", range(36))
def test_sparse_scatter_fuzz_against_tiled_and_oracle(ctx, oracle, seed):
    """predicate object -> filter (ah_filter_predicate_apply): every width incl. the 16- / 32-byte natives, nulls or none,
    a null-carrying predicate, bit offsets, selectivities from one row to everything (the forced sparse kernel must be
    right at ANY density, not only where the heuristic picks it)"""
    rng = np.random.default_rng(9700 + seed)
    n = [5, 4096, 4097, 70_003, 262_144 + 65, 1_500_007, 3_000_001][seed % 7]
    dt = DTYPES[seed % len(DTYPES)]
    h = _values(rng, dt, n, [None, 0.9, 0.3, 1.0][seed % 4])
    kind = ["0.001", "0.01", "0.0002", "one", "0.1", "runs", "none", "0.9", "all"][seed % 9]
    mask = HostArray(A.Boolean, _mask(rng, n, kind), (rng.random(n) < 0.9) if seed % 5 == 0 else None)
    exp = oracle.filter(h, mask)
    bo = int(rng.integers(0, 64)) if seed % 3 == 0 else 0
    dv, dm = h.to_device(ctx, bit_offset=bo), mask.to_device(ctx, bit_offset=(bo * 5) % 64)
    for force in ("1", "0"):
        with _force(force):
            pred = K.FilterBuilder(dm).optimize().build()
            assert pred.count() == len(exp)
            g = HostArray.from_device(pred.filter(dv))
        assert_logical_eq(g, exp, f"sparse={force} seed {seed} n {n} {dt} mask {kind}")
        assert_same_nulls_presence(g, exp, f"sparse={force} seed {seed}")


@pytest.mark.parametrize("seed", range(18))
def test_sparse_filter_boolean_against_tiled_and_oracle(ctx, oracle, seed):
    """filter_boolean (filter.rs:723-729 -> filter_bits :680-720) and the validity stream of every nullable filter through
    the BIT-ONLY form of the sparse kernel (round 4; the tiled W == 0 kernel was the only choice before): Boolean values
    with and without nulls, value / validity / predicate bit offsets, a null-carrying predicate, every density — forced
    sparse, forced tiled and the heuristic's own choice against the oracle, through the predicate object AND plain filter."""
    rng = np.random.default_rng(9900 + seed)
    n = [7, 4095, 4097, 70_003, 262_144 + 65, 1_500_007][seed % 6]
    pv = [None, 0.9, 0.3][seed % 3]
    h = HostArray(A.Boolean, rng.random(n) < [0.5, 0.02, 0.98][(seed // 3) % 3], None if pv is None else rng.random(n) < pv)
    kind = ["0.001", "0.01", "0.0002", "one", "0.1", "runs", "0.9"][seed % 7]
    mask = HostArray(A.Boolean, _mask(rng, n, kind), (rng.random(n) < 0.9) if seed % 4 == 0 else None)
    exp = oracle.filter(h, mask)
    bo = int(rng.integers(0, 64)) if seed % 2 == 0 else 0
    dv, dm = h.to_device(ctx, bit_offset=bo), mask.to_device(ctx, bit_offset=(bo * 7) % 64)
    for force in ("1", "0", None):
        ctxm = _force(force) if force is not None else _force("")
        with ctxm:
            if force is None:
                os.environ.pop("AH_FILTER_SPARSE", None)
            pred = K.FilterBuilder(dm).optimize().build()
            g = HostArray.from_device(pred.filter(dv))
            g2 = HostArray.from_device(K.filter(dv, dm))
        for label, got in (("predicate", g), ("filter", g2)):
            assert_logical_eq(got, exp, f"{label} sparse={force} seed {seed} n {n} mask {kind}")
            assert_same_nulls_presence(got, exp, f"{label} sparse={force} seed {seed}")


def test_sparse_kernel_is_what_runs_for_a_sparse_predicate(ctx, oracle):
    """the heuristic: 3 M rows, 0.1 % selected -> the sparse kernel (name in the kernel profile is the scatter's either
    way; the A/B below shows both give the same bytes); 10 % selected -> the tiled kernel.  Sliced (unaligned) values."""
    rng = np.random.default_rng(77)
    n = 3_000_000
    h = _values(rng, A.Int64, n + 3, 0.9)
    d = h.to_device(ctx).slice(3, n)
    hs = h.slice(3, n)
    for sel in (0.001, 0.1):
        mask = HostArray(A.Boolean, rng.random(n) < sel)
        exp = oracle.filter(hs, mask)
        dm = mask.to_device(ctx)
        got_default = HostArray.from_device(K.filter(d, dm))
        with _force("0"):
            got_tiled = HostArray.from_device(K.filter(d, dm))
        with _force("1"):
            got_sparse = HostArray.from_device(K.filter(d, dm))
        for label, g in (("default", got_default), ("tiled", got_tiled), ("sparse", got_sparse)):
            assert_logical_eq(g, exp, f"{label} sel {sel}")
            assert_same_nulls_presence(g, exp, f"{label} sel {sel}")


@pytest.mark.parametrize("seed", range(6))
def test_sparse_scatter_record_batch_and_coalescer(ctx, oracle, seed):
    """the multi-column launch (filter_record_batch above the small-batch limit) and the coalescer's windowed,
    NULL-counting scatters through the sparse kernel: a batch that straddles output batches is two windowed launches"""
    rng = np.random.default_rng(9800 + seed)
    n = [1_200_003, 2_000_000][seed % 2]
    dts = [A.Int64, A.Float64, A.Int64, A.Int32][:2 + seed % 3]
    pv = [0.9, None, 0.5] if seed % 2 else [0.9, 0.7, 0.5]  # (all-nullable batches take the coalescer's fused launches)
    cols = [_values(rng, dt, n, pv[i % 3]) for i, dt in enumerate(dts)]
    mask = HostArray(A.Boolean, rng.random(n) < [0.002, 0.01, 0.0005][seed % 3], (rng.random(n) < 0.97) if seed % 2 else None)
    rb = A.RecordBatch([f"c{i}" for i in range(len(cols))], [c.to_device(ctx) for c in cols], n)
    dm = mask.to_device(ctx)
    exps = [oracle.filter(c, mask) for c in cols]
    for force in ("1", "0"):
        with _force(force):
            out = K.filter_record_batch(rb, dm)
            for i, e in enumerate(exps):
                g = HostArray.from_device(out.columns[i])
                assert_logical_eq(g, e, f"record batch sparse={force} seed {seed} column {i}")
                assert_same_nulls_presence(g, e, f"record batch sparse={force} seed {seed} column {i}")
            # coalescer: target smaller than K so that pushes straddle output batches (windows), two pushes
            k = len(exps[0])
            target = max(1, k // 3 + 1)
            names = [f"c{i}" for i in range(len(cols))]
            for grouped in (False, True):  # single pushes (ah_filter_apply_into_acc_cols) / one grouped push (ah_filter_apply_multi)
                co = K.BatchCoalescer.new(names, dts, target, ctx)
                if grouped:
                    co.push_batches_with_filters([(rb, dm), (rb, dm)])
                else:
                    co.push_batch_with_filter(rb, dm)
                    co.push_batch_with_filter(rb, dm)
                co.finish_buffered_batch()
                got = [[] for _ in cols]
                while co.has_completed_batch():
                    b = co.next_completed_batch()
                    for i in range(len(cols)):
                        got[i].append(HostArray.from_device(b.columns[i]))
                label = f"coalescer sparse={force} grouped={grouped}"
                for i, e in enumerate(exps):
                    vals = np.concatenate([g.values for g in got[i]])
                    valid = np.concatenate([g.valid if g.valid is not None else np.ones(len(g), dtype=bool) for g in got[i]])
                    e_valid = e.valid if e.valid is not None else np.ones(len(e), dtype=bool)
                    assert len(vals) == 2 * k, label
                    assert np.array_equal(valid, np.concatenate([e_valid, e_valid])), f"{label} validity column {i}"
                    ev = np.concatenate([e.values, e.values])
                    assert np.array_equal(vals[valid], ev[valid]), f"{label} values column {i}"
                    for g in got[i]:  # null-buffer presence per output batch == has nulls (coalesce/primitive.rs finish)
                        assert (g.valid is None) == (g.null_count == 0), f"{label} null buffer presence column {i}"


def _strings(rng, n, maxlen=14):
    lens = rng.integers(0, maxlen, n)
    pool = rng.integers(97, 123, int(lens.sum()) + 1).astype(np.uint8).tobytes().decode()
    offs = np.concatenate([[0], np.cumsum(lens)])
    return [pool[offs[i]:offs[i + 1]] for i in range(n)]


@pytest.mark.parametrize("seed", range(10))
def test_sparse_string_filter_against_tiled_and_oracle(ctx, oracle, seed):
    """filter_bytes through string_filter_ranges_sparse_kernel (only the offsets of selected rows are read; validity bits
    through the wave's LDS strip, valid rows counted from the output bitmap afterwards) and through the tiled ranges
    kernel, both against the oracle (filter.rs:790-928, :512-532): Utf8 / LargeUtf8, nulls or none, a null-carrying
    predicate, one row ... every row, lengths around the 4096-row tile, a sliced column (offsets not starting at 0)"""
    rng = np.random.default_rng(9900 + seed)
    n = [1, 4095, 4097, 50_003, 200_001][seed % 5]
    dt = [A.Utf8, A.LargeUtf8][seed % 2]
    sv = _strings(rng, n + 7)
    full = HostArray(dt, sv, (rng.random(n + 7) < 0.85) if seed % 3 else None)
    off = [0, 3, 7][seed % 3]
    h = full.slice(off, n)
    kind = ["0.001", "0.02", "one", "0.3", "none", "all", "runs"][seed % 7]
    mask = HostArray(A.Boolean, _mask(rng, n, kind), (rng.random(n) < 0.9) if seed % 4 == 0 else None)
    exp = oracle.filter(h, mask)
    d = full.to_device(ctx).slice(off, n)
    dm = mask.to_device(ctx)
    for force in ("1", "0"):
        with _force(force):
            got = K.filter(d, dm)
            g = HostArray.from_device(got)
        assert_logical_eq(g, exp, f"string sparse={force} seed {seed} n {n} {dt} mask {kind}")
        assert_same_nulls_presence(g, exp, f"string sparse={force} seed {seed}")
        assert got.null_count() == exp.null_count
