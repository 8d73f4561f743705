"""The lazy predicate (csrc/filter_expr.hip, `ah_filter_predicate_build_expr`): `filter(v, join(cmp(..), cmp(..), ...))`
with the comparisons evaluated inside the filter's count pass, against the oracle's MATERIALISED chain
(oracle.compare -> oracle.boolean_binary -> oracle.filter = arrow-ord/src/cmp.rs:220-382, arrow-arith/src/boolean.rs:60-300,
arrow-select/src/filter.rs:167-171,201).  Bit-exact: values, validity, null count, selected-row count."""
import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K
import orc
from orc import HostArray, assert_logical_eq, assert_same_nulls_presence

pytestmark = pytest.mark.gpu

OPS = {"eq": 0, "neq": 1, "lt": 2, "lt_eq": 3, "gt": 4, "gt_eq": 5}
JOINS = {"and": 0, "or": 1, "and_kleene": 3, "or_kleene": 4}
TYPES = [A.Int8, A.Int16, A.Int32, A.Int64, A.UInt8, A.UInt16, A.UInt32, A.UInt64, A.Float32, A.Float64]


def _column(rng, dt, n, p_valid):
    npdt = dt.np_dtype
    if np.issubdtype(npdt, np.floating):
        v = (rng.standard_normal(n) * 3).astype(npdt)
        if n:
            k = max(1, n // 50)
            pos = rng.integers(0, n, k)
            v[pos] = rng.choice(np.array([np.nan, -np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0], dtype=npdt), k)
        v = np.round(v) if rng.random() < 0.5 else v  # ties, so eq / lt_eq / gt_eq differ from lt / gt
    else:
        info = np.iinfo(npdt)
        lo, hi = (max(info.min, -4), min(info.max, 4)) if rng.random() < 0.5 else (info.min, info.max)
        v = rng.integers(lo, hi, n, dtype=npdt, endpoint=True)
    valid = None if p_valid is None else rng.random(n) < p_valid
    return HostArray(dt, v, valid)


def _scalar(rng, dt, null=False):
    h = _column(rng, dt, 1, None)
    if null:
        h = HostArray(dt, h.values, np.array([False]))
    return h


def _oracle_mask(oracle, terms, joins):
    acc = None
    for i, (op, l, ls, r, rs) in enumerate(terms):
        m = oracle.compare(OPS[op], l, r, l_scalar=ls, r_scalar=rs)
        acc = m if acc is None else oracle.boolean_binary(JOINS[joins[i - 1]], acc, m)
    return acc


def _device_predicate(ctx, terms, joins, bit_offset=0):
    dts = []
    for op, l, ls, r, rs in terms:
        dl = l.to_device(ctx, bit_offset=0 if ls else bit_offset)
        dr = r.to_device(ctx, bit_offset=0 if rs else bit_offset)
        dts.append((op, A.Scalar(dl) if ls else dl, A.Scalar(dr) if rs else dr))
    return K.FilterBuilder.from_terms(dts, joins).build()


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_lazy_predicate_equals_materialised_chain(ctx, oracle, seed):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([0, 1, 2, 63, 64, 65, 127, 129, 1023, 1024, 1025, 4097, 65535, 65537, 131073, 300_001]))
    nterms = int(rng.integers(1, 5))
    same_type = rng.random() < 0.6  # one width for every term: the fast instantiations (8- and 4-byte types)
    dt0 = TYPES[int(rng.integers(0, len(TYPES)))] if not same_type else [A.Int64, A.Float64, A.Int32, A.Float32, A.UInt64][seed % 5]
    terms, joins = [], []
    all_scalar_rhs = rng.random() < 0.5
    for k in range(nterms):
        dt = dt0 if same_type else TYPES[int(rng.integers(0, len(TYPES)))]
        pv = [None, 0.9, 0.5, 1.0][int(rng.integers(0, 4))]
        l = _column(rng, dt, n, pv)
        if all_scalar_rhs or rng.random() < 0.3:
            r, rs = _scalar(rng, dt, null=rng.random() < 0.1), True
        else:
            r, rs = _column(rng, dt, n, [None, 0.8][int(rng.integers(0, 2))]), False
        ls = False
        if not rs and rng.random() < 0.1:  # scalar on the LEFT: the generic instantiation
            l, ls = _scalar(rng, dt), True
        terms.append((list(OPS)[int(rng.integers(0, 6))], l, ls, r, rs))
        if k:
            joins.append(list(JOINS)[int(rng.integers(0, 4))])
    values = _column(rng, [A.Int64, A.Float64, A.Int16, A.Int8][seed % 4], n, [None, 0.85][seed % 2])
    mask = _oracle_mask(oracle, terms, joins)
    exp = oracle.filter(values, mask)
    pred = _device_predicate(ctx, terms, joins, bit_offset=int(rng.integers(0, 64)) if rng.random() < 0.5 else 0)
    sel = mask.values & (mask.valid if mask.valid is not None else True)
    assert pred.count() == int(np.count_nonzero(sel)), "FilterPredicate::count"
    got = pred.filter(values.to_device(ctx))
    assert_logical_eq(HostArray.from_device(got), exp, f"seed {seed} {[(t[0]) for t in terms]} {joins}")
    assert_same_nulls_presence(HostArray.from_device(got), exp, f"seed {seed}")
    # the same predicate object applied to a second column and to a Boolean column
    v2 = _column(rng, A.Float32, n, 0.7)
    assert_logical_eq(HostArray.from_device(pred.filter(v2.to_device(ctx))), oracle.filter(v2, mask), f"seed {seed} second column")
    vb = HostArray(A.Boolean, rng.random(n) < 0.5, rng.random(n) < 0.9)
    assert_logical_eq(HostArray.from_device(pred.filter(vb.to_device(ctx))), oracle.filter(vb, mask), f"seed {seed} bool column")


@pytest.mark.parametrize("dt", [A.Int64, A.Float64, A.Int32, A.UInt32], ids=str)
def test_sliced_operands_take_the_unaligned_path(ctx, oracle, dt):
    """`values` pointers advanced by an odd number of elements are not aligned for two-element vector loads."""
    rng = np.random.default_rng(11)
    n = 70_001
    a, b = _column(rng, dt, n + 3, 0.9), _column(rng, dt, n + 3, 0.8)
    da, db = a.to_device(ctx).slice(3, n), b.to_device(ctx).slice(1, n)
    ha, hb = a.slice(3, n), b.slice(1, n)
    s = _scalar(rng, dt)
    pred = K.FilterBuilder.from_terms([("lt", da, db), ("gt_eq", da, A.Scalar(s.to_device(ctx)))], ["or_kleene"]).build()
    mask = oracle.boolean_binary(JOINS["or_kleene"], oracle.compare(OPS["lt"], ha, hb), oracle.compare(OPS["gt_eq"], ha, s, r_scalar=True))
    assert_logical_eq(HostArray.from_device(pred.filter(da)), oracle.filter(ha, mask), "sliced")


def test_where_a_lt_0_and_b_ge_0_the_bench_shape(ctx, oracle):
    """bench.py's predicate_filter_fused on 3 M rows: Int64 a, Float64 b, both with NullBuffers, scalar literals."""
    rng = np.random.default_rng(3)
    n = 3_000_017
    a = HostArray(A.Int64, rng.integers(-2**63, 2**63 - 1, n), rng.random(n) < 0.9)
    b = HostArray(A.Float64, rng.uniform(-1e6, 1e6, n), rng.random(n) < 0.9)
    s0, s1 = HostArray(A.Int64, np.array([0])), HostArray(A.Float64, np.array([0.0]))
    da, db = a.to_device(ctx), b.to_device(ctx)
    pred = K.FilterBuilder.from_terms([("lt", da, A.Scalar(s0.to_device(ctx))), ("gt_eq", db, A.Scalar(s1.to_device(ctx)))],
                                      ["and_kleene"]).build()
    mask = oracle.boolean_binary(JOINS["and_kleene"], oracle.compare(OPS["lt"], a, s0, r_scalar=True),
                                 oracle.compare(OPS["gt_eq"], b, s1, r_scalar=True))
    exp = oracle.filter(a, mask)
    got = pred.filter(da)
    assert_logical_eq(HostArray.from_device(got), exp, "bench shape")
    assert got.null_count() == exp.null_count
    # and it equals the device's own materialised chain
    m = K.and_kleene(K.lt(da, A.Scalar(s0.to_device(ctx))), K.gt_eq(db, A.Scalar(s1.to_device(ctx))))
    assert_logical_eq(HostArray.from_device(K.filter(da, m)), exp, "materialised chain")


def test_lazy_predicate_error_texts(ctx):
    a = HostArray(A.Int64, np.arange(10)).to_device(ctx)
    b = HostArray(A.Int64, np.arange(11)).to_device(ctx)
    f = HostArray(A.Float64, np.arange(10.0)).to_device(ctx)
    with pytest.raises(A.InvalidArgumentError, match="Cannot compare arrays of different lengths, got 10 vs 11"):
        K.FilterBuilder.from_terms([("lt", a, b)]).build()
    with pytest.raises(A.InvalidArgumentError, match="Invalid comparison operation: Int64 < Float64"):
        K.FilterBuilder.from_terms([("lt", a, f)]).build()
    with pytest.raises(A.ComputeError, match="Cannot perform bitwise operation on arrays of different length"):
        K.FilterBuilder.from_terms([("lt", a, a), ("gt", b, b)], ["and"]).build()
    with pytest.raises(A.InvalidArgumentError, match="1..4 comparison terms"):
        K.FilterBuilder.from_terms([("lt", a, a)] * 5, ["and"] * 4).build()
    s = HostArray.from_pylist(["x"], A.Utf8).to_device(ctx)
    with pytest.raises(A.NotYetImplemented):
        K.FilterBuilder.from_terms([("eq", s, s)]).build()
    # an empty expression predicate selects nothing and filters like FilterBuilder on an empty mask
    e = HostArray(A.Int64, np.zeros(0, dtype=np.int64)).to_device(ctx)
    p = K.FilterBuilder.from_terms([("lt", e, e)]).build()
    assert p.count() == 0 and p.filter(e).length == 0


# ---------------------------------------------------------------------------------- the one-call form (ah_filter_expr)
def _dev_terms(ctx, terms):
    out = []
    for op, l, ls, r, rs in terms:
        dl, dr = l.to_device(ctx), r.to_device(ctx)
        out.append((op, A.Scalar(dl) if ls else dl, A.Scalar(dr) if rs else dr))
    return out


@pytest.mark.parametrize("seed", range(14))
def test_filter_expr_millions_of_rows_equals_materialised_chain(ctx, oracle, seed):
    """ah_filter_expr beyond the small-batch sizes (1 - 5 M rows of 8-byte columns, several count groups): against
    oracle.filter(values, fold(oracle.compare ...)): count, values, validity, null-buffer presence.  Shapes: 1-4 terms,
    scalar or array right sides, values column that is / is not an operand, nulls everywhere, selectivity from nothing to
    everything, odd lengths."""
    rng = np.random.default_rng(8800 + seed)
    n = [1_048_577, 1_200_001, 2_500_003, 4_194_304 + 4097, 3_000_017, 1_300_000, 5_000_011][seed % 7]
    dts = [A.Int64, A.Float64, A.UInt64]
    nterms = 1 + seed % 4
    scalar_rhs = seed % 2 == 0
    vals = _column(rng, [A.Int64, A.Float64][seed % 2], n, [0.9, None, 0.5][seed % 3])
    terms, joins = [], []
    for k in range(nterms):
        dt = dts[(seed + k) % 3]
        l = vals if (k == 0 and seed % 3 != 2 and dt == vals.data_type) else _column(rng, dt, n, [None, 0.8][k % 2])
        if l is vals:
            dt = vals.data_type
        if scalar_rhs:
            r, rs = _scalar(rng, dt, null=(seed == 6 and k == 1)), True
        else:
            r, rs = _column(rng, dt, n, [None, 0.9][(k + seed) % 2]), False
        op = ["lt", "gt_eq", "neq", "eq", "lt_eq", "gt"][(seed + k) % 6]
        terms.append((op, l, False, r, rs))
        if k:
            joins.append(["and_kleene", "or", "and", "or_kleene"][(seed + k) % 4])
    if seed == 4:  # nothing selected / everything selected
        terms, joins = [("lt", vals, False, HostArray(vals.data_type, np.array([np.iinfo(np.int64).min])), True)], []
    if seed == 5:
        vals = HostArray(A.Float64, np.abs(vals.values), None)
        terms, joins = [("gt_eq", vals, False, HostArray(A.Float64, np.array([-1.0])), True)], []
    mask = _oracle_mask(oracle, terms, joins)
    exp = oracle.filter(vals, mask)
    dvals = vals.to_device(ctx)
    dterms = []
    for op, l, ls, r, rs in terms:
        dl = dvals if l is vals else l.to_device(ctx)
        dr = r.to_device(ctx)
        dterms.append((op, dl, A.Scalar(dr) if rs else dr))
    got = K.filter_expr(dvals, dterms, joins)
    g = HostArray.from_device(got)
    assert_logical_eq(g, exp, f"filter_expr seed {seed} n {n} {[t_[0] for t_ in terms]} {joins}")
    assert_same_nulls_presence(g, exp, f"filter_expr seed {seed}")
    # and through the predicate object
    pred = K.FilterBuilder.from_terms(dterms, joins).build()
    assert pred.count() == len(exp)
    assert_logical_eq(HostArray.from_device(pred.filter(dvals)), exp, f"two pass seed {seed}")


def test_filter_expr_other_widths_and_sizes(ctx, oracle):
    rng = np.random.default_rng(1)
    for n, dt in ((1000, A.Int64), (70_000, A.Int32), (2_000_000, A.Int32), (2_000_001, A.Int16)):
        a, b = _column(rng, dt, n, 0.9), _column(rng, dt, n, None)
        s = _scalar(rng, dt)
        da, db = a.to_device(ctx), b.to_device(ctx)
        got = K.filter_expr(da, [("lt", da, db), ("gt", db, A.Scalar(s.to_device(ctx)))], ["or"])
        mask = oracle.boolean_binary(JOINS["or"], oracle.compare(OPS["lt"], a, b), oracle.compare(OPS["gt"], b, s, r_scalar=True))
        assert_logical_eq(HostArray.from_device(got), oracle.filter(a, mask), f"n {n} {dt}")


# ------------------------------------------------------- ah_filter_expr around the chunk / tile / group boundaries
def _boundary_case(rng, seed):
    """values column = the array operand of one term; selectivity from nothing to everything, dense stretches"""
    n = [1, 2, 63, 1023, 1024, 1025, 4095, 4097, 8191, 65_536 + 1, 70_001, 262_147, 1_000_003][seed % 13]
    dt = [A.Int64, A.Float64, A.Int32, A.UInt64, A.Float32, A.UInt32][seed % 6]
    p_valid = [0.9, None, 0.5, 0.999][seed % 4]
    npdt = dt.np_dtype
    if np.issubdtype(npdt, np.floating):
        v = rng.standard_normal(n).astype(npdt)
        if n > 10:
            v[rng.integers(0, n, max(1, n // 40))] = np.nan
    elif np.issubdtype(npdt, np.signedinteger):
        v = rng.integers(-1000, 1000, n).astype(npdt)
    else:
        v = rng.integers(0, 2000, n).astype(npdt)
    vals = HostArray(dt, v, None if p_valid is None else rng.random(n) < p_valid)
    # threshold chosen for a target selectivity; clustered variant: a stretch of the column selects everything
    sel = [0.02, 0.2, 0.45, 0.7, 0.0, 1.0, 0.3][seed % 7]
    finite = v[np.isfinite(v.astype(np.float64))] if n else v
    thr = np.quantile(finite.astype(np.float64), sel) if len(finite) else 0
    thr_h = HostArray(dt, np.array([thr]).astype(npdt))
    terms = [("lt", vals, False, thr_h, True)]
    joins = []
    if seed % 3 != 0:
        other = _column(rng, dt, n, [None, 0.8][seed % 2])
        if seed % 2:
            terms.append(("gt_eq", other, False, _scalar(rng, dt), True))
        else:  # array right sides everywhere (the RS = false instantiation)
            terms = [("lt", vals, False, HostArray(dt, np.full(n, thr).astype(npdt)), False),
                     ("neq", other, False, _column(rng, dt, n, None), False)]
        joins.append(["and_kleene", "or_kleene", "and", "or"][seed % 4])
        if seed % 5 == 0:  # the values column as the SECOND term's operand
            terms.reverse()
    if seed % 11 == 7 and n > 5000:  # one dense stretch: a few chunks overflow their slots, the rest are sparse
        v2 = v.copy()
        v2[2048:2048 + 3000] = (thr - 1) if not np.issubdtype(npdt, np.unsignedinteger) else 0
        vals2 = HostArray(dt, v2, vals.valid)
        terms = [(op, vals2 if l is vals else l, ls, r, rs) for op, l, ls, r, rs in terms]
        vals = vals2
    return vals, terms, joins


@pytest.mark.parametrize("seed", range(52))
def test_filter_expr_boundary_sizes_fuzz(ctx, oracle, seed):
    """ah_filter_expr where the filtered column is an operand of the expression: against the oracle's materialised chain —
    every size class around the chunk (1024) / tile (4096) / count-group (65 536) boundaries and the small-batch limit,
    8- and 4-byte types, with and without nulls, scalar and array right sides, the column as first or second term, a dense
    stretch inside a sparse selection, nothing / everything selected.  (Written for a variant of the count pass that
    stashed the selected values — profiles/r03_single_pass_lookback.md — and kept for the shapes.)"""
    rng = np.random.default_rng(9900 + seed)
    vals, terms, joins = _boundary_case(rng, seed)
    mask = _oracle_mask(oracle, terms, joins)
    exp = oracle.filter(vals, mask)
    dvals = vals.to_device(ctx)
    dterms = []
    for op, l, ls, r, rs in terms:
        dl = dvals if l is vals else l.to_device(ctx)
        dr = r.to_device(ctx)
        dterms.append((op, dl, A.Scalar(dr) if rs else dr))
    got = K.filter_expr(dvals, dterms, joins)
    g = HostArray.from_device(got)
    assert_logical_eq(g, exp, f"seed {seed} n {len(vals)} {vals.data_type} {[t_[0] for t_ in terms]} {joins}")
    assert_same_nulls_presence(g, exp, f"seed {seed}")
    assert got.null_count() == exp.null_count


def test_shrink_to_fit_after_a_small_batch_filter(ctx, oracle):
    """the one-launch filter allocates for the worst case; ah_array_shrink_to_fit copies into an exact-size buffer"""
    import ctypes as C
    rng = np.random.default_rng(2)
    n = 1_000_003
    a = HostArray(A.Int64, rng.integers(-100, 100, n), rng.random(n) < 0.9)
    m = HostArray(A.Boolean, rng.random(n) < 0.05)
    da, dm = a.to_device(ctx), m.to_device(ctx)
    out = A._lib.ArrayOut()
    vv, mv = da.view(), dm.view()
    ctx.check(ctx.lib.ah_filter(ctx.handle, C.byref(vv), C.byref(mv), C.byref(out)))
    assert out.values_bytes == n * 8 and out.length < n // 10  # worst-case capacity
    ctx.check(ctx.lib.ah_array_shrink_to_fit(ctx.handle, C.byref(out)))
    assert out.values_bytes == out.length * 8  # (the 125 KB bitmap is within the 1 MiB slack the call leaves alone)
    got = A.Array._from_out(ctx, out, A.Int64)
    assert_logical_eq(HostArray.from_device(got), oracle.filter(a, m), "shrunk")
