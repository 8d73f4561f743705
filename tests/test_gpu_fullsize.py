"""Full-coverage parity at the BASELINE.json sizes.

Every row of the device result is compared with the oracle, not a window of it: the inputs are produced by the
same counter-based generators on both sides (`ah_gen_*` in HBM, `orc_gen_*` on the host, row-number keyed, so any
chunk of the global column can be regenerated independently), the oracle runs chunk by chunk on the host cores
(its kernels are chunk-concatenable: filter is order preserving, the element-wise kernels are row independent
when chunks start on 64-row boundaries), and each chunk's output is compared with the matching slice of the device
output — raw bytes, validity bits and null counts.

  configs[1]  filter + take on 1e9 Int64 rows, 10 % nulls, 10 % selectivity, 1e8 u32 indices (+ 10 %-null indices)
  configs[2]  add_wrapping and lt on 1e9 Float64 rows with NullBuffers (NaN / inf / -0 / denormal rows patched in)
  configs[3]  Int64 -> Float64 -> LargeUtf8 on 2^29 rows (the > 2^31-byte i64 offsets included)

TEST INFRASTRUCTURE: uses the oracle.  Needs ~40 GB of host memory and a GPU with ~40 GB free."""
import concurrent.futures as cf
import ctypes as C
import os

import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import _lib as L
from arrow_rs_amd import compute as K
from arrow_rs_amd.array import _copy_dtoh

import bench as B
import orc

def _host_memory_ok():
    """The oracle side holds the 8 GB column, its filtered parts and per-chunk scratch on the host (~25 GB at the peak):
    on a box that cannot hold that, skipping beats an out-of-memory kill of the whole test run."""
    try:
        import psutil
        return psutil.virtual_memory().available >= 48 * 2**30
    except Exception:  # noqa: BLE001 - no psutil: assume the pool's standard box
        return True


pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not _host_memory_ok(), reason="full-size parity needs ~48 GB of free host memory")]

CH = 1 << 24  # rows per oracle chunk (a multiple of 64: bitmaps split on word boundaries)
NT = max(4, min(64, (os.cpu_count() or 8) - 2))


def _view(type_id, length, values=None, validity=None, null_count=-1):
    v = orc.View()
    v.type, v.length, v.null_count = type_id, length, null_count
    if values is not None:
        v.values = values.ctypes.data
    if validity is not None:
        v.validity = validity.ctypes.data
    return v


def _bits(oracle, rows, seed, p, row0):
    b = np.zeros(((rows + 63) // 64) * 8, dtype=np.uint8)
    oracle.lib.orc_gen_bernoulli_bits(b.ctypes.data, rows, seed, p, row0)
    return b


def _unpack(bits, n):
    return np.unpackbits(bits, bitorder="little", count=n).view(np.bool_)


def _out_bytes(ptr, nbytes):
    if not ptr or nbytes <= 0:
        return np.empty(0, dtype=np.uint8)
    return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=np.uint8).copy()


def _valid_of(out, rows):
    return _unpack(_out_bytes(out.validity, (rows + 7) // 8), rows) if out.validity else np.ones(rows, dtype=bool)


def _chunks(n):
    return [(c0, min(CH, n - c0)) for c0 in range(0, n, CH)]


def _dev_bytes(ctx, buf, byte_off, nbytes):
    return _copy_dtoh(ctx, buf.ptr + byte_off, nbytes)


def _gen_i64_chunk(oracle, c0, rows, seed, p_valid, lo=-2**63, hi=2**63 - 1):
    v = np.empty(rows, dtype=np.int64)
    oracle.lib.orc_gen_uniform_i64(v.ctypes.data, rows, seed, lo, hi, c0)
    vb = _bits(oracle, rows, seed + 1, p_valid, c0)
    np.multiply(v, _unpack(vb, rows), out=v)  # ah_zero_null_slots (a ufunc: runs without the GIL)
    return v, vb


def _gen_f64_chunk(oracle, c0, rows, seed, p_valid):
    v = np.empty(rows, dtype=np.float64)
    oracle.lib.orc_gen_uniform_f64(v.ctypes.data, rows, seed, -1e6, 1e6, c0)
    vb = _bits(oracle, rows, seed + 1, p_valid, c0)
    iv = v.view(np.int64)
    np.multiply(iv, _unpack(vb, rows), out=iv)  # null slots -> +0.0 bits, like ah_zero_null_slots
    return v, vb


# ------------------------------------------------------------------------------------------- configs[1]
def test_config1_filter_take_every_row(ctx, oracle):
    n = 1_000_000_000
    col = B.gen_i64_column(A, ctx, n, 42, 0.9, 0)
    pred = B.gen_predicate(A, ctx, n, 44, 0.1, 0)
    f = K.filter(col, pred)
    K_dev = f.length
    assert abs(K_dev / n - 0.1) < 1e-3 and f.validity is not None

    # host copy of the whole column (8 GB + bitmap): the take oracle needs random access to it
    host_vals = np.empty(n, dtype=np.int64)
    host_valid = np.zeros(n // 8, dtype=np.uint8)

    def filt(job):
        c0, rows = job
        v, vb = _gen_i64_chunk(oracle, c0, rows, 42, 0.9)
        host_vals[c0:c0 + rows] = v
        host_valid[c0 // 8:(c0 + rows) // 8] = vb[:rows // 8]
        mb = _bits(oracle, rows, 44, 0.1, c0)
        out = orc.Out()
        st = oracle.lib.orc_filter(C.byref(_view(L.AH_INT64, rows, v, vb)), C.byref(_view(L.AH_BOOL, rows, mb, None, 0)),
                                   C.byref(out))
        assert st == 0
        k = out.length
        vals = _out_bytes(out.values, k * 8).view(np.int64)
        valid = _valid_of(out, k)
        nulls = out.null_count
        oracle.lib.orc_release(C.byref(out))
        return k, vals, valid, nulls

    with cf.ThreadPoolExecutor(NT) as ex:
        parts = list(ex.map(filt, _chunks(n)))
    assert sum(p[0] for p in parts) == K_dev, "selected row count"
    assert sum(p[3] for p in parts) == f.null_count(), "filter null_count"
    dvals = _dev_bytes(ctx, f.values, 0, K_dev * 8).view(np.int64)
    dvalid = _unpack(_dev_bytes(ctx, f.validity, 0, (K_dev + 7) // 8), K_dev)
    off = 0
    for k, vals, valid, _ in parts:
        assert np.array_equal(dvals[off:off + k], vals), f"filter values differ in output rows [{off}, {off + k})"
        assert np.array_equal(dvalid[off:off + k], valid), f"filter validity differs in output rows [{off}, {off + k})"
        off += k
    del dvals, dvalid, parts, f

    # take: all 1e8 indices, no index nulls, then the 10 %-null-index variant (take.rs:432-457 first arm)
    m = n // 10
    ib = ctx.alloc(m * 4)
    ctx.check(ctx.lib.ah_gen_uniform_u32(ctx.handle, ib.ptr, m, 45, n, 0))
    idx = B.mk_array(A, ctx, A.UInt32, m, ib)
    hidx = np.empty(m, dtype=np.uint32)
    oracle.lib.orc_gen_uniform_u32(hidx.ctypes.data, m, 45, n, 0)
    ivb_dev = ctx.alloc(((m + 63) // 64) * 8)
    ctx.check(ctx.lib.ah_gen_bernoulli_bits(ctx.handle, ivb_dev.ptr, m, 47, 0.9, 0))
    hivb = _bits(oracle, m, 47, 0.9, 0)
    R = A.array._RawMem
    idx_nulls = A.Array(ctx, A.UInt32, m, idx.values, 0, R(ivb_dev.ptr, ivb_dev.nbytes, ivb_dev), 0,
                        m - int(_unpack(hivb, m).sum()))
    full = _view(L.AH_INT64, n, host_vals, host_valid)
    for name, darr, with_nulls in (("take", idx, False), ("take with 10 % null indices", idx_nulls, True)):
        t = K.take(col, darr)
        tv = _dev_bytes(ctx, t.values, 0, m * 8).view(np.int64)
        tb = _unpack(_dev_bytes(ctx, t.validity, 0, (m + 7) // 8), m)

        def take(job):
            c0, rows = job
            iv = _view(L.AH_UINT32, rows, hidx[c0:c0 + rows], hivb[c0 // 8:] if with_nulls else None, -1 if with_nulls else 0)
            out = orc.Out()
            st = oracle.lib.orc_take(C.byref(full), C.byref(iv), 0, C.byref(out))
            assert st == 0
            ok = np.array_equal(_out_bytes(out.values, rows * 8).view(np.int64), tv[c0:c0 + rows])
            ok = ok and np.array_equal(_valid_of(out, rows), tb[c0:c0 + rows])
            nulls = out.null_count
            oracle.lib.orc_release(C.byref(out))
            return ok, nulls

        with cf.ThreadPoolExecutor(NT) as ex:
            res = list(ex.map(take, [(c0, min(1 << 22, m - c0)) for c0 in range(0, m, 1 << 22)]))
        assert all(r[0] for r in res), f"{name}: device rows differ from the oracle"
        assert sum(r[1] for r in res) == t.null_count(), f"{name}: null_count"


# ------------------------------------------------------------------------------ BatchCoalescer at the bench's size
def test_coalescer_filtered_pushes_every_row(ctx, oracle):
    """SURVEY 8f-1 at bench.py's `coalesce` size: 1e9 rows of {Int64, Float64} pushed as 2^24-row batches with a 10 %
    filter (coalesce.rs:229); every output batch — size, values, validity, null count — is compared with the oracle's
    filter of the same rows cut into `target`-row batches.  Covers the one-launch-for-all-columns scatter and its
    output windows (pushes that straddle output batches) at large destination offsets."""
    n, br = 1_000_000_000, 1 << 24
    a = B.gen_i64_column(A, ctx, n, 42, 0.9, 0)
    b = B.gen_f64_column(A, ctx, n, 52, 0.9, 0)
    pred = B.gen_predicate(A, ctx, n, 44, 0.1, 0)
    target = int(br * 0.1 * 4)
    co = K.BatchCoalescer.new(["a", "b"], [A.Int64, A.Float64], target, ctx)
    got = []
    for i in range(0, n, br):
        rows = min(br, n - i)
        co.push_batch_with_filter(A.RecordBatch(["a", "b"], [a.slice(i, rows), b.slice(i, rows)]), pred.slice(i, rows))
        while co.has_completed_batch():
            got.append(co.next_completed_batch())
    co.finish_buffered_batch()
    while co.has_completed_batch():
        got.append(co.next_completed_batch())
    # round 4: the same stream through the PIPELINED grouped push (begin of group g + 1 before end of group g: the counts
    # of the next group make their round trip while this group's scatters run) must produce the same output batches
    co2 = K.BatchCoalescer.new(["a", "b"], [A.Int64, A.Float64], target, ctx)
    pairs = [(A.RecordBatch(["a", "b"], [a.slice(i, min(br, n - i)), b.slice(i, min(br, n - i))]), pred.slice(i, min(br, n - i)))
             for i in range(0, n, br)]
    got2, pending = [], None
    for g0 in range(0, len(pairs), 8):
        nxt = co2.push_batches_with_filters_begin(pairs[g0:g0 + 8])
        if pending is not None:
            pending.end()
        pending = nxt
        while co2.has_completed_batch():
            got2.append(co2.next_completed_batch())
    pending.end()
    co2.finish_buffered_batch()
    while co2.has_completed_batch():
        got2.append(co2.next_completed_batch())
    assert [g.num_rows() for g in got2] == [g.num_rows() for g in got]

    def filt(job):
        c0, rows = job
        mb = _bits(oracle, rows, 44, 0.1, c0)
        res = []
        for gen, seed, tid, dt in ((_gen_i64_chunk, 42, L.AH_INT64, np.int64), (_gen_f64_chunk, 52, L.AH_FLOAT64, np.float64)):
            v, vb = gen(oracle, c0, rows, seed, 0.9)
            out = orc.Out()
            assert oracle.lib.orc_filter(C.byref(_view(tid, rows, v, vb)), C.byref(_view(L.AH_BOOL, rows, mb, None, 0)), C.byref(out)) == 0
            k = out.length
            res.append((_out_bytes(out.values, k * 8).view(np.uint64), _valid_of(out, k)))
            oracle.lib.orc_release(C.byref(out))
        return res

    with cf.ThreadPoolExecutor(NT) as ex:
        parts = list(ex.map(filt, _chunks(n)))
    total = sum(len(p[0][0]) for p in parts)
    assert sum(g.num_rows() for g in got) == total
    assert all(g.num_rows() == target for g in got[:-1]) and 0 < got[-1].num_rows() <= target
    for c in range(2):
        ev = np.concatenate([p[c][0] for p in parts])
        eb = np.concatenate([p[c][1] for p in parts])
        for which, batches in (("single pushes", got), ("pipelined grouped pushes", got2)):
            off = 0
            for bi, g in enumerate(batches):
                k, col = g.num_rows(), g.columns[c]
                assert col.length == k
                dv = _dev_bytes(ctx, col.values, 0, k * 8).view(np.uint64)
                assert np.array_equal(dv, ev[off:off + k]), f"{which}: column {c}, output batch {bi}: values"
                exp_valid = eb[off:off + k]
                if col.validity is None:
                    assert exp_valid.all(), f"{which}: column {c}, output batch {bi}: null buffer missing"
                else:
                    assert np.array_equal(_unpack(_dev_bytes(ctx, col.validity, 0, (k + 7) // 8), k), exp_valid), \
                        f"{which}: column {c}, output batch {bi}: validity"
                assert col.null_count() == int(k - exp_valid.sum()), f"{which}: column {c}, output batch {bi}: null_count"
                off += k


# ------------------------------------------------------------------------------------------- configs[2]
_SPECIALS = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 2.2250738585072014e-308, 1.7976931348623157e308,
                      -1.7976931348623157e308, 1.0, -1.0], dtype=np.float64)


def _patch_specials(a, b):
    """rows [0, 4096) of chunk 0: every pair of special operands (NaN payloads of both signs included)"""
    k = len(_SPECIALS)
    sa = np.tile(np.repeat(_SPECIALS, k), 4096 // (k * k) + 1)[:4096].copy()
    sb = np.tile(np.tile(_SPECIALS, k), 4096 // (k * k) + 1)[:4096].copy()
    sa.view(np.uint64)[5::97] = 0xFFF8000000000123  # negative quiet NaN with a payload
    sb.view(np.uint64)[7::89] = 0x7FF0000000000456  # positive signalling NaN with a payload
    a[:4096], b[:4096] = sa, sb
    return sa, sb


def test_config2_add_wrapping_and_lt_every_row(ctx, oracle):
    n = 1_000_000_000
    a = B.gen_f64_column(A, ctx, n, 52, 0.9, 0)
    b = B.gen_f64_column(A, ctx, n, 62, 0.9, 0)
    sa, sb = _patch_specials(np.zeros(4096), np.zeros(4096))
    ctx.check(ctx.lib.ah_memcpy_htod(ctx.handle, a.values.ptr, sa.ctypes.data, sa.nbytes))
    ctx.check(ctx.lib.ah_memcpy_htod(ctx.handle, b.values.ptr, sb.ctypes.data, sb.nbytes))
    s = K.add_wrapping(a, b)
    l = K.lt(a, b)
    assert s.length == n and l.length == n

    def work(job):
        c0, rows, sv, sbits, lv, lbits = job
        va, vab = _gen_f64_chunk(oracle, c0, rows, 52, 0.9)
        vb, vbb = _gen_f64_chunk(oracle, c0, rows, 62, 0.9)
        if c0 == 0:
            _patch_specials(va, vb)  # patched AFTER the null-slot zeroing on both sides
        av, bv = _view(L.AH_FLOAT64, rows, va, vab), _view(L.AH_FLOAT64, rows, vb, vbb)
        out = orc.Out()
        assert oracle.lib.orc_arith(1, C.byref(av), 0, C.byref(bv), 0, C.byref(out)) == 0
        ev = _out_bytes(out.values, rows * 8).view(np.uint64)
        eb = _out_bytes(out.validity, rows // 8)
        nulls = out.null_count
        oracle.lib.orc_release(C.byref(out))
        both_nan = np.isnan(va) & np.isnan(vb)  # operand order of a commutative SSE op is the compiler's choice
        ok_add = np.array_equal(ev[~both_nan], sv.view(np.uint64)[~both_nan]) and np.all(np.isnan(sv[both_nan]))
        ok_add = ok_add and np.array_equal(eb, sbits)
        out = orc.Out()
        assert oracle.lib.orc_compare(2, C.byref(av), 0, C.byref(bv), 0, C.byref(out)) == 0
        ok_lt = np.array_equal(_out_bytes(out.values, rows // 8), lv) and np.array_equal(_out_bytes(out.validity, rows // 8), lbits)
        lnulls = out.null_count
        oracle.lib.orc_release(C.byref(out))
        return ok_add, ok_lt, nulls, lnulls

    def jobs():
        for c0, rows in _chunks(n):
            assert rows % 64 == 0
            yield (c0, rows, _dev_bytes(ctx, s.values, c0 * 8, rows * 8).view(np.float64),
                   _dev_bytes(ctx, s.validity, c0 // 8, rows // 8), _dev_bytes(ctx, l.values, c0 // 8, rows // 8),
                   _dev_bytes(ctx, l.validity, c0 // 8, rows // 8))

    res = []
    with cf.ThreadPoolExecutor(NT) as ex:
        pending = []
        for job in jobs():  # device -> host copies stay on this thread (one context = one thread)
            pending.append(ex.submit(work, job))
            if len(pending) >= NT:
                res.append(pending.pop(0).result())
        res += [p.result() for p in pending]
    assert all(r[0] for r in res), "add_wrapping differs from the oracle"
    assert all(r[1] for r in res), "lt differs from the oracle"
    assert sum(r[2] for r in res) == s.null_count() and sum(r[3] for r in res) == l.null_count()


# ------------------------------------------------------------------------- the lazy predicate, 1e9 rows
def test_lazy_predicate_filter_every_row(ctx, oracle):
    """`filter(a, and_kleene(lt(a, 0), gt_eq(b, 0.0)))` with the comparisons evaluated inside the filter's count pass
    (ah_filter_predicate_build_expr) at 1e9 rows: every output row against the oracle's MATERIALISED chain
    (orc_compare x 2 -> orc_boolean_binary(and_kleene) -> orc_filter), null operands on both sides, NaN / inf / -0
    rows patched into b."""
    n = 1_000_000_000
    a = B.gen_i64_column(A, ctx, n, 42, 0.9, 0)
    b = B.gen_f64_column(A, ctx, n, 52, 0.9, 0)
    sa, sb = _patch_specials(np.zeros(4096), np.zeros(4096))
    ctx.check(ctx.lib.ah_memcpy_htod(ctx.handle, b.values.ptr, sb.ctypes.data, sb.nbytes))
    s0, s1 = A.Scalar.new(0, A.Int64, ctx), A.Scalar.new(0.0, A.Float64, ctx)
    pred = K.FilterBuilder.from_terms([("lt", a, s0), ("gt_eq", b, s1)], ["and_kleene"]).build()
    f2 = pred.filter(a)  # through the predicate object
    f = K.filter_expr(a, [("lt", a, s0), ("gt_eq", b, s1)], ["and_kleene"])  # the one-call form
    K_dev = f.length
    assert K_dev == pred.count() == f2.length and 0.15 < K_dev / n < 0.25
    assert f.null_count() == f2.null_count()
    assert (f.validity is None) == (f2.validity is None)  # (`a < 0` is null where a is: every selected a is valid)
    for off in range(0, K_dev, 1 << 26):  # the two forms against each other, byte for byte (and below against the oracle)
        m = min(1 << 26, K_dev - off)
        assert np.array_equal(_dev_bytes(ctx, f.values, off * 8, m * 8), _dev_bytes(ctx, f2.values, off * 8, m * 8))
    del f2, pred
    z0, z1 = np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.float64)

    def work(job):
        c0, rows = job
        va, vab = _gen_i64_chunk(oracle, c0, rows, 42, 0.9)
        vb, vbb = _gen_f64_chunk(oracle, c0, rows, 52, 0.9)
        if c0 == 0:
            _patch_specials(np.zeros(4096), vb)
        av, bv = _view(L.AH_INT64, rows, va, vab), _view(L.AH_FLOAT64, rows, vb, vbb)
        m1, m2, m = orc.Out(), orc.Out(), orc.Out()
        assert oracle.lib.orc_compare(2, C.byref(av), 0, C.byref(_view(L.AH_INT64, 1, z0, None, 0)), 1, C.byref(m1)) == 0
        assert oracle.lib.orc_compare(5, C.byref(bv), 0, C.byref(_view(L.AH_FLOAT64, 1, z1, None, 0)), 1, C.byref(m2)) == 0

        def as_view(o):
            v = orc.View()
            v.type, v.length, v.null_count, v.values, v.validity = L.AH_BOOL, o.length, o.null_count, o.values, o.validity
            return v
        assert oracle.lib.orc_boolean_binary(3, C.byref(as_view(m1)), C.byref(as_view(m2)), C.byref(m)) == 0
        out = orc.Out()
        assert oracle.lib.orc_filter(C.byref(av), C.byref(as_view(m)), C.byref(out)) == 0
        k = out.length
        vals = _out_bytes(out.values, k * 8).view(np.int64)
        valid = _valid_of(out, k)
        nulls = out.null_count
        for o in (m1, m2, m, out):
            oracle.lib.orc_release(C.byref(o))
        return k, vals, valid, nulls

    with cf.ThreadPoolExecutor(NT) as ex:
        parts = list(ex.map(work, _chunks(n)))
    assert sum(p_[0] for p_ in parts) == K_dev, "selected row count"
    assert sum(p_[3] for p_ in parts) == f.null_count(), "null_count"
    dvals = _dev_bytes(ctx, f.values, 0, K_dev * 8).view(np.int64)
    dvalid = _unpack(_dev_bytes(ctx, f.validity, 0, (K_dev + 7) // 8), K_dev) if f.validity is not None else np.ones(K_dev, dtype=bool)
    off = 0
    for k, vals, valid, _ in parts:
        assert np.array_equal(dvalid[off:off + k], valid), f"validity differs in output rows [{off}, {off + k})"
        assert np.array_equal(dvals[off:off + k][valid], vals[valid]), f"values differ in output rows [{off}, {off + k})"
        off += k


# ------------------------------------------------------------------------------------------- configs[3]
def test_config3_cast_chain_every_row(ctx, oracle):
    n = 1 << 29
    # bench.py's config-4 source: [-1e6, 1e6] with 1 % full-range rows (>= 2^53 rounding, exponent forms)
    src = B.gen_cast_source(A, K, ctx, n, 0.9, 0)
    f64 = K.cast(src, A.Float64)
    txt = K.cast(f64, A.LargeUtf8)
    assert txt.length == n and txt.null_count() == src.null_count() == f64.null_count()
    total = int(_dev_bytes(ctx, txt.offsets, n * 8, 8).view(np.int64)[0])
    assert total > 2**31, "the chain must cross the i32 offset limit (that is why it is LargeUtf8)"

    def work(job):
        c0, rows, dv, dbits, doffs, dtext = job
        v, vb = _gen_i64_chunk(oracle, c0, rows, 42, 0.9, -10**6, 10**6)
        full = np.empty(rows, dtype=np.int64)
        oracle.lib.orc_gen_uniform_i64(full.ctypes.data, rows, 78, -2**63, 2**63 - 1, c0)
        pick = _unpack(_bits(oracle, rows, 79, 0.01, c0), rows)
        np.copyto(v, full, where=pick & _unpack(vb, rows))  # null slots stay 0
        o1 = orc.Out()
        assert oracle.lib.orc_cast(C.byref(_view(L.AH_INT64, rows, v, vb)), L.AH_FLOAT64, 1, C.byref(o1)) == 0
        ok1 = np.array_equal(_out_bytes(o1.values, rows * 8).view(np.uint64), dv.view(np.uint64))
        ok1 = ok1 and np.array_equal(_out_bytes(o1.validity, rows // 8), dbits)
        fv = orc.View()
        fv.type, fv.length, fv.null_count, fv.values, fv.validity = L.AH_FLOAT64, rows, o1.null_count, o1.values, o1.validity
        o2 = orc.Out()
        assert oracle.lib.orc_cast(C.byref(fv), L.AH_LARGE_UTF8, 1, C.byref(o2)) == 0
        eoffs = _out_bytes(o2.offsets, (rows + 1) * 8).view(np.int64)
        ok2 = np.array_equal(eoffs, doffs - doffs[0])
        ok2 = ok2 and np.array_equal(_out_bytes(o2.values, int(eoffs[-1])), dtext)
        oracle.lib.orc_release(C.byref(o1))
        oracle.lib.orc_release(C.byref(o2))
        return ok1, ok2

    def jobs():
        for c0, rows in _chunks(n):
            doffs = _dev_bytes(ctx, txt.offsets, c0 * 8, (rows + 1) * 8).view(np.int64)
            yield (c0, rows, _dev_bytes(ctx, f64.values, c0 * 8, rows * 8).view(np.float64),
                   _dev_bytes(ctx, f64.validity, c0 // 8, rows // 8), doffs,
                   _dev_bytes(ctx, txt.values, int(doffs[0]), int(doffs[-1] - doffs[0])))

    res = []
    with cf.ThreadPoolExecutor(NT) as ex:
        pending = []
        for job in jobs():
            pending.append(ex.submit(work, job))
            if len(pending) >= NT:
                res.append(pending.pop(0).result())
        res += [p.result() for p in pending]
    assert all(r[0] for r in res), "Int64 -> Float64 differs from the oracle"
    assert all(r[1] for r in res), "Float64 -> LargeUtf8 differs from the oracle"
    tb = _dev_bytes(ctx, txt.validity, 0, n // 8)
    assert np.array_equal(tb, _dev_bytes(ctx, f64.validity, 0, n // 8)), "string validity = input validity"
    # round 5: the same chain as ONE call (ah_cast_chain: text straight from the Int64 column, no Float64 array) — every
    # offset, every byte and the validity identical to the step-by-step result checked against the oracle above
    del f64
    chain = K.cast_chain(src, [A.Float64, A.LargeUtf8])
    assert chain.length == n and chain.null_count() == txt.null_count()
    # compared ON THE DEVICE (18 GB of offsets and text would take minutes over PCIe): not_distinct is true where both rows are
    # null or both hold the same bytes; the offsets are then pinned by the last one (the total) and the row count
    same = K.not_distinct(chain, txt)
    cnt = C.c_int64()
    ctx.check(ctx.lib.ah_count_set_bits(ctx.handle, same.values.ptr, 0, n, C.byref(cnt)))
    assert cnt.value == n, f"{n - cnt.value} rows of the chain differ from the two-call result"
    assert int(_dev_bytes(ctx, chain.offsets, n * 8, 8).view(np.int64)[0]) == total
    assert np.array_equal(_dev_bytes(ctx, chain.validity, 0, n // 8), tb), "chain validity"


def test_string_filter_take_every_row_commutes_with_cast(ctx):
    """String filter / take at the bench's scale (2^28 rows = 65 536 filter tiles, 2^26 indices = 65 536 take rounds: 64
    chained scan workgroups each, scan_chain.hpp).  filter / take of the printed column must equal the printed filter / take
    of the numbers — the right-hand sides never run a string scan — compared row by row ON THE DEVICE (not_distinct) plus
    the byte totals; the numeric filter / take and the cast are the paths the other full-size tests pin to the oracle."""
    n = 1 << 28
    src = B.gen_cast_source(A, K, ctx, n, 0.9, 0)
    txt = K.cast(src, A.LargeUtf8)

    def same_rows(a, b, what):
        assert a.length == b.length and a.null_count() == b.null_count(), what
        k = a.length
        ta = int(_dev_bytes(ctx, a.offsets, k * 8, 8).view(np.int64)[0]) - int(_dev_bytes(ctx, a.offsets, 0, 8).view(np.int64)[0])
        tb = int(_dev_bytes(ctx, b.offsets, k * 8, 8).view(np.int64)[0]) - int(_dev_bytes(ctx, b.offsets, 0, 8).view(np.int64)[0])
        assert ta == tb, f"{what}: {ta} bytes against {tb}"
        same = K.not_distinct(a, b)
        cnt = C.c_int64()
        ctx.check(ctx.lib.ah_count_set_bits(ctx.handle, same.values.ptr, 0, k, C.byref(cnt)))
        assert cnt.value == k, f"{what}: {k - cnt.value} rows differ"

    mask = B.gen_predicate(A, ctx, n, 91, 0.1, 0)
    same_rows(K.filter(txt, mask), K.cast(K.filter(src, mask), A.LargeUtf8), "filter")
    k = 1 << 26
    ib = ctx.alloc(k * 4)
    ctx.check(ctx.lib.ah_gen_uniform_u32(ctx.handle, ib.ptr, k, 92, n, 0))
    idx = B.mk_array(A, ctx, A.UInt32, k, ib)
    same_rows(K.take(txt, idx), K.cast(K.take(src, idx), A.LargeUtf8), "take")


# ---------------------------------------------------------------------- 32-byte natives (i256) through filter / take
@pytest.mark.parametrize("n", [1, 63, 1000, 1025, 40_000, 1_000_003])
def test_filter_take_32_byte_natives(ctx, oracle, n):
    """filter_native / take_native are width generic and include i256 (filter.rs:731-770, take.rs:432-457):
    the W = 32 instantiations against the oracle — validity, slices, index nulls and every index width class."""
    rng = np.random.default_rng(n)
    dt = A.Decimal256(50, 3)
    vals = np.zeros(n, dtype=dt.np_dtype)
    for w in ("w0", "w1", "w2", "w3"):
        vals[w] = rng.integers(0, 2**63, n, dtype=np.int64).astype(vals[w].dtype)
    for valid in (None, rng.random(n) < 0.85):
        h = orc.HostArray(dt, vals, valid)
        d = h.to_device(ctx)
        for sel in (0.0, 0.07, 0.5, 0.93, 1.0):
            m = orc.HostArray(A.Boolean, rng.random(n) < sel, (rng.random(n) < 0.9) if sel == 0.5 else None)
            got = K.filter(d, m.to_device(ctx))
            orc.assert_logical_eq(orc.HostArray.from_device(got), oracle.filter(h, m), f"W=32 filter n={n} sel={sel}")
        for it, npd in ((A.UInt32, np.uint32), (A.Int64, np.int64), (A.UInt8, np.uint8)):
            hi = min(n, np.iinfo(npd).max)
            k = max(1, n // 3)
            idx = orc.HostArray(it, rng.integers(0, hi, k).astype(npd), (rng.random(k) < 0.8) if it is A.Int64 else None)
            got = K.take(d, idx.to_device(ctx))
            orc.assert_logical_eq(orc.HostArray.from_device(got), oracle.take(h, idx), f"W=32 take n={n} {it}")
        if n > 70:  # a real zero-copy slice at an odd row: 32-byte elements are still 16-byte aligned
            sl, hs = d.slice(3, n - 7), h.slice(3, n - 7)
            m = orc.HostArray(A.Boolean, rng.random(n - 7) < 0.3)
            orc.assert_logical_eq(orc.HostArray.from_device(K.filter(sl, m.to_device(ctx))), oracle.filter(hs, m), "W=32 sliced")
