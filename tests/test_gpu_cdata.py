"""Arrow C Data Interface boundary: pyarrow speaks the same structs as
`FFI_ArrowArray` / `FFI_ArrowSchema` (arrow-data/src/ffi.rs:37-66, arrow-schema/src/ffi.rs:76-98),
so it plays the foreign producer/consumer the reference tests build by hand
(arrow-array/src/ffi.rs:620-1900 round-trip tests).  pyarrow is Arrow C++, not the reference: it is
used as the *transport peer*, and as a second opinion only where its semantics coincide with the
reference's (filter, take, wrapping integer add — SURVEY.md §8c)."""
import ctypes as C
import datetime
import decimal

import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K, ffi

pa = pytest.importorskip("pyarrow")
import pyarrow.compute as pc  # noqa: E402

pytestmark = pytest.mark.gpu


def _rng_valid(n, seed, p=0.8):
    return np.random.default_rng(seed).random(n) < p


def _pa_cases():
    n = 1000
    rng = np.random.default_rng(7)
    out = []
    for t in (pa.int8(), pa.int16(), pa.int32(), pa.int64(), pa.uint8(), pa.uint16(), pa.uint32(), pa.uint64()):
        info = np.iinfo(t.to_pandas_dtype())
        vals = rng.integers(info.min, info.max, n, dtype=t.to_pandas_dtype(), endpoint=True)
        out.append(pa.array(vals, type=t, mask=~_rng_valid(n, 1)))
        out.append(pa.array(vals, type=t))
    for t, npdt in ((pa.float16(), np.float16), (pa.float32(), np.float32), (pa.float64(), np.float64)):
        out.append(pa.array(rng.standard_normal(n).astype(npdt), type=t, mask=~_rng_valid(n, 2)))
    out.append(pa.array(rng.random(n) < 0.5, type=pa.bool_(), mask=~_rng_valid(n, 3)))
    out.append(pa.array(rng.random(n) < 0.5, type=pa.bool_()))
    i64 = rng.integers(-2**40, 2**40, n)
    for t in (pa.timestamp("us", tz="UTC"), pa.timestamp("ns"), pa.timestamp("s", tz="America/New_York"),
              pa.duration("ms"), pa.date64(), pa.time64("ns")):
        src = i64 // 86_400_000 * 86_400_000 if t == pa.date64() else (i64 % 86_400_000_000_000 if t == pa.time64("ns") else i64)
        out.append(pa.array(src, type=pa.int64(), mask=~_rng_valid(n, 4)).view(t))
    i32 = rng.integers(0, 86399, n).astype(np.int32)
    for t in (pa.date32(), pa.time32("s")):
        out.append(pa.array(i32, type=pa.int32(), mask=~_rng_valid(n, 5)).view(t))
    decs = [None if i % 7 == 0 else decimal.Decimal(int(v)).scaleb(-3) for i, v in enumerate(i64)]
    out.append(pa.array(decs, type=pa.decimal128(20, 3)))
    out.append(pa.array(decs, type=pa.decimal256(50, 3)))
    strs = [None if i % 5 == 0 else ("s" * (i % 23)) + str(i) for i in range(n)]
    out.append(pa.array(strs, type=pa.string()))
    out.append(pa.array(strs, type=pa.large_string()))
    out.append(pa.array([None if s is None else s.encode() + b"\xff\x00" for s in strs], type=pa.binary()))
    out.append(pa.array([None if s is None else s.encode() for s in strs], type=pa.large_binary()))
    return out


@pytest.mark.parametrize("arr", _pa_cases(), ids=lambda a: f"{a.type}-{a.null_count}")
def test_round_trip_every_layout(ctx, arr):
    for sl in (arr, arr.slice(3, 700), arr.slice(13, 64), arr.slice(999, 1), arr.slice(5, 0)):
        dev = A.Array.from_pyarrow(sl, ctx)
        assert dev.length == len(sl) and dev.null_count() == sl.null_count
        back = dev.to_pyarrow()
        assert back.type == sl.type
        assert back.null_count == sl.null_count
        assert back.equals(sl), (arr.type, sl.offset, len(sl))
        back.validate(full=True)


def test_export_realigns_bit_offsets(ctx):
    """`FFI_ArrowArray::new` re-aligns nulls to the exported offset (arrow-data/src/ffi.rs:104-121):
    device slices carry bit offsets, the exported array always starts at offset 0."""
    n = 5000
    vals = np.arange(n, dtype=np.int64)
    valid = _rng_valid(n, 11)
    bools = _rng_valid(n, 12, 0.5)
    d = A.Array.from_numpy(vals, valid, ctx=ctx, bit_offset=5)
    b = A.Array.from_numpy(bools, valid, ctx=ctx, bit_offset=3)
    for off, ln in ((0, n), (1, 100), (7, 999), (64, 64), (77, 4000), (4999, 1)):
        got = d.slice(off, ln).to_pyarrow()
        assert got.offset == 0
        assert got.equals(pa.array(vals, mask=~valid).slice(off, ln))
        got = b.slice(off, ln).to_pyarrow()
        assert got.equals(pa.array(bools, mask=~valid).slice(off, ln))


def test_unknown_null_count_is_counted_on_device(ctx):
    arr = pa.array([1, None, 3, None, None, 6] * 50, type=pa.int32()).slice(2, 200)
    a, s = ffi.FFI_ArrowArray(), ffi.FFI_ArrowSchema()
    arr._export_to_c(C.addressof(a), C.addressof(s))
    a.null_count = -1  # "not yet computed" (null_count_opt, arrow-data/src/ffi.rs:327)
    dev = ffi.from_ffi(a, s, ctx)
    assert dev.null_count() == arr.null_count
    assert dev.to_pyarrow().equals(arr)
    for st in (a, s):
        C.CFUNCTYPE(None, C.c_void_p)(st.release)(C.addressof(st))


def test_import_errors(ctx):
    arr = pa.array([1, 2, 3], type=pa.int64())
    a, s = ffi.FFI_ArrowArray(), ffi.FFI_ArrowSchema()
    arr._export_to_c(C.addressof(a), C.addressof(s))
    a.n_buffers = 3
    with pytest.raises(A.CDataInterfaceError) as e:
        ffi.from_ffi(a, s, ctx)
    assert str(e.value) == ('C Data interface error: The datatype "Int64" expects 2 buffers, but requested 3. '
                            'Please verify that the C data interface is correctly implemented.')
    a.n_buffers = 2
    saved = a.buffers[1]
    a.buffers[1] = None
    with pytest.raises(A.CDataInterfaceError, match="The external buffer at position 1 is null."):
        ffi.from_ffi(a, s, ctx)
    a.buffers[1] = saved
    assert ffi.from_ffi(a, s, ctx).to_pyarrow().equals(arr)
    for st in (a, s):
        C.CFUNCTYPE(None, C.c_void_p)(st.release)(C.addressof(st))
    with pytest.raises(A.CDataInterfaceError):  # released structs (release == NULL)
        ffi.from_ffi(a, s, ctx)
    for bad in (pa.array([[1, 2], [3]]), pa.array(["a", "b"]).dictionary_encode(),
                pa.array(["a"], type=pa.string_view()), pa.nulls(3)):
        with pytest.raises(A.NotYetImplemented):
            A.Array.from_pyarrow(bad, ctx)


def test_kernels_between_pyarrow_arrays(ctx, oracle):
    """pyarrow array -> C Data -> HBM -> kernel -> C Data -> pyarrow; second opinion from Arrow C++
    where its semantics coincide with the reference."""
    n = 200_000
    rng = np.random.default_rng(3)
    vals = pa.array(rng.integers(-2**62, 2**62, n), mask=~_rng_valid(n, 21, 0.9))
    mask = pa.array(rng.random(n) < 0.1, mask=~_rng_valid(n, 22, 0.95))
    idx = pa.array(rng.integers(0, n, n // 3, dtype=np.uint32), mask=~_rng_valid(n // 3, 23, 0.9))
    dv, dm, di = (A.Array.from_pyarrow(x, ctx) for x in (vals, mask, idx))
    assert K.filter(dv, dm).to_pyarrow().equals(pc.filter(vals, mask, null_selection_behavior="drop"))
    assert K.take(dv, di).to_pyarrow().equals(pc.take(vals, idx))
    small = pa.array(rng.integers(-2**40, 2**40, n), mask=~_rng_valid(n, 24, 0.9))
    ds = A.Array.from_pyarrow(small, ctx)
    assert K.add_wrapping(dv, ds).to_pyarrow().equals(pc.add(vals, small))
    lt = K.lt(dv, ds).to_pyarrow()
    assert lt.equals(pc.less(vals, small))
    strs = pa.array([None if i % 9 == 0 else f"row-{i}" * (i % 4) for i in range(n)])
    dstr = A.Array.from_pyarrow(strs, ctx)
    assert K.filter(dstr, dm).to_pyarrow().equals(pc.filter(strs, mask, null_selection_behavior="drop"))
    assert K.take(dstr, di).to_pyarrow().equals(pc.take(strs, idx))
    ts = vals.view(pa.timestamp("us", tz="UTC"))  # logical type survives the device round trip
    got = K.filter(A.Array.from_pyarrow(ts, ctx), dm).to_pyarrow()
    assert got.type == ts.type and got.equals(pc.filter(ts, mask, null_selection_behavior="drop"))
    dec = pa.array([None if i % 11 == 0 else decimal.Decimal(i).scaleb(-2) for i in range(5000)],
                   type=pa.decimal128(12, 2))
    m2 = pa.array(rng.random(5000) < 0.3)
    got = K.filter(A.Array.from_pyarrow(dec, ctx), A.Array.from_pyarrow(m2, ctx)).to_pyarrow()
    assert got.type == dec.type and got.equals(pc.filter(dec, m2))


def test_record_batch_bridge(ctx):
    n = 10_000
    rng = np.random.default_rng(5)
    rb = pa.record_batch({
        "k": pa.array(rng.integers(0, 1000, n), mask=~_rng_valid(n, 31)),
        "v": pa.array(rng.standard_normal(n)),
        "s": pa.array([f"{i:x}" for i in range(n)]),
        "t": pa.array([datetime.date(2020, 1, 1) + datetime.timedelta(days=int(i % 400)) for i in range(n)]),
    })
    mask = pa.array(rng.random(n) < 0.25)
    drb = ffi.from_pyarrow(rb, ctx)
    got = ffi.to_pyarrow(K.filter_record_batch(drb, A.Array.from_pyarrow(mask, ctx)))
    assert got.equals(rb.filter(mask))
    assert got.schema.equals(rb.schema)


def test_to_ffi_struct_fields_and_release(ctx):
    d = A.Array.from_numpy(np.arange(10, dtype=np.int32), np.arange(10) % 3 != 0, ctx=ctx)
    ex = ffi.to_ffi(d)
    assert (ex.array.length, ex.array.null_count, ex.array.offset, ex.array.n_buffers, ex.array.n_children) == (10, 4, 0, 2, 0)
    assert ex.schema.format == b"i" and ex.schema.flags == 2 and ex.schema.n_children == 0
    assert ex.array.buffers[0] and ex.array.buffers[1] and ex.array.release and ex.schema.release
    host = np.ctypeslib.as_array(C.cast(ex.array.buffers[1], C.POINTER(C.c_int32)), (10,))
    assert host.tolist() == list(range(10))
    ex.release()
    assert not ex.array.release and not ex.schema.release  # arrow-data/src/ffi.rs:96
    ex.release()  # idempotent
    nn = ffi.to_ffi(A.Array.from_numpy(np.arange(4, dtype=np.float64), ctx=ctx))
    assert nn.array.null_count == 0 and not nn.array.buffers[0]  # no null buffer -> NULL (ffi.rs:170-178)
    s = ffi.to_ffi(A.Array.from_strings(["a", "", "ccc"], [True, False, True], ctx=ctx))
    assert s.array.n_buffers == 3 and s.schema.format == b"u"


# ------------------------------------------------------------------ C Device Data Interface
def test_device_interface_zero_copy_round_trip(ctx, oracle):
    """ArrowDeviceArray export / import: same HBM pointers on both sides, device_type ROCM, no sync event; bit
    offsets are re-aligned for the exported struct (offset 0) and honoured on import."""
    from orc import HostArray
    rng = np.random.default_rng(5)
    n = 10_000
    h = HostArray(A.Int64, rng.integers(-2**60, 2**60, n), rng.random(n) < 0.8)
    d = h.to_device(ctx)                      # validity at bit offset 0: exported by pointer
    ex = ffi.to_device_ffi(d)
    da = ex.array
    assert (da.device_type, da.device_id, da.sync_event) == (ffi.L.ARROW_DEVICE_ROCM, ctx.device, None)
    assert (da.array.length, da.array.null_count, da.array.offset, da.array.n_buffers) == (n, h.null_count, 0, 2)
    assert da.array.buffers[1] == d.values.ptr and da.array.buffers[0] == d.validity.ptr
    assert ex.schema.format == b"l"
    back = ffi.from_device_ffi(da, ex.schema, ctx, keepalive=ex)
    assert back.values.ptr == d.values.ptr and back.null_count() == h.null_count
    mask = HostArray(A.Boolean, rng.random(n) < 0.3)
    check = K.filter(back, mask.to_device(ctx))
    import orc
    orc.assert_logical_eq(HostArray.from_device(check), oracle.filter(h, mask), "filter on imported view")
    # consumer-side offset: a producer that slices by `offset` (values AND validity shift together)
    da.array.offset = 77
    da.array.length = 1000
    da.array.null_count = -1
    sl = ffi.from_device_ffi(da, ex.schema, ctx, keepalive=ex)
    orc.assert_logical_eq(HostArray.from_device(sl), h.slice(77, 1000), "offset import")
    assert sl.null_count() == h.slice(77, 1000).null_count
    da.array.offset, da.array.length, da.array.null_count = 0, n, h.null_count
    # sliced / bit-offset source: values by pointer, validity re-aligned into a fresh buffer
    s = h.to_device(ctx, bit_offset=5).slice(13, 5000)
    ex2 = ffi.to_device_ffi(s)
    assert ex2.array.array.offset == 0 and ex2.array.array.buffers[1] == s.values.ptr
    assert ex2.array.array.buffers[0] != s.validity.ptr
    orc.assert_logical_eq(HostArray.from_device(ffi.from_device_ffi(ex2.array, ex2.schema, ctx, keepalive=ex2)),
                          h.slice(13, 5000), "re-aligned validity")
    # strings and booleans
    strs = A.Array.from_strings([f"s{i}" * (i % 3) for i in range(500)], [i % 4 != 0 for i in range(500)], ctx=ctx)
    ex3 = ffi.to_device_ffi(strs)
    assert ex3.array.array.n_buffers == 3 and ex3.schema.format == b"u"
    assert ffi.from_device_ffi(ex3.array, ex3.schema, ctx, keepalive=ex3).to_pylist() == strs.to_pylist()
    b = HostArray(A.Boolean, rng.random(999) < 0.5, rng.random(999) < 0.9).to_device(ctx, bit_offset=3)
    ex4 = ffi.to_device_ffi(b)
    assert ffi.from_device_ffi(ex4.array, ex4.schema, ctx, keepalive=ex4).to_pylist() == b.to_pylist()
    # errors: wrong device type / id, released struct
    da.device_type = 2  # CUDA
    with pytest.raises(A.CDataInterfaceError, match="not ROCm memory"):
        ffi.from_device_ffi(da, ex.schema, ctx)
    da.device_type, da.device_id = ffi.L.ARROW_DEVICE_ROCM, ctx.device + 1
    with pytest.raises(A.CDataInterfaceError, match="lives on device"):
        ffi.from_device_ffi(da, ex.schema, ctx)
    da.device_id = ctx.device
    ex.release()
    assert not ex.array.array.release
    with pytest.raises(A.CDataInterfaceError):
        ffi.from_device_ffi(ex.array, ex.schema, ctx)


def test_device_interface_moves_ownership(ctx):
    """`owned` != NULL: the kernel result's buffers move into the ArrowDeviceArray and are freed by its release
    callback (checked through the allocator hook: every buffer handed out is freed exactly once)."""
    L = A._lib
    live = {}
    arena = ctx.alloc(4 << 20)  # sub-allocated by the hook: pointers the library's own pool has never seen
    top = [0]

    @L.ALLOC_FN
    def alloc(user, nbytes):
        off = (top[0] + 255) & ~255
        top[0] = off + nbytes
        live[arena.ptr + off] = nbytes
        return arena.ptr + off

    @L.FREE_FN
    def free(user, ptr, nbytes):
        assert live.pop(ptr) == nbytes

    vals = A.Array.from_numpy(np.arange(5000, dtype=np.int64), np.arange(5000) % 3 != 0, ctx=ctx)
    mask = A.Array.from_numpy(np.arange(5000) % 2 == 0, ctx=ctx)
    ctx.lib.ah_context_set_allocator(ctx.handle, alloc, free, None)
    try:
        out = L.ArrayOut()
        vv, mv = vals.view(), mask.view()
        ctx.check(ctx.lib.ah_filter(ctx.handle, C.byref(vv), C.byref(mv), C.byref(out)))
        assert len(live) == 2  # values + validity of the result
        res_view = L.ArrayView()
        res_view.type, res_view.length, res_view.null_count = out.type, out.length, out.null_count
        res_view.values, res_view.validity = out.values, out.validity
        dev, sch = L.FFI_ArrowDeviceArray(), ffi.FFI_ArrowSchema()
        ctx.check(ctx.lib.ah_export_c_device_data(ctx.handle, C.byref(res_view), C.byref(out), None, C.byref(dev), C.byref(sch)))
        assert not out.values and not out.validity and len(live) == 2      # moved, nothing freed yet
        assert dev.array.length == 2500 and dev.array.buffers[1] in live
        C.CFUNCTYPE(None, C.c_void_p)(dev.array.release)(C.addressof(dev.array))
        assert live == {} and not dev.array.release
        C.CFUNCTYPE(None, C.c_void_p)(sch.release)(C.addressof(sch))
    finally:
        ctx.lib.ah_context_set_allocator(ctx.handle, L.ALLOC_FN(0), L.FREE_FN(0), None)


def test_import_survives_mutated_structs(ctx):
    """Round 4: the C Data Interface carries no buffer SIZES (the consumer must trust `length` /
    `offset`, exactly like arrow-array/src/ffi.rs:470-500), so what can be validated is everything else — and every such
    violation must come back as a status from the C entry point (AH_C_DATA_INTERFACE / AH_NOT_YET_IMPLEMENTED / ...),
    never as a crash: format strings (unknown, truncated, garbage bytes), buffer counts, negative length / offset, null
    buffer pointers, non-monotonic or negative string offsets, a released struct, a dictionary pointer."""
    L = A._lib
    rng = np.random.default_rng(99)
    sources = [pa.array(rng.integers(0, 100, 50), mask=rng.random(50) < 0.2), pa.array([None if i % 4 == 0 else "s" * (i % 7) for i in range(50)]),
               pa.array(rng.random(50) < 0.5), pa.array(rng.standard_normal(50)), pa.array([b"ab", None, b""] * 10, type=pa.large_binary())]
    formats = [b"", b"?", b"zz", b"tsq:", b"d:", b"d:abc", b"w:", b"+l", b"+s", b"\xff\xfe", b"t", b"td", b"tt", b"l" * 300, b"tsu", b"d:1", b"w:-4"]
    seen = {}
    spare = ffi.FFI_ArrowArray()
    for it in range(600):
        arr = sources[it % len(sources)]
        a, s_ = ffi.FFI_ArrowArray(), ffi.FFI_ArrowSchema()
        arr._export_to_c(C.addressof(a), C.addressof(s_))
        rel_a, rel_s, fmt0, bufs0 = a.release, s_.release, s_.format, [a.buffers[i] for i in range(a.n_buffers)]
        keep = []  # mutated format strings / offsets must outlive the call
        kind = int(rng.integers(0, 9))
        if kind == 0:
            buf = C.create_string_buffer(formats[int(rng.integers(0, len(formats)))])
            keep.append(buf)
            s_.format = C.cast(buf, C.c_char_p)
        elif kind == 1:
            a.n_buffers = int(rng.integers(-2, 6))
        elif kind == 2:
            a.length = -int(rng.integers(1, 100))
        elif kind == 3:
            a.offset = -int(rng.integers(1, 100))
        elif kind == 4 and a.n_buffers >= 2:
            a.buffers[int(rng.integers(1, a.n_buffers))] = None
        elif kind == 5 and pa.types.is_string(arr.type):
            offs = np.frombuffer((C.c_int32 * (len(arr) + 1)).from_address(a.buffers[1]), dtype=np.int32).copy()
            offs[int(rng.integers(0, len(offs)))] = int(rng.choice([-5, 2**30, -2**31]))
            offs[-1] = int(rng.choice([offs[-1], -1, 0]))
            keep.append(offs)
            a.buffers[1] = offs.ctypes.data
        elif kind == 6:
            a.dictionary = C.pointer(spare)
        elif kind == 7:
            a.release = None  # "already released"
        out = L.ArrayOut()
        st = ctx.lib.ah_import_c_data(ctx.handle, C.byref(a), C.byref(s_), C.byref(out))
        seen[(kind, st)] = seen.get((kind, st), 0) + 1
        assert st in (L.AH_OK, L.AH_C_DATA_INTERFACE, L.AH_NOT_YET_IMPLEMENTED, L.AH_INVALID_ARGUMENT, L.AH_CAST_ERROR, L.AH_PARSE_ERROR), (kind, st)
        if st == L.AH_OK:
            if kind == 8:
                assert A.Array._from_out(ctx, out, ffi.data_type_from_format(ctx, fmt0.decode())).to_pyarrow().equals(arr)
            else:
                ctx.lib.ah_array_release(ctx.handle, C.byref(out))
        elif kind in (1, 2, 3, 7):
            assert st == L.AH_C_DATA_INTERFACE, (kind, st)
        elif kind == 6:
            assert st == L.AH_NOT_YET_IMPLEMENTED, (kind, st)
        # give the producer back exactly what it exported, then release it
        a.release, s_.release, s_.format, a.dictionary = rel_a, rel_s, fmt0, None
        a.n_buffers = len(bufs0)
        for i, b in enumerate(bufs0):
            a.buffers[i] = b
        for st_ in (a, s_):
            C.CFUNCTYPE(None, C.c_void_p)(st_.release)(C.addressof(st_))
    ok = sum(v for (k, st), v in seen.items() if st == L.AH_OK)
    refused = sum(v for (k, st), v in seen.items() if st != L.AH_OK)
    assert ok >= 60 and refused >= 250, seen
    assert K.filter(A.Array.from_numpy(np.arange(10), ctx=ctx), A.Array.from_numpy(np.arange(10) % 2 == 0, ctx=ctx)).length == 5
