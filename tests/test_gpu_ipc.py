"""Arrow IPC framing of device-resident record batches (arrow-ipc/src/writer.rs, reader.rs).
The peer is pyarrow (Arrow C++): it must read what we write and we must read what it writes — the same
cross-implementation check the reference runs against the arrow-testing integration files."""
import ctypes as C
import decimal
import io

import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K, ffi, ipc

pa = pytest.importorskip("pyarrow")

pytestmark = pytest.mark.gpu


def _table(n, seed=0):
    rng = np.random.default_rng(seed)
    m = lambda p=0.8: ~(rng.random(n) < p)  # noqa: E731
    i64 = rng.integers(-2**40, 2**40, n)
    return pa.table({
        "i8": pa.array(rng.integers(-128, 127, n, dtype=np.int8), mask=m()),
        "u16": pa.array(rng.integers(0, 65535, n, dtype=np.uint16)),
        "i32": pa.array(rng.integers(-2**31, 2**31 - 1, n, dtype=np.int32), mask=m(0.5)),
        "i64": pa.array(i64, mask=m()),
        "f32": pa.array(rng.standard_normal(n).astype(np.float32), mask=m()),
        "f64": pa.array(rng.standard_normal(n)),
        "b": pa.array(rng.random(n) < 0.5, mask=m()),
        "bnn": pa.array(rng.random(n) < 0.5),
        "s": pa.array([None if i % 7 == 0 else "v" * (i % 19) + str(i) for i in range(n)]),
        "ls": pa.array([("x" * (i % 5)) for i in range(n)], type=pa.large_string()),
        "bin": pa.array([None if i % 5 == 0 else bytes([i % 256]) * (i % 9) for i in range(n)], type=pa.binary()),
        "ts": pa.array(i64, mask=m()).view(pa.timestamp("us", tz="UTC")),
        "d32": pa.array(rng.integers(0, 20000, n, dtype=np.int32), mask=m()).view(pa.date32()),
        "dec": pa.array([None if i % 11 == 0 else decimal.Decimal(int(v)).scaleb(-3) for i, v in enumerate(i64)],
                        type=pa.decimal128(20, 3)),
        "allnull": pa.array([None] * n, type=pa.int32()),
    })


def _device_batches(tbl, ctx, max_chunksize):
    return [ffi.from_pyarrow(b, ctx) for b in tbl.to_batches(max_chunksize=max_chunksize)]


@pytest.mark.parametrize("alignment", [8, 64])
def test_pyarrow_reads_our_stream(ctx, alignment):
    tbl = _table(5000, 1)
    batches = _device_batches(tbl, ctx, 1777)
    stream = ipc.write_stream(batches, ctx=ctx, alignment=alignment)
    back = pa.ipc.open_stream(stream).read_all()
    assert back.schema.equals(tbl.schema)
    assert back.equals(tbl)
    back.validate(full=True)


def test_we_read_pyarrows_stream(ctx):
    tbl = _table(4000, 2)
    legacy = pa.ipc.IpcWriteOptions(use_legacy_format=True, metadata_version=pa.ipc.MetadataVersion.V4)
    for opts in (pa.ipc.IpcWriteOptions(), legacy):  # V5 framing and the pre-0.15 bare-length prefix
        sink = io.BytesIO()
        with pa.ipc.new_stream(sink, tbl.schema, options=opts) as w:
            for b in tbl.to_batches(max_chunksize=999):
                w.write_batch(b)
            w.write_batch(tbl.slice(17, 1234).to_batches()[0])  # sliced columns: offsets / bitmaps re-based by the writer
            w.write_batch(tbl.slice(0, 0).to_batches(max_chunksize=10)[0] if tbl.slice(0, 0).to_batches() else
                          pa.RecordBatch.from_pylist([], schema=tbl.schema))
        rdr = ipc.StreamReader(sink.getvalue(), ctx)
        assert [f.name for f in rdr.schema.fields] == tbl.schema.names
        got = [ffi.to_pyarrow(b) for b in rdr]
        want = tbl.to_batches(max_chunksize=999) + [tbl.slice(17, 1234).to_batches()[0]]
        assert len(got) == len(want) + 1 and got[-1].num_rows == 0
        for g, w_ in zip(got, want):
            assert g.equals(w_)


def test_sliced_device_arrays_are_rebased(ctx):
    """write_array_data truncates values to the slice, re-aligns bitmaps to bit 0 and rebases string offsets
    (writer.rs:2246-2290,:2331-2346,:2483-2489)."""
    tbl = _table(3000, 3)
    rb = ffi.from_pyarrow(tbl.to_batches()[0], ctx)
    for off, ln in ((1, 2999), (13, 100), (64, 640), (2999, 1), (5, 0)):
        sl = A.RecordBatch(rb.names, [c.slice(off, ln) for c in rb.columns], ln)
        back = pa.ipc.open_stream(ipc.write_stream([sl], ctx=ctx)).read_all()
        assert back.equals(tbl.slice(off, ln)), (off, ln)


def test_device_round_trip_is_zero_copy(ctx):
    """encode -> (metadata, one HBM body) -> decode: every column is a view into the body."""
    tbl = _table(2048, 4)
    rb = ffi.from_pyarrow(tbl.to_batches()[0], ctx)
    meta, body = ipc.encode_batch(rb, 64)
    assert len(meta) % 64 == 0 and body.nbytes % 64 == 0
    dec = ipc.decode_batch(meta, body.ptr, body.nbytes, ipc.Schema.of(rb), ctx, keepalive=(body,))
    assert dec.num_rows() == 2048
    for name, c, orig in zip(rb.names, dec.columns, rb.columns):
        assert body.ptr <= c.values.ptr < body.ptr + body.nbytes, name
        assert c.values.ptr % 64 == 0
        if c.validity is not None:
            assert body.ptr <= c.validity.ptr < body.ptr + body.nbytes and c.validity_bit_offset == 0
        assert (c.validity is None) == (orig.null_count() == 0)  # reader.rs:271: dropped when null_count == 0
        assert c.null_count() == orig.null_count()
    assert ffi.to_pyarrow(dec).equals(tbl.to_batches()[0])
    # the views feed kernels directly
    f = K.filter(dec.columns[3], dec.columns[7])
    assert f.to_pyarrow().equals(tbl["i64"].combine_chunks().filter(tbl["bnn"].combine_chunks()))


def test_underaligned_foreign_buffers_are_copied(ctx):
    """A body whose 16-byte values land on an 8-byte boundary (legal for an 8-byte-aligning writer): the column
    is copied into aligned memory like `align_buffers` (reader.rs:301) instead of being viewed."""
    n = 101  # validity = 13 bytes -> padded to 16; then i8 column of 101 bytes -> 104: decimals start at 8 mod 16
    tbl = pa.table({"a": pa.array(np.arange(n, dtype=np.int8)),
                    "d": pa.array([decimal.Decimal(i) for i in range(n)], type=pa.decimal128(10, 0))})
    sink = io.BytesIO()
    with pa.ipc.new_stream(sink, tbl.schema) as w:
        w.write_table(tbl)
    got = [ffi.to_pyarrow(b) for b in ipc.StreamReader(sink.getvalue(), ctx)]
    assert pa.Table.from_batches(got).equals(tbl)


def test_unsupported_and_malformed(ctx):
    tbl = pa.table({"a": pa.array(range(100))})
    for codec in ("lz4", "zstd"):
        sink = io.BytesIO()
        try:
            with pa.ipc.new_stream(sink, tbl.schema, options=pa.ipc.IpcWriteOptions(compression=codec)) as w:
                w.write_table(tbl)
        except pa.ArrowNotImplementedError:
            continue
        with pytest.raises(A.array.NotYetImplemented):
            list(ipc.StreamReader(sink.getvalue(), ctx))
    d = pa.table({"d": pa.array(["a", "b", "a"]).dictionary_encode()})
    sink = io.BytesIO()
    with pa.ipc.new_stream(sink, d.schema) as w:
        w.write_table(d)
    with pytest.raises(A.array.NotYetImplemented):
        ipc.StreamReader(sink.getvalue(), ctx)
    good = ipc.write_stream([ffi.from_pyarrow(tbl.to_batches()[0], ctx)], ctx=ctx)
    with pytest.raises(A.array.IpcError):
        list(ipc.StreamReader(good[:len(good) - 300], ctx))  # body cut short
    rb = ffi.from_pyarrow(tbl.to_batches()[0], ctx)
    meta, body = ipc.encode_batch(rb)
    with pytest.raises(A.array.IpcError):  # body shorter than the metadata announces
        ipc.decode_batch(meta, body.ptr, body.nbytes - 64, ipc.Schema.of(rb), ctx)
    wrong = ipc.Schema([ipc.Field("a", A.Int64), ipc.Field("b", A.Int64)])
    with pytest.raises(A.array.IpcError):  # more fields than nodes
        ipc.decode_batch(meta, body.ptr, body.nbytes, wrong, ctx)
    with pytest.raises(A.array.InvalidArgumentError, match="Alignment should be 8, 16, 32, or 64."):
        ipc.encode_batch(rb, alignment=24)
    w = ipc.StreamWriter(io.BytesIO(), ipc.Schema.of(rb), ctx)
    w.finish()
    with pytest.raises(A.array.IpcError):
        w.write(rb)


def test_large_batch_moves_as_one_body(ctx, oracle):
    """2^24 rows x (Int64 + Float64 + Boolean): the body is one allocation whose size is the sum of the padded
    buffers, and a filter on the decoded views equals the filter on the originals."""
    n = 1 << 24
    from orc import HostArray
    vals = oracle.gen_i64(n, 5, -10**9, 10**9)
    valid = oracle.gen_bits(n, 6, 0.9)
    mask = oracle.gen_bits(n, 7, 0.1)
    a = HostArray(A.Int64, vals, valid).to_device(ctx, bit_offset=3)
    b = HostArray(A.Float64, vals.astype(np.float64)).to_device(ctx)
    m = HostArray(A.Boolean, mask).to_device(ctx, bit_offset=5)
    rb = A.RecordBatch(["a", "b", "m"], [a, b, m], n)
    meta, body = ipc.encode_batch(rb)
    assert body.nbytes == 4 * (n // 8) + 2 * n * 8  # validity of a, b (all ones), m + m's value bits; 2 x 8-byte values
    dec = ipc.decode_batch(meta, body.ptr, body.nbytes, ipc.Schema.of(rb), ctx, keepalive=(body,))
    got = K.filter(dec.columns[0], dec.columns[2])
    want = K.filter(a, m)
    assert got.length == want.length and got.null_count() == want.null_count()
    assert np.array_equal(got.values_numpy(), want.values_numpy()) and np.array_equal(got.valid_mask(), want.valid_mask())


@pytest.mark.parametrize("alignment", [8, 64])
def test_ipc_file_format_both_directions(ctx, alignment):
    """FileWriter / FileReader (writer.rs:1645-1768, reader.rs:944-1370): pyarrow opens the file we write from
    device batches (random access through OUR footer); we open the file pyarrow writes (random access through ITS
    footer), every batch one H2D copy of its body."""
    tbl = _table(6000, 5)
    batches = _device_batches(tbl, ctx, 2048)
    data = ipc.write_file(batches, ctx=ctx, alignment=alignment)
    assert data[:6] == b"ARROW1" and data[-6:] == b"ARROW1"
    rd = pa.ipc.open_file(pa.py_buffer(data))
    assert rd.schema.equals(tbl.schema) and rd.num_record_batches == len(batches)
    want = tbl.to_batches(max_chunksize=2048)
    for i in (2, 0, 1):  # random access
        assert rd.get_batch(i).equals(want[i])
    assert rd.read_all().equals(tbl)
    # our reader on our own file, and on Arrow C++'s file
    mine = ipc.FileReader(data, ctx)
    assert mine.num_batches() == len(batches) and [f.name for f in mine.schema.fields] == tbl.schema.names
    assert ffi.to_pyarrow(mine.read_batch(1)).equals(want[1])
    sink = io.BytesIO()
    with pa.ipc.new_file(sink, tbl.schema) as w:
        for b in want:
            w.write_batch(b)
    theirs = ipc.FileReader(sink.getvalue(), ctx)
    assert theirs.num_batches() == len(want)
    for i in (2, 0, 1):
        assert ffi.to_pyarrow(theirs.read_batch(i)).equals(want[i])
    assert [ffi.to_pyarrow(b).num_rows for b in theirs] == [b.num_rows for b in want]
    # closed writer / bad magic: the reference's texts
    w = ipc.FileWriter(io.BytesIO(), ipc.Schema.of(batches[0]), ctx)
    w.finish()
    with pytest.raises(A.IpcError) as ei:
        w.write(batches[0])
    assert ei.value.message == "Cannot write record batch to file writer as it is closed"
    with pytest.raises(A.IpcError) as ei:
        w.finish()
    assert ei.value.message == "Cannot write footer to file writer as it is closed"
    assert ipc.FileReader(b"NOTARROW" + data[8:], ctx).num_batches() == len(batches)  # the header magic is never read
    with pytest.raises(A.ParseError) as ei:
        ipc.FileReader(data[:-3] + b"XYZ", ctx)
    assert ei.value.message == "Arrow file does not contain correct footer"


def _one_batch_stream(tbl):
    sink = io.BytesIO()
    with pa.ipc.new_stream(sink, tbl.schema) as w:
        w.write_batch(tbl.to_batches()[0])
    return bytearray(sink.getvalue())


def test_corrupt_stream_metadata_is_refused(ctx):
    """FieldNode / offsets sanity the reference gets from ArrayData validation
    (arrow-data/src/data.rs:730-790).  A hostile or damaged stream must end in IpcError, never in zero-copy columns
    whose later kernels read outside the message body."""
    n = 1000
    tbl = pa.table({"s": pa.array([("v" * (i % 13)) for i in range(n)]),
                    "i": pa.array(np.arange(n, dtype=np.int64), mask=(np.arange(n) % 5 == 0))})
    good = _one_batch_stream(tbl)
    assert sum(b.num_rows() for b in ipc.StreamReader(bytes(good), ctx)) == n

    def expect_refusal(stream, what):
        with pytest.raises((A.IpcError, A.ParseError)) as ei:
            list(ipc.StreamReader(bytes(stream), ctx))
        assert str(ei.value), what

    # FieldNode {length, null_count} of column "i": (1000, 200) as two little-endian i64
    node = np.array([n, 200], dtype="<i8").tobytes()
    at = bytes(good).find(node)
    assert at > 0
    for bad_nulls in (-1, n + 1, 2**62):
        s = bytearray(good)
        s[at + 8:at + 16] = np.array([bad_nulls], dtype="<i8").tobytes()
        expect_refusal(s, f"null_count {bad_nulls}")
    s = bytearray(good)
    s[at:at + 8] = np.array([-5], dtype="<i8").tobytes()
    expect_refusal(s, "negative length")
    s = bytearray(good)
    s[at:at + 8] = np.array([2**61], dtype="<i8").tobytes()  # (len + 1) * w would wrap in 64 bits
    expect_refusal(s, "overflowing length")
    # string offsets: the i32 run 0, 0, 1, 3, 6, ... sits in the body; break monotonicity / point past the data buffer
    offs = np.cumsum([0] + [i % 13 for i in range(n)]).astype("<i4").tobytes()
    at = bytes(good).find(offs)
    assert at > 0
    for slot, val in ((n, 2**30), (500, -7), (10, 2**20)):
        s = bytearray(good)
        s[at + 4 * slot:at + 4 * slot + 4] = np.array([val], dtype="<i4").tobytes()
        expect_refusal(s, f"offset slot {slot} = {val}")
    # a message length that is negative is an error, not "read the rest of the stream"
    s = bytearray(good)
    s[4:8] = np.array([-1], dtype="<i4").tobytes()
    with pytest.raises(A.IpcError):
        ipc.StreamReader(bytes(s), ctx)


def test_decimal_columns_keep_their_logical_type_through_ffi(ctx):
    """a 'd:p,s' column arriving through the C Data Interface / IPC must be the typed
    Decimal128 (descriptor with precision and scale), so arithmetic and casts on it take the decimal path."""
    vals = [decimal.Decimal(v).scaleb(-2) for v in (12345, -250, 99999, 1)]
    col = pa.array(vals + [None], type=pa.decimal128(12, 2))
    d = ffi.from_pyarrow(col, ctx)
    assert d.data_type == A.Decimal128(12, 2) and d.data_type.logical is not None
    assert d.data_type.descriptor().precision == 12 and d.data_type.descriptor().scale == 2
    s = K.add(d, d)
    assert s.data_type == A.Decimal128(13, 2)
    back = ffi.to_pyarrow(s)
    assert back.to_pylist() == [v * 2 for v in vals] + [None]
    native = A.Array.from_pylist([12345, -250, 99999, 1, None], A.Decimal128(12, 2), ctx)  # a natively typed operand
    assert ffi.to_pyarrow(K.add(native, d)).to_pylist() == [v * 2 for v in vals] + [None]
    wide = ffi.from_pyarrow(pa.array([decimal.Decimal(7), None], type=pa.decimal256(40, 0)), ctx)
    assert wide.data_type == A.Decimal256(40, 0) and wide.data_type.width == 32
    assert ffi.to_pyarrow(K.filter(wide, ffi.from_pyarrow(pa.array([True, True]), ctx))).to_pylist() == [decimal.Decimal(7), None]


def test_decode_batch_survives_mutated_metadata(ctx):
    """Round 4: a byte-mutation fuzz of ``ah_ipc_decode_batch`` — the one IPC decoder that needs a
    device (its columns are zero-copy views of the body in HBM and string offsets are validated by a kernel).  A real
    RecordBatch message of {Int64 with nulls, Utf8, Boolean, Float64} is mutated 1 500 times (bit flips, boundary values
    in 32- / 64-bit words, truncations); every call must return a status.  Whatever is still ACCEPTED must be safe to
    use: every column is run through a kernel that touches all of its buffers (null count + filter + for strings a
    gather), with the allocator's red zones on in a child... here: results are only required not to fault the device
    (a fault would fail every later test of this session)."""
    rng = np.random.default_rng(1234)
    n = 4000
    cols = [A.Array.from_numpy(rng.integers(-2**62, 2**62, n), rng.random(n) < 0.8, ctx=ctx),
            A.Array.from_strings([("s" * int(k)) + str(i) for i, k in enumerate(rng.integers(0, 9, n))], rng.random(n) < 0.9, A.Utf8, ctx),
            A.Array.from_numpy(rng.random(n) < 0.5, ctx=ctx),
            A.Array.from_numpy(rng.standard_normal(n), ctx=ctx)]
    batch = A.RecordBatch(["i", "s", "b", "f"], cols, n)
    schema = ipc.Schema.of(batch)
    meta, body = ipc.encode_batch(batch, 64)
    good = ipc.decode_batch(meta, body.ptr, body.nbytes, schema, ctx, keepalive=(body,))
    assert good.num_rows() == n and good.columns[1].to_pylist() == cols[1].to_pylist()
    boundary = [0, 1, -1, 7, 8, 64, 255, 65535, 2**31 - 1, -2**31, 2**31, 2**40, 2**62, -2**62, n, n + 1, n - 1, body.nbytes, body.nbytes + 1]
    mask = A.Array.from_numpy(rng.random(n) < 0.3, ctx=ctx)
    accepted = refused = 0
    for it in range(1500):
        m = bytearray(meta)
        for _ in range(int(rng.integers(1, 4))):
            kind = int(rng.integers(0, 5))
            if len(m) == 0:  # (an earlier truncation left nothing to mutate)
                break
            if kind == 0:
                m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1 and len(m) >= 8:
                at = int(rng.integers(0, len(m) // 8)) * 8
                m[at:at + 8] = int(boundary[int(rng.integers(0, len(boundary)))]).to_bytes(8, "little", signed=True)
            elif kind == 2 and len(m) >= 4:
                at = int(rng.integers(0, len(m) // 4)) * 4
                v = int(boundary[int(rng.integers(0, 12))])
                m[at:at + 4] = (v & 0xFFFFFFFF).to_bytes(4, "little")
            elif kind == 3:
                m = m[:int(rng.integers(0, len(m) + 1))]
            elif len(m):
                at = int(rng.integers(0, len(m)))
                m[at:at + 6] = bytes(rng.integers(0, 256, 6, dtype=np.uint8))[:max(0, min(6, len(m) - at))]
        try:
            got = ipc.decode_batch(bytes(m), body.ptr, body.nbytes, schema, ctx, keepalive=(body,))
        except (A.ArrowError, A.array.HipError):
            refused += 1
            continue
        accepted += 1
        # an accepted message must describe buffers INSIDE the body: use every column
        for c in got.columns:
            if c.length == 0:
                continue
            c.null_count()
            if c.length <= mask.length:
                f = K.filter(c, mask.slice(0, c.length))
                assert f.length <= c.length
    assert refused > 300 and accepted > 50, (accepted, refused)
    # the context still works
    assert K.filter(cols[0], mask).length == int(np.asarray(mask.values_numpy()).sum())
