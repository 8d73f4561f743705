"""arrow::ipc for device-resident record batches — the mirror of ``arrow_ipc::writer::{StreamWriter, FileWriter}``
/ ``arrow_ipc::reader::{StreamReader, FileReader}`` (arrow-ipc/src/writer.rs:1419-1768, reader.rs:944-1560) over the
C ABI's ``ah_ipc_*`` entry points.

The body of every RecordBatch message is assembled (encode) or received (decode) as ONE contiguous HBM
buffer: writing a stream is one D2H copy per batch, reading one H2D copy per batch after which the columns
are zero-copy views of that buffer; between GPUs the same body is what ``distributed.Communicator`` ships.
"""
import ctypes as C
import io

import numpy as np

from . import _lib as L
from . import array as A
from . import ffi

CONTINUATION = b"\xff\xff\xff\xff"
ARROW_MAGIC = b"ARROW1"


class Field:
    """arrow_schema::Field (name, data_type, nullable)."""

    def __init__(self, name, data_type, nullable=True):
        self.name, self.data_type, self.nullable = name, data_type, nullable

    def __eq__(self, o):
        return (self.name, self.data_type, self.nullable) == (o.name, o.data_type, o.nullable)

    def __repr__(self):
        return f"Field({self.name!r}, {self.data_type}, nullable={self.nullable})"


class Schema:
    def __init__(self, fields):
        self.fields = list(fields)

    @classmethod
    def of(cls, batch):
        return cls([Field(n, c.data_type, True) for n, c in zip(batch.names, batch.columns)])

    @property
    def names(self):
        return [f.name for f in self.fields]

    def __eq__(self, o):
        return self.fields == o.fields

    def _c_fields(self, ctx):
        n = len(self.fields)
        arr = (L.IpcField * max(n, 1))()
        keep = []
        for i, f in enumerate(self.fields):
            fmt = ffi.format_of(f.data_type) or ctx.lib.ah_format_of_type(f.data_type.physical).decode()
            nb, fb = f.name.encode(), fmt.encode()
            keep += [nb, fb]
            arr[i].name, arr[i].format, arr[i].nullable = nb, fb, 1 if f.nullable else 0
        return arr, keep


def _take_host_bytes(ctx, ptr, n):
    data = C.string_at(ptr, n)
    ctx.lib.ah_host_free(ptr)
    return data


def schema_to_bytes(schema, ctx=None, alignment=64):
    """Framed Schema message (``IpcDataGenerator::schema_to_bytes``)."""
    ctx = ctx or A.default_context()
    arr, _keep = schema._c_fields(ctx)
    out, n = C.c_void_p(), C.c_int64()
    ctx.check(ctx.lib.ah_ipc_schema_message(ctx.handle, len(schema.fields), arr, alignment, C.byref(out), C.byref(n)))
    return _take_host_bytes(ctx, out, n.value)


def schema_from_bytes(msg, ctx=None):
    ctx = ctx or A.default_context()
    n, fields = C.c_int32(), C.POINTER(L.IpcField)()
    ctx.check(ctx.lib.ah_ipc_decode_schema(ctx.handle, msg, len(msg), C.byref(n), C.byref(fields)))
    try:
        return Schema([Field(fields[i].name.decode(), ffi.data_type_from_format(ctx, fields[i].format.decode()),
                             bool(fields[i].nullable)) for i in range(n.value)])
    finally:
        ctx.lib.ah_host_free(fields)


class _Body:
    """Owner of an encoded message body in HBM (released through the context allocator)."""

    def __init__(self, ctx, ptr, nbytes):
        self.ctx, self.ptr, self.nbytes = ctx, ptr, nbytes
        out = L.ArrayOut()
        out.values, out.values_bytes = ptr, max(nbytes, 8)
        self._owner = A._OutOwner(ctx, out)

    def to_bytes(self):
        host = np.empty(self.nbytes, dtype=np.uint8)
        if self.nbytes:
            self.ctx.check(self.ctx.lib.ah_memcpy_dtoh(self.ctx.handle, host.ctypes.data, self.ptr, self.nbytes))
        return host.tobytes()


def encode_batch(batch, alignment=64):
    """``record_batch_to_bytes``: (framed metadata bytes, body in HBM)."""
    cols = batch.columns
    ctx = cols[0].ctx if cols else A.default_context()
    views = (L.ArrayView * max(len(cols), 1))()
    for i, c in enumerate(cols):
        views[i] = c.view()
    meta, mlen, body, blen = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
    ctx.check(ctx.lib.ah_ipc_encode_batch(ctx.handle, len(cols), views, batch.num_rows(), alignment, C.byref(meta),
                                          C.byref(mlen), C.byref(body), C.byref(blen)))
    return _take_host_bytes(ctx, meta, mlen.value), _Body(ctx, body.value, blen.value)


def decode_batch(meta, body_ptr, body_len, schema, ctx=None, keepalive=()):
    """``RecordBatchDecoder``: columns are zero-copy views of the device body (kept alive by ``keepalive``)."""
    ctx = ctx or A.default_context()
    n = len(schema.fields)
    arr, _keep = schema._c_fields(ctx)
    outs = (L.ArrayOut * max(n, 1))()
    rows = C.c_int64()
    ctx.check(ctx.lib.ah_ipc_decode_batch(ctx.handle, meta, len(meta), body_ptr, body_len, n, arr, outs, C.byref(rows)))
    cols = []
    for i, f in enumerate(schema.fields):
        o = L.ArrayOut()
        C.memmove(C.byref(o), C.byref(outs[i]), C.sizeof(L.ArrayOut))
        cols.append(A.Array._from_out(ctx, o, f.data_type, keepalive=tuple(keepalive)))
    return A.RecordBatch(schema.names, cols, rows.value)


class StreamWriter:
    """``StreamWriter::try_new(writer, &schema)`` / ``write`` / ``finish`` (writer.rs:1419-1600)."""

    def __init__(self, sink, schema, ctx=None, alignment=64):
        self.sink, self.schema, self.alignment = sink, schema, alignment
        self.ctx = ctx or A.default_context()
        self.finished = False
        sink.write(schema_to_bytes(schema, self.ctx, alignment))

    def write(self, batch):
        if self.finished:
            raise A.IpcError("Cannot write record batch to stream writer as it is closed")
        if [c.data_type for c in batch.columns] != [f.data_type for f in self.schema.fields]:
            raise A.InvalidArgumentError("batch schema does not match the stream schema")
        meta, body = encode_batch(batch, self.alignment)
        self.sink.write(meta)
        self.sink.write(body.to_bytes())

    def finish(self):
        if self.finished:
            raise A.IpcError("Cannot write footer to stream writer as it is closed")
        self.sink.write(CONTINUATION + b"\x00\x00\x00\x00")
        self.finished = True


class StreamReader:
    """``StreamReader::try_new(reader, None)`` (reader.rs:1380): iterate device RecordBatches."""

    def __init__(self, source, ctx=None):
        self.src = io.BytesIO(source) if isinstance(source, (bytes, bytearray, memoryview)) else source
        self.ctx = ctx or A.default_context()
        first = self._next_message()
        if first is None:
            raise A.IpcError("Unexpected end of stream before the schema message")
        self.schema = schema_from_bytes(first, self.ctx)

    def _next_message(self):
        head = self.src.read(4)
        if len(head) < 4:
            return None  # end of file without an EOS marker is accepted like the reference (reader.rs:1518)
        if head == CONTINUATION:
            size = self.src.read(4)
            if len(size) < 4:
                return None
            head += size
        else:
            size = head  # legacy framing: a bare length
        n = int.from_bytes(size, "little", signed=True)
        if n == 0:
            return None  # end-of-stream marker
        if n < 0:  # `read(-1)` would swallow the rest of the stream; the reference reports an invalid metadata length
            raise A.IpcError(f"Invalid metadata length: {n}")
        meta = self.src.read(n)
        if len(meta) < n:
            raise A.IpcError("Unexpected end of stream inside a message")
        return head + meta

    def __iter__(self):
        return self

    def __next__(self):
        ctx = self.ctx
        while True:
            msg = self._next_message()
            if msg is None:
                raise StopIteration
            ht, blen = C.c_int32(), C.c_int64()
            ctx.check(ctx.lib.ah_ipc_message_info(ctx.handle, msg, len(msg), C.byref(ht), C.byref(blen)))
            body = self.src.read(blen.value)
            if len(body) < blen.value:
                raise A.IpcError("Unexpected end of stream inside a message body")
            if ht.value == 1:
                raise A.IpcError("Not expecting a schema when messages are read")
            dev = A.DeviceBuffer.from_numpy(ctx, np.frombuffer(body, dtype=np.uint8)) if blen.value else A.DeviceBuffer(ctx, 8)
            return decode_batch(msg, dev.ptr, blen.value, self.schema, ctx, keepalive=(dev,))


def write_stream(batches, schema=None, ctx=None, alignment=64):
    """All batches as one IPC stream (bytes)."""
    batches = list(batches)
    schema = schema or Schema.of(batches[0])
    sink = io.BytesIO()
    w = StreamWriter(sink, schema, ctx, alignment)
    for b in batches:
        w.write(b)
    w.finish()
    return sink.getvalue()


# ------------------------------------------------------------------------------------------------ file format
def _pad(n, alignment):
    return (-n) % alignment


class FileWriter:
    """``FileWriter::try_new(writer, &schema)`` / ``write`` / ``finish`` (writer.rs:1645-1768): "ARROW1" + padding,
    the schema message, one framed message + body per batch (their positions recorded as ``Block``s), the
    end-of-stream marker and the footer trailer built by ``ah_ipc_file_footer``."""

    def __init__(self, sink, schema, ctx=None, alignment=64):
        self.sink, self.schema, self.alignment = sink, schema, alignment
        self.ctx = ctx or A.default_context()
        self.finished = False
        self.blocks = []
        header = ARROW_MAGIC + b"\x00" * _pad(len(ARROW_MAGIC), alignment)
        sink.write(header)
        msg = schema_to_bytes(schema, self.ctx, alignment)
        sink.write(msg)
        self.block_offsets = len(header) + len(msg)

    def write(self, batch):
        if self.finished:
            raise A.IpcError("Cannot write record batch to file writer as it is closed")
        if [c.data_type for c in batch.columns] != [f.data_type for f in self.schema.fields]:
            raise A.InvalidArgumentError("batch schema does not match the file schema")
        meta, body = encode_batch(batch, self.alignment)
        self.sink.write(meta)
        self.sink.write(body.to_bytes())
        self.blocks.append((self.block_offsets, len(meta), body.nbytes))
        self.block_offsets += len(meta) + body.nbytes

    def finish(self):
        if self.finished:
            raise A.IpcError("Cannot write footer to file writer as it is closed")
        self.sink.write(CONTINUATION + b"\x00\x00\x00\x00")
        self.sink.write(file_footer(self.schema, self.blocks, self.ctx))
        self.finished = True


def file_footer(schema, blocks, ctx=None):
    """Footer trailer of an IPC file: [Footer flatbuffer][i32 length]["ARROW1"]; ``blocks`` = (offset, metadata
    length, body length) per record batch."""
    ctx = ctx or A.default_context()
    arr, _keep = schema._c_fields(ctx)
    bl = (L.IpcBlock * max(len(blocks), 1))()
    for i, (off, mlen, blen) in enumerate(blocks):
        bl[i].offset, bl[i].meta_data_length, bl[i].body_length = off, mlen, blen
    out, n = C.c_void_p(), C.c_int64()
    ctx.check(ctx.lib.ah_ipc_file_footer(ctx.handle, len(schema.fields), arr, len(blocks), bl, C.byref(out), C.byref(n)))
    return _take_host_bytes(ctx, out, n.value)


def read_footer(tail, ctx=None):
    """``read_footer_length`` + the Footer (reader.rs:944-1260) from the last bytes of a file:
    (Schema, [(offset, metadata length, body length)], footer length).  With only the last 10 bytes it returns
    (None, None, footer length) so the caller knows how much more to read."""
    ctx = ctx or A.default_context()
    flen = C.c_int64()
    tail = bytes(tail)
    ctx.check(ctx.lib.ah_ipc_decode_footer(ctx.handle, tail, len(tail), C.byref(flen), None, None, None, None))
    if len(tail) < flen.value + 10:
        return None, None, flen.value
    n, fields, nb, blocks = C.c_int32(), C.POINTER(L.IpcField)(), C.c_int32(), C.POINTER(L.IpcBlock)()
    ctx.check(ctx.lib.ah_ipc_decode_footer(ctx.handle, tail, len(tail), C.byref(flen), C.byref(n), C.byref(fields),
                                           C.byref(nb), C.byref(blocks)))
    try:
        schema = Schema([Field(fields[i].name.decode(), ffi.data_type_from_format(ctx, fields[i].format.decode()),
                               bool(fields[i].nullable)) for i in range(n.value)])
        bl = [(blocks[i].offset, blocks[i].meta_data_length, blocks[i].body_length) for i in range(nb.value)]
    finally:
        ctx.lib.ah_host_free(fields)
        ctx.lib.ah_host_free(blocks)
    return schema, bl, flen.value


class FileReader:
    """``FileReader::try_new(reader, None)`` (reader.rs:1290-1370): random access to the record batches of an IPC
    file through its footer; every batch is one H2D copy of its body, the columns are views of it."""

    def __init__(self, source, ctx=None):
        self.src = io.BytesIO(source) if isinstance(source, (bytes, bytearray, memoryview)) else source
        self.ctx = ctx or A.default_context()
        # like the reference (reader.rs:1226-1260) only the TRAILING magic is checked: the header is never read
        size = self.src.seek(0, io.SEEK_END)
        _, _, flen = read_footer(self._read_at(size - 10, 10) if size >= 10 else b"", self.ctx)
        self.schema, self.blocks, _ = read_footer(self._read_at(size - 10 - flen, flen + 10), self.ctx)

    def _read_at(self, pos, n):
        self.src.seek(max(pos, 0))
        return self.src.read(n)

    def num_batches(self):
        return len(self.blocks)

    def read_batch(self, i):
        ctx = self.ctx
        off, mlen, blen = self.blocks[i]
        msg = self._read_at(off, mlen)
        if len(msg) < mlen:
            raise A.IpcError("Unexpected end of file inside a message")
        body = self._read_at(off + mlen, blen)
        if len(body) < blen:
            raise A.IpcError("Unexpected end of file inside a message body")
        dev = A.DeviceBuffer.from_numpy(ctx, np.frombuffer(body, dtype=np.uint8)) if blen else A.DeviceBuffer(ctx, 8)
        return decode_batch(msg, dev.ptr, blen, self.schema, ctx, keepalive=(dev,))

    def __iter__(self):
        return (self.read_batch(i) for i in range(len(self.blocks)))


def write_file(batches, schema=None, ctx=None, alignment=64):
    """All batches as one IPC file (bytes)."""
    batches = list(batches)
    schema = schema or Schema.of(batches[0])
    sink = io.BytesIO()
    w = FileWriter(sink, schema, ctx, alignment)
    for b in batches:
        w.write(b)
    w.finish()
    return sink.getvalue()
