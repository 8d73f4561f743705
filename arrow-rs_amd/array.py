"""Host-side mirror of the reference's array types for arrays living in HBM.

Mirrors (names, argument meaning, error behaviour):
  * ``ArrowError``            arrow-schema/src/error.rs:26-69
  * ``DataType``              arrow-schema/src/datatype.rs (only the physical layouts of the hot path)
  * ``Array`` (PrimitiveArray / BooleanArray / GenericStringArray)
                              arrow-array/src/array/primitive_array.rs:596-601,
                              arrow-array/src/array/boolean_array.rs:68
  * ``Scalar`` / ``Datum``    arrow-array/src/scalar.rs:78-149

Values, validity and offsets are DEVICE buffers; nothing here touches device
bytes on the host except the explicit ``to_numpy`` / ``to_pylist`` copies.
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib as L


class ArrowError(Exception):
    """ArrowError (arrow-schema/src/error.rs:26-69). ``str()`` == Rust ``Display``."""
    variant = "ArrowError"
    prefix = ""

    def __init__(self, message):
        super().__init__(message)
        self.message = message

    def __str__(self):
        return f"{self.prefix}{self.message}"


class InvalidArgumentError(ArrowError):
    variant, prefix = "InvalidArgumentError", "Invalid argument error: "


class ComputeError(ArrowError):
    variant, prefix = "ComputeError", "Compute error: "


class ArithmeticOverflow(ArrowError):
    variant, prefix = "ArithmeticOverflow", "Arithmetic overflow: "


class DivideByZero(ArrowError):
    variant, prefix = "DivideByZero", ""

    def __str__(self):
        return "Divide by zero error"


class CastError(ArrowError):
    variant, prefix = "CastError", "Cast error: "


class OffsetOverflowError(ArrowError):
    variant, prefix = "OffsetOverflowError", "Offset overflow error: "


class CDataInterfaceError(ArrowError):
    variant, prefix = "CDataInterface", "C Data interface error: "


class IpcError(ArrowError):
    variant, prefix = "IpcError", "Ipc error: "


class ParseError(ArrowError):
    variant, prefix = "ParseError", "Parser error: "


class NotYetImplemented(ArrowError):
    variant, prefix = "NotYetImplemented", "Not yet implemented: "


class Panic(RuntimeError):
    """The reference would ``panic!`` here (e.g. out-of-bounds take without
    check_bounds, take.rs:447,454; offset overflow, generic_bytes_builder.rs:86-87)."""


class HipError(RuntimeError):
    pass


_STATUS = {
    L.AH_INVALID_ARGUMENT: InvalidArgumentError,
    L.AH_COMPUTE_ERROR: ComputeError,
    L.AH_ARITHMETIC_OVERFLOW: ArithmeticOverflow,
    L.AH_DIVIDE_BY_ZERO: DivideByZero,
    L.AH_CAST_ERROR: CastError,
    L.AH_NOT_YET_IMPLEMENTED: NotYetImplemented,
    L.AH_OFFSET_OVERFLOW_ERROR: OffsetOverflowError,
    L.AH_C_DATA_INTERFACE: CDataInterfaceError,
    L.AH_IPC_ERROR: IpcError,
    L.AH_PARSE_ERROR: ParseError,
    L.AH_OFFSET_OVERFLOW: Panic,
    L.AH_PANIC: Panic,
}


def raise_for_status(status, message):
    if status == L.AH_OK:
        return
    exc = _STATUS.get(status)
    if exc is None:
        raise HipError(f"status {status}: {message}")
    raise exc(message)


# --------------------------------------------------------------------- types
class DataType:
    """Logical type name + physical layout.  Logical types sharing a layout are
    preserved through filter/take exactly as the reference does
    (filter.rs:783-787, take.rs:414)."""

    def __init__(self, name, physical, np_dtype, logical=None):
        self.name = name
        self.physical = physical
        self.np_dtype = np_dtype
        # (AH_DT_* id, TimeUnit, timezone text or None) for the types whose CAST arithmetic depends on the
        # logical type (include/arrow_hip.h ah_data_type); None = the physical type says it all
        self.logical = logical

    def descriptor(self):
        """``ah_data_type`` of this type.  A Timestamp's zone must be a fixed offset (``Tz::from_str`` without the
        reference's optional chrono-tz feature, arrow-array/src/timezone.rs:280-290)."""
        d = L.DataTypeDesc()
        if self.logical is None:
            d.id = self.physical
            return d
        d.id, d.unit, tz = self.logical[:3]
        if len(self.logical) == 5:
            d.precision, d.scale = self.logical[3:]
        if tz is not None:
            d.has_tz, d.tz_offset_seconds = 1, parse_fixed_offset(tz)
        return d

    def __eq__(self, other):
        return isinstance(other, DataType) and self.name == other.name

    def __hash__(self):
        return hash(self.name)

    def __repr__(self):
        return self.name

    @property
    def width(self):
        return {L.AH_BOOL: 0, L.AH_FIXED16: 16, L.AH_FIXED32: 32, L.AH_UTF8_VIEW: 16, L.AH_BINARY_VIEW: 16}.get(
            self.physical, np.dtype(self.np_dtype).itemsize if self.np_dtype is not None else -1)

    def is_primitive(self):
        return self.physical not in (L.AH_BOOL, L.AH_UTF8, L.AH_LARGE_UTF8)


Boolean = DataType("Boolean", L.AH_BOOL, np.bool_)
Int8 = DataType("Int8", L.AH_INT8, np.int8)
Int16 = DataType("Int16", L.AH_INT16, np.int16)
Int32 = DataType("Int32", L.AH_INT32, np.int32)
Int64 = DataType("Int64", L.AH_INT64, np.int64)
UInt8 = DataType("UInt8", L.AH_UINT8, np.uint8)
UInt16 = DataType("UInt16", L.AH_UINT16, np.uint16)
UInt32 = DataType("UInt32", L.AH_UINT32, np.uint32)
UInt64 = DataType("UInt64", L.AH_UINT64, np.uint64)
Float16 = DataType("Float16", L.AH_FLOAT16, np.float16)
Float32 = DataType("Float32", L.AH_FLOAT32, np.float32)
Float64 = DataType("Float64", L.AH_FLOAT64, np.float64)
Utf8 = DataType("Utf8", L.AH_UTF8, None)
LargeUtf8 = DataType("LargeUtf8", L.AH_LARGE_UTF8, None)
# ByteView (arrow-data/src/byte_view.rs): u128 = length:i32 | prefix:4 bytes | buffer_index:i32 | offset:i32,
# or length + up to 12 inlined bytes
_VIEW = np.dtype([("length", "<i4"), ("prefix", "<u4"), ("buffer_index", "<i4"), ("offset", "<i4")])
Utf8View = DataType("Utf8View", L.AH_UTF8_VIEW, _VIEW)
BinaryView = DataType("BinaryView", L.AH_BINARY_VIEW, _VIEW)
MAX_INLINE_VIEW_LEN = 12
# logical types over the same physical layouts (the 14 temporal types of
# filter.rs:1090-1174 and the Duration/Decimal128 cases of take.rs:1263-1625)
SECOND, MILLISECOND, MICROSECOND, NANOSECOND = 0, 1, 2, 3  # TimeUnit (AH_SECOND ..)
_UNIT = ["Second", "Millisecond", "Microsecond", "Nanosecond"]


def parse_fixed_offset(tz):
    """``parse_fixed_offset`` (arrow-array/src/timezone.rs:25-49): "+09:00", "-09" or "+0930" -> seconds east of UTC;
    anything else is the reference's ParseError for builds without chrono-tz."""
    b = tz.encode()
    if len(b) == 6 and b[3:4] == b":":
        digits = b[1:3] + b[4:6]
    elif len(b) == 5:
        digits = b[1:5]
    elif len(b) == 3:
        digits = b[1:3] + b"00"
    else:
        digits = b"x"
    if not digits.isdigit() or b[0:1] not in (b"+", b"-"):
        raise ParseError(f'Invalid timezone "{tz}": only offset based timezones supported without chrono-tz feature')
    secs = int(digits[0:2]) * 3600 + int(digits[2:4]) * 60
    if secs >= 86400:  # FixedOffset::east_opt / west_opt
        raise ParseError(f'Invalid timezone "{tz}": only offset based timezones supported without chrono-tz feature')
    return secs if b[0:1] == b"+" else -secs


def Timestamp(unit, tz=None):
    """``DataType::Timestamp(unit, tz)``; the name is the reference's Debug form."""
    z = "None" if tz is None else f'Some("{tz}")'
    return DataType(f"Timestamp({_UNIT[unit]}, {z})", L.AH_INT64, np.int64, (L.AH_DT_TIMESTAMP, unit, tz))


def Duration(unit):
    return DataType(f"Duration({_UNIT[unit]})", L.AH_INT64, np.int64, (L.AH_DT_DURATION, unit, None))


def Time32(unit):
    return DataType(f"Time32({_UNIT[unit]})", L.AH_INT32, np.int32, (L.AH_DT_TIME32, unit, None))


def Time64(unit):
    return DataType(f"Time64({_UNIT[unit]})", L.AH_INT64, np.int64, (L.AH_DT_TIME64, unit, None))


def data_type_from_descriptor(d, like=None):
    """``ah_data_type`` -> DataType.  `like` supplies what the descriptor cannot carry: the zone TEXT of a Timestamp
    result (``array.with_timezone_opt(l.timezone())``, numeric.rs:536)."""
    if d.id == L.AH_DT_TIMESTAMP:
        tz = like.logical[2] if (like is not None and like.logical and like.logical[0] == L.AH_DT_TIMESTAMP) else None
        return Timestamp(d.unit, tz)
    if d.id == L.AH_DT_DECIMAL128:
        return Decimal128(d.precision, d.scale)
    if d.id == L.AH_DT_DURATION:
        return Duration(d.unit)
    if d.id == L.AH_DT_TIME32:
        return Time32(d.unit)
    if d.id == L.AH_DT_TIME64:
        return Time64(d.unit)
    if d.id == L.AH_DT_DATE32:
        return Date32
    if d.id == L.AH_DT_DATE64:
        return Date64
    return like if (like is not None and like.physical == d.id) else _PHYSICAL_DEFAULT[d.id]


Date32 = DataType("Date32", L.AH_INT32, np.int32, (L.AH_DT_DATE32, 0, None))
Date64 = DataType("Date64", L.AH_INT64, np.int64, (L.AH_DT_DATE64, 0, None))
Time32Second, Time32Millisecond = Time32(SECOND), Time32(MILLISECOND)
Time64Microsecond, Time64Nanosecond = Time64(MICROSECOND), Time64(NANOSECOND)
DurationSecond, DurationMillisecond = Duration(SECOND), Duration(MILLISECOND)
DurationMicrosecond, DurationNanosecond = Duration(MICROSECOND), Duration(NANOSECOND)
TimestampSecond, TimestampMillisecond = Timestamp(SECOND), Timestamp(MILLISECOND)
TimestampMicrosecond, TimestampNanosecond = Timestamp(MICROSECOND), Timestamp(NANOSECOND)
_DEC128 = np.dtype([("lo", "<u8"), ("hi", "<i8")])


def Decimal128(precision, scale):
    return DataType(f"Decimal128({precision}, {scale})", L.AH_FIXED16, _DEC128,
                    (L.AH_DT_DECIMAL128, 0, None, precision, scale))


_DEC256 = np.dtype([("w0", "<u8"), ("w1", "<u8"), ("w2", "<u8"), ("w3", "<i8")])


def Decimal256(precision, scale):
    """``DataType::Decimal256``: i256 natives (arrow-buffer/src/bigint/mod.rs), little-endian 4 x u64.  A 32-byte
    fixed-width value for the width-generic selection kernels (filter_native / take_native, filter.rs:731-770)."""
    return DataType(f"Decimal256({precision}, {scale})", L.AH_FIXED32, _DEC256)


_PHYSICAL_DEFAULT = {
    L.AH_BOOL: Boolean, L.AH_INT8: Int8, L.AH_INT16: Int16, L.AH_INT32: Int32, L.AH_INT64: Int64,
    L.AH_UINT8: UInt8, L.AH_UINT16: UInt16, L.AH_UINT32: UInt32, L.AH_UINT64: UInt64,
    L.AH_FLOAT16: Float16, L.AH_FLOAT32: Float32, L.AH_FLOAT64: Float64,
    L.AH_UTF8: Utf8, L.AH_LARGE_UTF8: LargeUtf8,
}


# ------------------------------------------------------------------- context
class Context:
    """One ah_context (one HIP stream + pooled HBM allocator) on one GPU."""

    def __init__(self, device=0):
        self.lib = L.load()
        h = C.c_void_p()
        st = self.lib.ah_context_create(int(device), C.byref(h))
        if st != L.AH_OK:
            raise HipError(
                f"ah_context_create(device={device}) failed with status {st}: no usable MI355X/HIP "
                "device. arrow_rs_amd has no CPU fallback.")
        self.handle = h
        self.device = device
        self._finalizer = weakref.finalize(self, self.lib.ah_context_destroy, h)

    def check(self, status):
        if status != L.AH_OK:
            raise_for_status(status, self.lib.ah_last_error(self.handle).decode())

    def synchronize(self):
        self.check(self.lib.ah_synchronize(self.handle))

    # opt-in asynchronous mode (ah_context_set_deferred, include/arrow_hip.h): infallible fixed-shape kernels
    # only enqueue; their results carry an unknown null count that Array.null_count() resolves lazily
    def set_deferred(self, on=True):
        self.lib.ah_context_set_deferred(self.handle, 1 if on else 0)

    @property
    def deferred(self):
        return bool(self.lib.ah_context_deferred(self.handle))

    def deferred_mode(self):
        """``with ctx.deferred_mode(): ...`` — deferred inside, synchronised and back to synchronous after."""
        ctx = self

        class _Scope:
            def __enter__(self):
                self.prev = ctx.deferred
                ctx.set_deferred(True)
                return ctx

            def __exit__(self, *exc):
                ctx.set_deferred(self.prev)
                if not self.prev:
                    ctx.synchronize()
                return False

        return _Scope()

    # raw buffers
    def graph_capture(self):
        """``with ctx.graph_capture() as g: ...deferred calls...`` records the calls made inside the block into a hipGraph
        (``ah_graph_begin`` / ``ah_graph_end``) instead of running them; afterwards ``g.launch()`` replays the whole sequence
        over the current bytes of the captured inputs, into the outputs the recorded calls returned.  Deferred mode is on
        inside the block; entry points that must wait on the device fail fast there."""
        return _GraphCapture(self)

    def memory_stats(self, reset_peaks=False):
        """``ah_context_stats``: live / high-water / cached bytes of the pooled device allocator and its call counts — the
        reference's ``MemoryPool::used`` for HBM (arrow-buffer/src/pool.rs:73-93)."""
        st = L.ContextStats()
        self.check(self.lib.ah_context_stats(self.handle, C.byref(st), int(reset_peaks)))
        return {n: int(getattr(st, n)) for n, _ in st._fields_}

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def profile(self, on=True):
        self.lib.ah_profile_enable(self.handle, 1 if on else 0)

    def profile_reset(self):
        self.lib.ah_profile_reset(self.handle)

    def profile_get(self, kernel):
        ms, n = C.c_double(), C.c_int64()
        self.check(self.lib.ah_profile_get(self.handle, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


_default_ctx = None


class _GraphCapture:
    def __init__(self, ctx):
        self.ctx, self._h, self.keep = ctx, None, []

    def __enter__(self):
        self.ctx.check(self.ctx.lib.ah_graph_begin(self.ctx.handle))
        return self

    def __exit__(self, et, ev, tb):
        h = C.c_void_p()
        st = self.ctx.lib.ah_graph_end(self.ctx.handle, C.byref(h))
        if et is None:
            self.ctx.check(st)
            self._h = h
            self._fin = weakref.finalize(self, self.ctx.lib.ah_graph_destroy, self.ctx.handle, h)
        elif st == 0:
            self.ctx.lib.ah_graph_destroy(self.ctx.handle, h)
        return False

    def node_count(self):
        return int(self.ctx.lib.ah_graph_node_count(self._h))

    def launch(self):
        """one replay, enqueued (no wait): ``ctx.synchronize()`` or any synchronous call orders behind it"""
        self.ctx.check(self.ctx.lib.ah_graph_launch(self.ctx.handle, self._h))


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def set_default_context(ctx):
    global _default_ctx
    _default_ctx = ctx


class DeviceBuffer:
    """An HBM allocation owned through the context pool (the analogue of
    ``Buffer::from_custom_allocation``, arrow-buffer/src/buffer/immutable.rs:170-176)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        ctx.check(ctx.lib.ah_device_alloc(ctx.handle, max(self.nbytes, 8), C.byref(p)))
        self.ptr = p.value
        self._finalizer = weakref.finalize(self, ctx.lib.ah_device_free, ctx.handle, C.c_void_p(self.ptr))

    @classmethod
    def from_numpy(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        buf = cls(ctx, arr.nbytes)
        if arr.nbytes:
            ctx.check(ctx.lib.ah_memcpy_htod(ctx.handle, buf.ptr, arr.ctypes.data, arr.nbytes))
        return buf

    def to_numpy(self, dtype=np.uint8, byte_offset=0, count=None):
        dt = np.dtype(dtype)
        n = (self.nbytes - byte_offset) // dt.itemsize if count is None else count
        out = np.empty(n, dtype=dt)
        if n:
            self.ctx.check(self.ctx.lib.ah_memcpy_dtoh(self.ctx.handle, out.ctypes.data,
                                                       self.ptr + byte_offset, n * dt.itemsize))
        return out

    # zero-copy hand-off to torch (RCCL collectives): __cuda_array_interface__ v2
    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2,
                "strides": None}


_SHRINK_MIN_BYTES = 256 << 10   # below this the slack is not worth a copy
_SHRINK_SLACK = 4               # allocation > 4 x what the rows need


class _OutOwner:
    """Keeps an ah_array_out alive; releases its buffers with ah_array_release."""

    def __init__(self, ctx, out, keepalive=()):
        self.ctx = ctx
        self.out = out
        self.keepalive = keepalive  # inputs a BORROWED result aliases
        self._finalizer = weakref.finalize(self, _release_out, ctx.lib, ctx.handle, out)


def _release_out(lib, handle, out):
    lib.ah_array_release(handle, C.byref(out))


class _RawMem:
    """Pointer + owner pair used by Array (owner: DeviceBuffer | _OutOwner | any keepalive)."""
    __slots__ = ("ptr", "nbytes", "owner")

    def __init__(self, ptr, nbytes, owner):
        self.ptr, self.nbytes, self.owner = ptr, nbytes, owner


class _OutBuffer:
    """A device buffer owned by an ah_array_out (a produced view array's data buffer): ptr / nbytes / to_numpy like
    DeviceBuffer, alive as long as the owner."""

    def __init__(self, ctx, ptr, nbytes, owner):
        self.ctx, self.ptr, self.nbytes, self.owner = ctx, int(ptr), int(nbytes), owner

    def to_numpy(self, dtype=np.uint8):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        if out.nbytes:
            self.ctx.check(self.ctx.lib.ah_memcpy_dtoh(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))
        return out


# AH_DEBUG_GUARD=1 (csrc/context.hip: every buffer ends against an unmapped page): uploaded bitmaps are then NOT padded to whole
# words, so a kernel that reads bitmap bytes the producer never promised is caught (tests/test_gpu_guard.py)
_TIGHT_BITMAPS = os.environ.get("AH_TIGHT_BITMAPS", os.environ.get("AH_DEBUG_GUARD", "0")) == "1"  # AH_TIGHT_BITMAPS=0/1 overrides


def pack_bits(bools, bit_offset=0):
    """LSB-first bit packing with a leading bit offset; padded to whole u64 words
    (BooleanBuffer layout, arrow-buffer/src/buffer/boolean.rs:97-104)."""
    bools = np.asarray(bools, dtype=bool)
    total = bit_offset + len(bools)
    padded = np.zeros(((total + 63) // 64) * 64, dtype=bool)
    padded[bit_offset:total] = bools
    packed = np.packbits(padded, bitorder="little")
    if _TIGHT_BITMAPS:  # what Arrow guarantees a consumer: ceil(bits / 8) bytes, nothing behind them
        packed = packed[:(total + 7) // 8].copy()
    return packed


def unpack_bits(byts, bit_offset, length):
    bits = np.unpackbits(np.asarray(byts, dtype=np.uint8), bitorder="little")
    return bits[bit_offset:bit_offset + length].astype(bool)


class Array:
    """A PrimitiveArray<T> / BooleanArray / GenericStringArray whose buffers live in HBM.

    ``values_ptr`` is already advanced by the slice offset for primitives
    (ScalarBuffer semantics, arrow-buffer/src/buffer/scalar.rs:188-209); boolean
    values and validity carry bit offsets (BooleanBuffer / NullBuffer)."""

    def __init__(self, ctx, data_type, length, values, values_bit_offset=0, validity=None,
                 validity_bit_offset=0, null_count=0, offsets=None):
        self.ctx = ctx
        self.data_type = data_type
        self.length = int(length)
        self.values = values            # _RawMem or None
        self.values_bit_offset = int(values_bit_offset)
        self.validity = validity        # _RawMem or None  (None == reference `nulls: None`)
        self.validity_bit_offset = int(validity_bit_offset)
        self._null_count = int(null_count)
        self.offsets = offsets          # _RawMem or None (strings)
        self.data_buffers = None        # view arrays: the variadic data buffers (list of DeviceBuffer)

    # ---- construction from host data
    @classmethod
    def from_numpy(cls, values, valid=None, data_type=None, ctx=None, bit_offset=0):
        """values: numpy array (bool -> BooleanArray). valid: bool mask (True=valid) or None.
        bit_offset: leading pad bits for the packed bitmaps (to exercise offsets)."""
        ctx = ctx or default_context()
        values = np.asarray(values)
        if data_type is None:
            if values.dtype == np.bool_:
                data_type = Boolean
            else:
                data_type = next(dt for dt in _PHYSICAL_DEFAULT.values()
                                 if dt.np_dtype is not None and np.dtype(dt.np_dtype) == values.dtype
                                 and dt.physical != L.AH_BOOL)
        n = len(values)
        if data_type.physical == L.AH_BOOL:
            vb = DeviceBuffer.from_numpy(ctx, pack_bits(values.astype(bool), bit_offset))
            vmem, vbo = _RawMem(vb.ptr, vb.nbytes, vb), bit_offset
        else:
            vb = DeviceBuffer.from_numpy(ctx, values.astype(data_type.np_dtype, copy=False))
            vmem, vbo = _RawMem(vb.ptr, vb.nbytes, vb), 0
        nmem, nulls = None, 0
        if valid is not None:
            valid = np.asarray(valid, dtype=bool)
            nb = DeviceBuffer.from_numpy(ctx, pack_bits(valid, bit_offset))
            nmem = _RawMem(nb.ptr, nb.nbytes, nb)
            nulls = int(n - valid.sum())
        return cls(ctx, data_type, n, vmem, vbo, nmem, bit_offset if nmem else 0, nulls)

    @classmethod
    def from_pylist(cls, items, data_type, ctx=None):
        """``PrimitiveArray::from(vec![Some(1), None, ...])``: null slots hold 0."""
        has_null = any(x is None for x in items)
        valid = np.array([x is not None for x in items], dtype=bool) if has_null else None
        if data_type.physical in (L.AH_UTF8, L.AH_LARGE_UTF8):
            return cls.from_strings([x if x is not None else "" for x in items], valid, data_type, ctx)
        if data_type.physical == L.AH_BOOL:
            vals = np.array([bool(x) if x is not None else False for x in items], dtype=bool)
        elif data_type.physical in (L.AH_FIXED16, L.AH_FIXED32):  # i128 / i256: two's complement, little-endian u64 limbs
            vals = np.zeros(len(items), dtype=data_type.np_dtype)
            names = data_type.np_dtype.names
            for i, x in enumerate(items):
                x = 0 if x is None else int(x)
                for k, nm in enumerate(names):
                    limb = (x >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
                    if k == len(names) - 1 and limb >= 1 << 63:  # the top limb is stored signed
                        limb -= 1 << 64
                    vals[nm][i] = limb
        else:
            vals = np.array([x if x is not None else 0 for x in items], dtype=data_type.np_dtype)
        return cls.from_numpy(vals, valid, data_type, ctx)

    @classmethod
    def from_strings(cls, strings, valid=None, data_type=None, ctx=None, bit_offset=0):
        """GenericStringArray: offsets (i32 / i64) + bytes (+ validity).  Null slots may carry
        bytes (the reference copies them through filter, filter.rs:890-892)."""
        ctx = ctx or default_context()
        data_type = data_type or Utf8
        odt = np.int32 if data_type.physical == L.AH_UTF8 else np.int64
        enc = [x.encode() if isinstance(x, str) else bytes(x) for x in strings]
        offs = np.zeros(len(enc) + 1, dtype=odt)
        if enc:
            offs[1:] = np.cumsum([len(b) for b in enc])
        data = np.frombuffer(b"".join(enc), dtype=np.uint8) if enc else np.empty(0, dtype=np.uint8)
        ob = DeviceBuffer.from_numpy(ctx, offs)
        db = DeviceBuffer.from_numpy(ctx, data)
        nmem, nulls = None, 0
        if valid is not None:
            valid = np.asarray(valid, dtype=bool)
            nb = DeviceBuffer.from_numpy(ctx, pack_bits(valid, bit_offset))
            nmem = _RawMem(nb.ptr, nb.nbytes, nb)
            nulls = int(len(enc) - valid.sum())
        return cls(ctx, data_type, len(enc), _RawMem(db.ptr, db.nbytes, db), 0, nmem,
                   bit_offset if nmem else 0, nulls, _RawMem(ob.ptr, ob.nbytes, ob))

    @classmethod
    def from_string_views(cls, items, data_type=None, ctx=None, bit_offset=0, block_size=8192):
        """StringViewArray / BinaryViewArray from str / bytes / None items, built the way
        GenericByteViewBuilder does (byte_view_array.rs:81-133): values <= 12 bytes are inlined, longer ones
        are appended to data blocks of ``block_size`` bytes and referenced by (buffer_index, offset)."""
        ctx = ctx or default_context()
        data_type = data_type or Utf8View
        views = np.zeros(len(items), dtype=_VIEW)
        raw = views.view(np.uint8).reshape(len(items), 16)
        blocks, cur = [], bytearray()
        for i, it in enumerate(items):
            if it is None:
                continue
            b = it.encode() if isinstance(it, str) else bytes(it)
            views["length"][i] = len(b)
            if len(b) <= MAX_INLINE_VIEW_LEN:
                raw[i, 4:4 + len(b)] = np.frombuffer(b, dtype=np.uint8)
            else:
                if len(cur) + len(b) > block_size and cur:
                    blocks.append(bytes(cur))
                    cur = bytearray()
                raw[i, 4:8] = np.frombuffer(b[:4], dtype=np.uint8)
                views["buffer_index"][i] = len(blocks)
                views["offset"][i] = len(cur)
                cur += b
        if cur:
            blocks.append(bytes(cur))
        valid = np.array([x is not None for x in items], dtype=bool) if any(x is None for x in items) else None
        vb = DeviceBuffer.from_numpy(ctx, views.view(np.uint8))
        nmem, nulls = None, 0
        if valid is not None:
            nb = DeviceBuffer.from_numpy(ctx, pack_bits(valid, bit_offset))
            nmem, nulls = _RawMem(nb.ptr, nb.nbytes, nb), int(len(items) - valid.sum())
        arr = cls(ctx, data_type, len(items), _RawMem(vb.ptr, vb.nbytes, vb), 0, nmem, bit_offset if nmem else 0, nulls)
        arr.data_buffers = [DeviceBuffer.from_numpy(ctx, np.frombuffer(b, dtype=np.uint8)) for b in blocks]
        return arr

    @classmethod
    def _from_out(cls, ctx, out, data_type, keepalive=()):
        w = data_type.width
        owner = _OutOwner(ctx, out, keepalive)  # first: if anything below raises, the buffers are still released
        if (w > 0 and out.flags == 0 and out.values and out.values_bytes >= _SHRINK_MIN_BYTES
                and out.values_bytes > _SHRINK_SLACK * max(out.length, 1) * w and not ctx.deferred):
            # the one-launch small filter allocates for the worst case (every row selected) so that it need not wait
            # for the count first; a long-lived, highly selective result would pin len / K times its size:
            # Buffer::shrink_to_fit it.  Results that are close to their allocation keep it (no copy on the fast path).
            # Not in deferred mode / while a graph is recorded: the shrink waits for the stream and copies.
            ctx.check(ctx.lib.ah_array_shrink_to_fit(ctx.handle, C.byref(out)))
        # `values_bytes` is the ALLOCATION (what release frees); the array itself is the first length * width bytes
        vbytes = min(out.values_bytes, out.length * w) if (w > 0 and out.flags == 0) else out.values_bytes
        vals = _RawMem(out.values, vbytes, owner) if out.values else None
        nmem = _RawMem(out.validity, out.validity_bytes, owner) if out.validity else None
        offs = _RawMem(out.offsets, out.offsets_bytes, owner) if out.offsets else None
        arr = cls(ctx, data_type, out.length, vals, out.values_bit_offset, nmem,
                  out.validity_bit_offset, out.null_count, offs)
        arr._owner = owner
        if data_type.physical in (L.AH_UTF8_VIEW, L.AH_BINARY_VIEW):
            if out.offsets or not keepalive:
                # a view array PRODUCED here (cast -> Utf8View): its one variadic data buffer travels in the `offsets`
                # fields of the ah_array_out (a view array has no offsets); no buffer at all when every string is inline
                arr.offsets = None
                arr.data_buffers = [_OutBuffer(ctx, out.offsets, out.offsets_bytes, owner)] if out.offsets else []
            else:
                # filter_byte_view / take_byte_view: same buffer list on the result (filter.rs:937, take.rs:638)
                src = next((k for k in keepalive if getattr(k, "data_buffers", None) is not None), None)
                arr.data_buffers = src.data_buffers if src is not None else []
        return arr

    # ---- Arrow C Data Interface / pyarrow (arrow-array/src/ffi.rs:231-254, arrow-pyarrow/src/lib.rs:199-257)
    @classmethod
    def from_pyarrow(cls, pa_array, ctx=None):
        from . import ffi
        return ffi.from_pyarrow(pa_array, ctx)

    def to_pyarrow(self):
        from . import ffi
        return ffi.to_pyarrow(self)

    # ---- reference API surface
    def __len__(self):
        return self.length

    def len(self):
        return self.length

    def is_empty(self):
        return self.length == 0

    def null_count(self):
        if self._null_count < 0:  # result of a deferred call: counted on first use (synchronises the stream)
            if self.validity is None:
                self._null_count = 0
            else:
                cnt = C.c_int64()
                self.ctx.check(self.ctx.lib.ah_count_set_bits(self.ctx.handle, self.validity.ptr,
                                                              self.validity_bit_offset, self.length, C.byref(cnt)))
                self._null_count = self.length - cnt.value
        return self._null_count

    def nulls(self):
        """``Array::nulls()``: None when the array carries no null buffer."""
        return self.validity

    def slice(self, offset, length):
        """``Array::slice`` (zero-copy).  Panics like the reference when out of range."""
        if offset < 0 or length < 0 or offset + length > self.length:  # usize arguments in the reference: never negative
            raise Panic("the length + offset of the sliced PrimitiveArray cannot exceed the existing length")
        w = self.data_type.width
        offs = self.offsets
        if self.data_type.physical in (L.AH_UTF8, L.AH_LARGE_UTF8):
            ow = 4 if self.data_type.physical == L.AH_UTF8 else 8
            vals, vbo = self.values, 0
            if offs is not None:
                offs = _RawMem(offs.ptr + offset * ow, (length + 1) * ow, offs.owner)
        elif self.data_type.physical == L.AH_BOOL:
            vals, vbo = self.values, self.values_bit_offset + offset
        elif self.values is None:
            vals, vbo = None, 0
        else:
            vals = _RawMem(self.values.ptr + offset * w, length * w, self.values.owner)
            vbo = 0
        nulls = 0
        if self.validity is not None:
            cnt = C.c_int64()
            self.ctx.check(self.ctx.lib.ah_count_set_bits(self.ctx.handle, self.validity.ptr,
                                                          self.validity_bit_offset + offset, length,
                                                          C.byref(cnt)))
            nulls = length - cnt.value
        out = Array(self.ctx, self.data_type, length, vals, vbo, self.validity,
                    self.validity_bit_offset + offset if self.validity else 0, nulls, offs)
        out.data_buffers = self.data_buffers
        return out

    def view(self):
        """The ah_array_view handed to the C ABI."""
        v = L.ArrayView()
        v.type = self.data_type.physical
        v.length = self.length
        v.null_count = self._null_count if self.validity is not None else 0
        v.values = self.values.ptr if self.values is not None else None
        v.values_bit_offset = self.values_bit_offset
        v.validity = self.validity.ptr if self.validity is not None else None
        v.validity_bit_offset = self.validity_bit_offset
        v.offsets = self.offsets.ptr if self.offsets is not None else None
        return v

    # Datum::get (arrow-array/src/scalar.rs:78-98)
    def get(self):
        return self, False

    # ---- host copies (tests / debugging only)
    def valid_mask(self):
        """numpy bool mask (True = valid), all-True when there is no null buffer."""
        if self.validity is None or self.length == 0:
            return np.ones(self.length, dtype=bool)
        first = self.validity_bit_offset // 8
        nbytes = (self.validity_bit_offset % 8 + self.length + 7) // 8
        raw = _copy_dtoh(self.ctx, self.validity.ptr + first, nbytes)
        return unpack_bits(raw, self.validity_bit_offset % 8, self.length)

    def values_numpy(self):
        if self.length == 0:
            if self.data_type.physical in (L.AH_UTF8, L.AH_LARGE_UTF8):
                return []
            return np.empty(0, dtype=self.data_type.np_dtype)
        p = self.data_type.physical
        if p == L.AH_BOOL:
            first = self.values_bit_offset // 8
            nbytes = (self.values_bit_offset % 8 + self.length + 7) // 8
            raw = _copy_dtoh(self.ctx, self.values.ptr + first, nbytes)
            return unpack_bits(raw, self.values_bit_offset % 8, self.length)
        if p in (L.AH_UTF8, L.AH_LARGE_UTF8):
            odt = np.int32 if p == L.AH_UTF8 else np.int64
            offs = _copy_dtoh(self.ctx, self.offsets.ptr, (self.length + 1) * np.dtype(odt).itemsize).view(odt)
            base, total = int(offs[0]), int(offs[-1])
            data = _copy_dtoh(self.ctx, self.values.ptr + base, total - base).tobytes() if total > base else b""
            return [data[offs[i] - base:offs[i + 1] - base].decode() for i in range(self.length)]
        w = self.data_type.width
        raw = _copy_dtoh(self.ctx, self.values.ptr, self.length * w)
        if p in (L.AH_UTF8_VIEW, L.AH_BINARY_VIEW):  # value_unchecked (byte_view_array.rs:337-352)
            views = raw.view(_VIEW)
            bufs = [b.to_numpy().tobytes() for b in (self.data_buffers or [])]
            out = []
            for i in range(self.length):
                n = int(views["length"][i])
                if n <= MAX_INLINE_VIEW_LEN:
                    b = raw[i * 16 + 4:i * 16 + 4 + n].tobytes()
                else:
                    o = int(views["offset"][i])
                    b = bufs[int(views["buffer_index"][i])][o:o + n]
                out.append(b.decode() if p == L.AH_UTF8_VIEW else b)
            return out
        return raw.view(self.data_type.np_dtype)

    def to_pylist(self):
        vals = self.values_numpy()
        mask = self.valid_mask()
        out = []
        for i in range(self.length):
            if not mask[i]:
                out.append(None)
            else:
                v = vals[i]
                out.append(v.item() if hasattr(v, "item") and self.data_type.np_dtype is not _DEC128 else v)
        return out

    def __repr__(self):
        return f"Array<{self.data_type}>(len={self.length}, nulls={self._null_count})"


def _copy_dtoh(ctx, ptr, nbytes):
    out = np.empty(nbytes, dtype=np.uint8)
    if nbytes:
        ctx.check(ctx.lib.ah_memcpy_dtoh(ctx.handle, out.ctypes.data, ptr, nbytes))
    return out


class Scalar:
    """``Scalar<T>``: a length-1 array used as a broadcast Datum
    (arrow-array/src/scalar.rs:128-152)."""

    def __init__(self, array):
        assert array.length == 1
        self.array = array

    @classmethod
    def new(cls, value, data_type, ctx=None):
        """``PrimitiveArray::new_scalar(v)`` (primitive_array.rs:694-700); None -> null scalar."""
        return cls(Array.from_pylist([value], data_type, ctx))

    def get(self):
        return self.array, True


class RecordBatch:
    """Minimal RecordBatch (arrow-array/src/record_batch.rs:224): named columns of equal length."""

    def __init__(self, names, columns, num_rows=None):
        self.names = list(names)
        self.columns = list(columns)
        self._num_rows = num_rows if num_rows is not None else (columns[0].length if columns else 0)

    def num_rows(self):
        return self._num_rows

    def num_columns(self):
        return len(self.columns)

    def column(self, i):
        return self.columns[i]
