"""Arrow C Data Interface for device arrays — the mirror of ``arrow::ffi::{from_ffi, to_ffi}``
(arrow-array/src/ffi.rs:231-254) and of the pyarrow bridge built on it
(arrow-pyarrow/src/lib.rs:199-257: ``_export_to_c`` / ``_import_from_c``).

A host-resident C-Data array is copied into HBM by ``ah_import_c_data``; results are copied
back into host C-Data structs by ``ah_export_c_data`` whose release callbacks free the host
copies.  The logical type (timestamp unit/zone, decimal precision, ...) never reaches the
kernels: it is carried here as the schema format string and handed back on export, which is how
the reference preserves ``data_type`` through filter/take (filter.rs:783-787).
"""
import ctypes as C

import numpy as np

from . import _lib as L
from . import array as A
from ._lib import FFI_ArrowArray, FFI_ArrowSchema

# format strings of the logical types array.py names (arrow-schema/src/ffi.rs:779-856)
_NAMED = {
    "b": A.Boolean, "c": A.Int8, "s": A.Int16, "i": A.Int32, "l": A.Int64,
    "C": A.UInt8, "S": A.UInt16, "I": A.UInt32, "L": A.UInt64,
    "e": A.Float16, "f": A.Float32, "g": A.Float64, "u": A.Utf8, "U": A.LargeUtf8,
    "tdD": A.Date32, "tdm": A.Date64, "tts": A.Time32Second, "ttm": A.Time32Millisecond,
    "ttu": A.Time64Microsecond, "ttn": A.Time64Nanosecond,
    "tDs": A.DurationSecond, "tDm": A.DurationMillisecond, "tDu": A.DurationMicrosecond,
    "tDn": A.DurationNanosecond,
    "tss:": A.TimestampSecond, "tsm:": A.TimestampMillisecond, "tsu:": A.TimestampMicrosecond,
    "tsn:": A.TimestampNanosecond,
}
_FORMAT_OF = {dt.name: fmt for fmt, dt in _NAMED.items()}
_NP_OF_PHYSICAL = {
    L.AH_BOOL: np.bool_, L.AH_INT8: np.int8, L.AH_INT16: np.int16, L.AH_INT32: np.int32, L.AH_INT64: np.int64,
    L.AH_UINT8: np.uint8, L.AH_UINT16: np.uint16, L.AH_UINT32: np.uint32, L.AH_UINT64: np.uint64,
    L.AH_FLOAT16: np.float16, L.AH_FLOAT32: np.float32, L.AH_FLOAT64: np.float64,
    L.AH_FIXED16: A._DEC128, L.AH_FIXED32: A._DEC256,
    L.AH_UTF8: None, L.AH_LARGE_UTF8: None,
}


def data_type_from_format(ctx, fmt):
    """``DataType::try_from(&FFI_ArrowSchema)`` (arrow-schema/src/ffi.rs:492): a named logical
    type where array.py has one, otherwise an opaque logical type that remembers its format."""
    if fmt in _NAMED:
        return _NAMED[fmt]
    if fmt[:3] in ("tss", "tsm", "tsu", "tsn") and fmt[3:4] == ":":
        # Timestamp with a zone (arrow-schema/src/ffi.rs:640-664).  Fixed offsets become a typed Timestamp the cast
        # kernels understand; a named zone stays opaque (layout and format string survive, casts refuse it).
        try:
            A.parse_fixed_offset(fmt[4:])
            dt = A.Timestamp("smun".index(fmt[2]), fmt[4:])
            dt.format = fmt
            return dt
        except A.ParseError:
            pass
    phys = C.c_int32(0)
    ctx.check(ctx.lib.ah_type_from_format(ctx.handle, fmt.encode(), C.byref(phys)))
    if fmt.startswith("d:"):
        # Decimal (arrow-schema/src/ffi.rs:600-636): "d:p,s" / "d:p,s,128" is the typed Decimal128 the arithmetic and
        # cast kernels understand (it carries the (AH_DT_DECIMAL128, precision, scale) descriptor); 256 bits is the
        # 32-byte native the selection kernels move verbatim
        parts = [x.strip() for x in fmt[2:].split(",")]
        bits = parts[2] if len(parts) > 2 else "128"
        if bits == "128":
            dt = A.Decimal128(int(parts[0]), int(parts[1]))
            dt.format = fmt
            return dt
        if bits == "256":
            dt = A.Decimal256(int(parts[0]), int(parts[1]))
            dt.format = fmt
            return dt
        name = f"Decimal{bits}({parts[0]}, {parts[1]})"
    else:
        name = f"ffi<{fmt}>"
    dt = A.DataType(name, phys.value, _NP_OF_PHYSICAL[phys.value])
    dt.format = fmt
    return dt


def format_of(data_type):
    fmt = getattr(data_type, "format", None) or _FORMAT_OF.get(data_type.name)
    if fmt is None and data_type.name.startswith("Decimal128("):
        p, s = data_type.name[len("Decimal128("):-1].split(",")
        fmt = f"d:{p.strip()},{s.strip()}"
    if fmt is None and data_type.name.startswith("Decimal256("):
        p, s = data_type.name[len("Decimal256("):-1].split(",")
        fmt = f"d:{p.strip()},{s.strip()},256"
    return fmt  # None: the library's default for the physical type


def from_ffi(array, schema, ctx=None):
    """Device Array from host C-Data structs (borrowed; the caller still releases them)."""
    ctx = ctx or A.default_context()
    fmt = schema.format.decode() if schema.format else None
    if fmt is None:
        raise A.CDataInterfaceError("Null pointer passed where a format string was expected")
    out = L.ArrayOut()
    ctx.check(ctx.lib.ah_import_c_data(ctx.handle, C.byref(array), C.byref(schema), C.byref(out)))
    return A.Array._from_out(ctx, out, data_type_from_format(ctx, fmt))


class _Exported:
    """Owns the (FFI_ArrowArray, FFI_ArrowSchema) pair until a consumer moves it out."""

    def __init__(self):
        self.array, self.schema = FFI_ArrowArray(), FFI_ArrowSchema()

    def release(self):
        for st in (self.array, self.schema):
            if st.release:
                C.CFUNCTYPE(None, C.c_void_p)(st.release)(C.addressof(st))

    def __del__(self):
        self.release()


def to_ffi(array):
    """Host C-Data structs holding a copy of the device array (``to_ffi``, ffi.rs:231)."""
    ctx = array.ctx
    ex = _Exported()
    fmt = format_of(array.data_type)
    v = array.view()
    ctx.check(ctx.lib.ah_export_c_data(ctx.handle, C.byref(v), fmt.encode() if fmt else None,
                                       C.byref(ex.array), C.byref(ex.schema)))
    return ex


# ---- pyarrow bridge (arrow-pyarrow/src/lib.rs:199-257) -------------------------------------
def from_pyarrow(obj, ctx=None):
    """pyarrow.Array -> device Array; pyarrow.RecordBatch -> device RecordBatch."""
    import pyarrow as pa
    ctx = ctx or A.default_context()
    if isinstance(obj, pa.RecordBatch):
        return A.RecordBatch(list(obj.schema.names), [from_pyarrow(c, ctx) for c in obj.columns], obj.num_rows)
    if isinstance(obj, pa.ChunkedArray):
        obj = obj.combine_chunks()
    arr, sch = FFI_ArrowArray(), FFI_ArrowSchema()
    obj._export_to_c(C.addressof(arr), C.addressof(sch))
    try:
        return from_ffi(arr, sch, ctx)
    finally:
        for st in (arr, sch):
            if st.release:
                C.CFUNCTYPE(None, C.c_void_p)(st.release)(C.addressof(st))


def to_pyarrow(obj):
    """device Array / RecordBatch -> pyarrow (which takes ownership of the exported structs)."""
    import pyarrow as pa
    if isinstance(obj, A.RecordBatch):
        return pa.RecordBatch.from_arrays([to_pyarrow(c) for c in obj.columns], names=list(obj.names))
    ex = to_ffi(obj)
    return pa.Array._import_from_c(C.addressof(ex.array), C.addressof(ex.schema))


# ---- Arrow C Device Data Interface (zero-copy, ROCm) ------------------------------------------
class _ExportedDevice:
    """Owns an (ArrowDeviceArray, ArrowSchema) pair; `keepalive` pins a borrowed source array."""

    def __init__(self, keepalive=None):
        self.array, self.schema = L.FFI_ArrowDeviceArray(), FFI_ArrowSchema()
        self.keepalive = keepalive

    def release(self):
        if self.array.array.release:
            C.CFUNCTYPE(None, C.c_void_p)(self.array.array.release)(C.addressof(self.array.array))
        if self.schema.release:
            C.CFUNCTYPE(None, C.c_void_p)(self.schema.release)(C.addressof(self.schema))
        self.keepalive = None

    def __del__(self):
        self.release()


def to_device_ffi(array):
    """Zero-copy export as `struct ArrowDeviceArray` (device_type ARROW_DEVICE_ROCM): the exported struct borrows
    the array's HBM buffers (kept alive by the returned holder)."""
    ctx = array.ctx
    ex = _ExportedDevice(keepalive=array)
    fmt = format_of(array.data_type)
    v = array.view()
    ctx.check(ctx.lib.ah_export_c_device_data(ctx.handle, C.byref(v), None, fmt.encode() if fmt else None,
                                              C.byref(ex.array), C.byref(ex.schema)))
    return ex


def from_device_ffi(dev_array, schema, ctx=None, keepalive=None):
    """Zero-copy import of a ROCm-resident `ArrowDeviceArray`: an Array viewing the producer's buffers
    (`keepalive` = whatever owns them)."""
    ctx = ctx or A.default_context()
    view = L.ArrayView()
    ctx.check(ctx.lib.ah_import_c_device_data(ctx.handle, C.byref(dev_array), C.byref(schema), C.byref(view)))
    dt = data_type_from_format(ctx, schema.format.decode())
    n = view.length
    w = dt.width
    mem = lambda p, nb: A._RawMem(p, nb, keepalive) if p else None  # noqa: E731
    vals = mem(view.values, (n * w) if w > 0 else 0)
    nulls = view.null_count
    arr = A.Array(ctx, dt, n, vals, view.values_bit_offset, mem(view.validity, (view.validity_bit_offset + n + 7) // 8),
                  view.validity_bit_offset, max(nulls, 0), mem(view.offsets, 0))
    if view.validity and nulls < 0:  # "not yet computed": count on the device
        cnt = C.c_int64()
        ctx.check(ctx.lib.ah_count_set_bits(ctx.handle, view.validity, view.validity_bit_offset, n, C.byref(cnt)))
        arr._null_count = n - cnt.value
    return arr
