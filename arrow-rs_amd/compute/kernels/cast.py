"""arrow::compute::kernels::cast == arrow_cast::cast (arrow-cast/src/cast/mod.rs:347,:790),
restricted to the hot path: numeric <-> numeric, numeric <-> Utf8 / LargeUtf8, Boolean <-> numeric, and the temporal
arms (Date32 / Date64 / Time32 / Time64 / Timestamp / Duration among themselves and to / from the numbers,
cast/mod.rs:1700-2260), which go through ``ah_cast_with_types`` because their arithmetic depends on the logical type."""
import ctypes as C
from dataclasses import dataclass

from ... import _lib as L
from ...array import Array


@dataclass
class CastOptions:
    """``CastOptions { safe, .. }`` (cast/mod.rs:95-111); default safe=true."""
    safe: bool = True


def can_cast_types(from_type, to_type):
    """``can_cast_types`` (cast/mod.rs:115), hot-path subset."""
    if from_type.logical is None and to_type.logical is None:
        return bool(L.load().ah_can_cast_types(from_type.physical, to_type.physical))
    f, t = from_type.descriptor(), to_type.descriptor()
    return bool(L.load().ah_can_cast_data_types(C.byref(f), C.byref(t)))


def cast_with_options(array, to_type, cast_options):
    """cast/mod.rs:790"""
    ctx = array.ctx
    out = L.ArrayOut()
    v = array.view()
    if array.data_type.logical is None and to_type.logical is None:
        ctx.check(ctx.lib.ah_cast(ctx.handle, C.byref(v), to_type.physical, int(cast_options.safe),
                                  C.byref(out)))
    else:
        f, t = array.data_type.descriptor(), to_type.descriptor()  # a named zone raises the reference's ParseError
        ctx.check(ctx.lib.ah_cast_with_types(ctx.handle, C.byref(v), C.byref(f), C.byref(t),
                                             int(cast_options.safe), C.byref(out)))
    return Array._from_out(ctx, out, to_type)


def cast(array, to_type):
    """cast/mod.rs:347 — ``cast_with_options(array, to_type, &CastOptions::default())``"""
    return cast_with_options(array, to_type, CastOptions())


def cast_chain(array, to_types, cast_options=None):
    """``cast(cast(array, to_types[0]), to_types[1]) ...`` as ONE call (``ah_cast_chain``): byte-identical results; the
    library decides what to materialise in between — Int64 -> Float64 -> Utf8 / LargeUtf8 (BASELINE configs[3]) never builds
    the Float64 array.  Plain (non-logical) types only; anything else: chain ``cast`` yourself."""
    opts = cast_options or CastOptions()
    if array.data_type.logical is not None or any(t.logical is not None for t in to_types):
        out = array
        for t in to_types:
            out = cast_with_options(out, t, opts)
        return out
    ctx = array.ctx
    out = L.ArrayOut()
    v = array.view()
    types = (C.c_int32 * len(to_types))(*[t.physical for t in to_types])
    ctx.check(ctx.lib.ah_cast_chain(ctx.handle, C.byref(v), len(to_types), types, int(opts.safe), C.byref(out)))
    return Array._from_out(ctx, out, to_types[-1])
