"""arrow::compute::kernels::cast == arrow_cast::cast (arrow-cast/src/cast/mod.rs:347,:790),
restricted to the hot path: numeric<->numeric and numeric->Utf8/LargeUtf8."""
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
    return bool(L.load().ah_can_cast_types(from_type.physical, to_type.physical))


def cast_with_options(array, to_type, cast_options):
    """cast/mod.rs:790"""
    ctx = array.ctx
    out = L.ArrayOut()
    v = array.view()
    ctx.check(ctx.lib.ah_cast(ctx.handle, C.byref(v), to_type.physical, int(cast_options.safe),
                              C.byref(out)))
    return Array._from_out(ctx, out, to_type)


def cast(array, to_type):
    """cast/mod.rs:347 — ``cast_with_options(array, to_type, &CastOptions::default())``"""
    return cast_with_options(array, to_type, CastOptions())
