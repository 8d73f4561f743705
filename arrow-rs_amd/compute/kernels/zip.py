"""arrow::compute::kernels::zip == arrow_select::zip (arrow-select/src/zip.rs)."""
import ctypes as C

from ... import _lib as L
from ...array import Array


def zip(mask, truthy, falsy):  # noqa: A001
    """``pub fn zip(mask: &BooleanArray, truthy: &dyn Datum, falsy: &dyn Datum)`` (zip.rs:99): rows of ``truthy`` where
    the mask is true, of ``falsy`` where it is false or null; either side may be a ``Scalar``."""
    t_arr, t_s = truthy.get()  # Datum::get (arrow-array/src/scalar.rs:78-98)
    f_arr, f_s = falsy.get()
    ctx = mask.ctx
    out = L.ArrayOut()
    mv, tv, fv = mask.view(), t_arr.view(), f_arr.view()
    ctx.check(ctx.lib.ah_zip(ctx.handle, C.byref(mv), C.byref(tv), int(t_s), C.byref(fv), int(f_s), C.byref(out)))
    return Array._from_out(ctx, out, t_arr.data_type)
