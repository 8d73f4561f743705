"""arrow::compute::kernels::concat == arrow_select::concat (arrow-select/src/concat.rs:495),
primitive / boolean arms (:334-343).  Also the multi-GPU reassembly primitive."""
import ctypes as C

from ... import _lib as L
from ...array import Array, InvalidArgumentError


def concat(arrays):
    if len(arrays) == 0:
        raise InvalidArgumentError("concat requires input of at least one array")
    ctx = arrays[0].ctx
    views = (L.ArrayView * len(arrays))()
    for i, a in enumerate(arrays):
        views[i] = a.view()
    out = L.ArrayOut()
    ctx.check(ctx.lib.ah_concat(ctx.handle, len(arrays), views, C.byref(out)))
    return Array._from_out(ctx, out, arrays[0].data_type)
