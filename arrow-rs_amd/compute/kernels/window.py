"""arrow::compute::kernels::window == arrow_select::window (arrow-select/src/window.rs): ``shift``."""
import ctypes as C

from ... import _lib as L
from ...array import Array


def shift(array, offset):
    """window.rs:56 — a positive ``offset`` shifts right, a negative one left; vacated slots are null; offset 0
    shares the input's buffers; ``abs(offset) >= len`` (or i64::MIN) gives an all-null array."""
    ctx = array.ctx
    out = L.ArrayOut()
    v = array.view()
    ctx.check(ctx.lib.ah_shift(ctx.handle, C.byref(v), int(offset), C.byref(out)))
    return Array._from_out(ctx, out, array.data_type, keepalive=(array,))
