"""arrow::compute::kernels::rank == arrow_ord::rank (arrow-ord/src/rank.rs)."""
import ctypes as C

from ... import _lib as L
from ...array import Array, UInt32
from .sort import SortOptions


def rank_array(array, options=None):
    """The ranks as a device UInt32 array (never null)."""
    options = options or SortOptions()
    ctx = array.ctx
    out = L.ArrayOut()
    v = array.view()
    ctx.check(ctx.lib.ah_rank(ctx.handle, C.byref(v), int(options.descending), int(options.nulls_first), C.byref(out)))
    return Array._from_out(ctx, out, UInt32)


def rank(array, options=None):
    """rank.rs:58 — ``Vec<u32>``: 1-based position in the sorted order, ties share the highest of their positions,
    nulls share one rank before (nulls_first, the default) or after all values."""
    r = rank_array(array, options)
    return r.values_numpy()
