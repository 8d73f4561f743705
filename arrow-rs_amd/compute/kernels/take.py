"""arrow::compute::kernels::take == arrow_select::take (arrow-select/src/take.rs)."""
import ctypes as C
from dataclasses import dataclass

from ... import _lib as L
from ...array import Array, RecordBatch


@dataclass
class TakeOptions:
    """``TakeOptions { check_bounds }`` (take.rs:388-394); default false."""
    check_bounds: bool = False


def take(values, indices, options=None):
    """``pub fn take(values: &dyn Array, indices: &dyn Array, options: Option<TakeOptions>)``
    (take.rs:89).  Out-of-bounds without check_bounds raises ``Panic`` with the
    reference's panic text; with check_bounds raises ``ComputeError``."""
    options = options or TakeOptions()
    ctx = values.ctx
    out = L.ArrayOut()
    vv, iv = values.view(), indices.view()
    ctx.check(ctx.lib.ah_take(ctx.handle, C.byref(vv), C.byref(iv), 1 if options.check_bounds else 0,
                              C.byref(out)))
    return Array._from_out(ctx, out, values.data_type, keepalive=(values,) if values.data_buffers is not None else ())


def take_arrays(arrays, indices, options=None):
    """``take_arrays`` (take.rs:155)."""
    return [take(a, indices, options) for a in arrays]


def take_record_batch(record_batch, indices):
    """``take_record_batch`` (take.rs:1124)."""
    cols = [take(c, indices, None) for c in record_batch.columns]
    return RecordBatch(record_batch.names, cols, num_rows=indices.length)
