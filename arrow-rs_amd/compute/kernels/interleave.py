"""arrow::compute::kernels::interleave == arrow_select::interleave (arrow-select/src/interleave.rs)."""
import ctypes as C

import numpy as np

from ... import _lib as L
from ...array import Array, InvalidArgumentError, RecordBatch, UInt32


def _index_arrays(ctx, indices):
    if isinstance(indices, tuple) and len(indices) == 2 and isinstance(indices[0], Array):
        return indices  # already (array_index, row_index) on the device
    pairs = np.asarray(list(indices), dtype=np.uint32).reshape(-1, 2)
    return (Array.from_numpy(np.ascontiguousarray(pairs[:, 0]), data_type=UInt32, ctx=ctx),
            Array.from_numpy(np.ascontiguousarray(pairs[:, 1]), data_type=UInt32, ctx=ctx))


def interleave(values, indices):
    """``pub fn interleave(values: &[&dyn Array], indices: &[(usize, usize)])`` (interleave.rs:74).  ``indices`` is a
    host list of (array, row) pairs, or a pair of device UInt32 arrays."""
    if not values:
        raise InvalidArgumentError("interleave requires input of at least one array")
    ctx = values[0].ctx
    ai, ri = _index_arrays(ctx, indices)
    views = (L.ArrayView * len(values))()
    for i, a in enumerate(values):
        views[i] = a.view()
    av, rv = ai.view(), ri.view()
    out = L.ArrayOut()
    ctx.check(ctx.lib.ah_interleave(ctx.handle, len(values), views, C.byref(av), C.byref(rv), C.byref(out)))
    return Array._from_out(ctx, out, values[0].data_type)


def interleave_record_batch(record_batches, indices):
    """interleave.rs:912"""
    ctx = record_batches[0].columns[0].ctx
    idx = _index_arrays(ctx, indices)
    cols = [interleave([rb.columns[i] for rb in record_batches], idx) for i in range(record_batches[0].num_columns())]
    return RecordBatch(record_batches[0].names, cols, num_rows=idx[0].length)
