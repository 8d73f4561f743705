"""arrow::compute::kernels::concat == arrow_select::concat (arrow-select/src/concat.rs:495),
primitive / boolean arms (:334-343).  Also the multi-GPU reassembly primitive."""
import ctypes as C

from ... import _lib as L
from ...array import Array, InvalidArgumentError, RecordBatch


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


def concat_batches(schema, input_batches):
    """``concat_batches(schema, batches)`` (arrow-select/src/concat.rs:607): the i-th column of the result is the
    concatenation of the batches' i-th columns, named by ``schema`` (a list of field names here; the batches' own
    names are not consulted, as in the reference).  An empty schema sums the row counts (:612-617); no input
    batches give an empty batch (:620-622; columns are left out because a name list carries no types)."""
    names = list(schema)
    batches = list(input_batches)
    if not names:
        return RecordBatch([], [], sum(b.num_rows() for b in batches))
    if not batches:
        return RecordBatch(names, [], 0)
    return RecordBatch(names, [concat([b.column(i) for b in batches]) for i in range(len(names))])
