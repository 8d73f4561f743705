"""arrow::compute::kernels::filter == arrow_select::filter
(arrow-select/src/filter.rs; re-export arrow/src/compute/kernels.rs:25-27).

Same names and argument order as the reference; the work happens in HBM through
``ah_filter*`` (include/arrow_hip.h)."""
import ctypes as C

from ... import _lib as L
from ...array import Array, Boolean, RecordBatch, InvalidArgumentError


def _check_predicate(predicate):
    if predicate.data_type != Boolean:
        raise InvalidArgumentError(f"filter predicate must be Boolean, got {predicate.data_type}")


def filter(values, predicate):
    """``pub fn filter(values: &dyn Array, predicate: &BooleanArray) -> Result<ArrayRef, ArrowError>``
    (filter.rs:201)."""
    _check_predicate(predicate)
    ctx = values.ctx
    out = L.ArrayOut()
    vv, pv = values.view(), predicate.view()
    ctx.check(ctx.lib.ah_filter(ctx.handle, C.byref(vv), C.byref(pv), C.byref(out)))
    return Array._from_out(ctx, out, values.data_type, keepalive=(values,))


def prep_null_mask_filter(filter_array):
    """``pub fn prep_null_mask_filter(filter: &BooleanArray) -> BooleanArray`` (filter.rs:167):
    nulls become false.  Device-side this is ``values AND validity``."""
    from .boolean import _and_validity
    return _and_validity(filter_array)


class FilterPredicate:
    """``FilterPredicate`` (filter.rs:442-533): count once, apply to many arrays."""

    def __init__(self, ctx, handle, predicate):
        self.ctx = ctx
        self._h = handle
        self._predicate = predicate  # keeps the predicate's device buffers alive
        import weakref
        self._fin = weakref.finalize(self, ctx.lib.ah_filter_predicate_free, ctx.handle, handle)

    def count(self):
        """``FilterPredicate::count`` (filter.rs:481)."""
        return self.ctx.lib.ah_filter_predicate_count(self._h)

    def filter(self, values):
        """``FilterPredicate::filter`` (filter.rs:451)."""
        out = L.ArrayOut()
        vv = values.view()
        self.ctx.check(self.ctx.lib.ah_filter_predicate_apply(self.ctx.handle, self._h, C.byref(vv),
                                                              C.byref(out)))
        return Array._from_out(self.ctx, out, values.data_type, keepalive=(values,))

    def filter_record_batch(self, record_batch):
        """``FilterPredicate::filter_record_batch`` (filter.rs:459-478)."""
        cols = [self.filter(c) for c in record_batch.columns]
        return RecordBatch(record_batch.names, cols, num_rows=self.count())


class FilterBuilder:
    """``FilterBuilder`` (filter.rs:248-324).  ``optimize()`` is a no-op marker: the
    device predicate always carries its prefix tables (the analogue of
    IterationStrategy::Indices)."""

    def __init__(self, predicate):
        _check_predicate(predicate)
        self._predicate = predicate

    @classmethod
    def new(cls, predicate):
        return cls(predicate)

    def optimize(self):
        return self

    @staticmethod
    def is_optimize_beneficial(data_type):
        return False  # only Struct / sparse Union in the reference (filter.rs:304-314)

    def build(self):
        ctx = self._predicate.ctx
        h = C.c_void_p()
        pv = self._predicate.view()
        ctx.check(ctx.lib.ah_filter_predicate_build(ctx.handle, C.byref(pv), C.byref(h)))
        return FilterPredicate(ctx, h, self._predicate)


def filter_record_batch(record_batch, predicate):
    """``pub fn filter_record_batch(record_batch: &RecordBatch, predicate: &BooleanArray)``
    (filter.rs:225): one predicate pass, then one scatter per column."""
    _check_predicate(predicate)
    ctx = predicate.ctx
    n = record_batch.num_columns()
    views = (L.ArrayView * max(n, 1))()
    for i, c in enumerate(record_batch.columns):
        views[i] = c.view()
    outs = (L.ArrayOut * max(n, 1))()
    rows = C.c_int64()
    pv = predicate.view()
    ctx.check(ctx.lib.ah_filter_record_batch(ctx.handle, n, views, C.byref(pv), outs, C.byref(rows)))
    cols = []
    for i, c in enumerate(record_batch.columns):
        o = L.ArrayOut()
        C.memmove(C.byref(o), C.byref(outs[i]), C.sizeof(L.ArrayOut))
        cols.append(Array._from_out(ctx, o, c.data_type, keepalive=(c,)))
    return RecordBatch(record_batch.names, cols, num_rows=rows.value)
