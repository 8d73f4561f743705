"""arrow::compute::kernels::filter == arrow_select::filter
(arrow-select/src/filter.rs; re-export arrow/src/compute/kernels.rs:25-27).

Same names and argument order as the reference; the work happens in HBM through
``ah_filter*`` (include/arrow_hip.h)."""
import ctypes as C

from ... import _lib as L
from ...array import Array, Boolean, RecordBatch, InvalidArgumentError


_CMP_OPS = {"eq": 0, "neq": 1, "lt": 2, "lt_eq": 3, "gt": 4, "gt_eq": 5}
_JOIN_OPS = {"and": 0, "or": 1, "and_kleene": 3, "or_kleene": 4}


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

    @classmethod
    def from_terms(cls, terms, joins=()):
        """The lazy predicate (``ah_filter_predicate_build_expr``): ``terms`` = [(op, lhs, rhs), ...] with ``op`` one of
        "eq" "neq" "lt" "lt_eq" "gt" "gt_eq" and lhs / rhs Datums (Array or Scalar); ``joins`` = "and" | "or" |
        "and_kleene" | "or_kleene" between consecutive terms, folded left to right.  Result-identical to
        ``FilterBuilder::new(&and_kleene(&lt(..)?, &gt_eq(..)?)?)`` (cmp.rs:113-164, boolean.rs:60-300, filter.rs:256-273)
        with the comparisons evaluated inside the filter's count pass instead of being materialised."""
        self = cls.__new__(cls)
        self._predicate = None
        self._terms = [(_CMP_OPS[op] if isinstance(op, str) else int(op), lhs, rhs) for op, lhs, rhs in terms]
        self._joins = [_JOIN_OPS[j] if isinstance(j, str) else int(j) for j in joins]
        if len(self._joins) != max(len(self._terms) - 1, 0):
            raise InvalidArgumentError("a filter expression of n terms takes n - 1 joins")
        return self

    def _build_expr(self):
        n = len(self._terms)
        arr = (L.FilterTerm * max(n, 1))()
        keep = []
        ctx = None
        for i, (op, lhs, rhs) in enumerate(self._terms):
            l, l_s = lhs.get()
            r, r_s = rhs.get()
            ctx = ctx or l.ctx
            lv, rv = l.view(), r.view()
            keep += [l, r, lv, rv]
            arr[i].op, arr[i].lhs, arr[i].lhs_is_scalar = op, C.pointer(lv), int(l_s)
            arr[i].rhs, arr[i].rhs_is_scalar = C.pointer(rv), int(r_s)
        joins = (C.c_int32 * max(len(self._joins), 1))(*self._joins)
        h = C.c_void_p()
        ctx.check(ctx.lib.ah_filter_predicate_build_expr(ctx.handle, n, arr, joins, C.byref(h)))
        return FilterPredicate(ctx, h, None)  # the operand buffers are only read during the call

    def build(self):
        if self._predicate is None:
            return self._build_expr()
        ctx = self._predicate.ctx
        h = C.c_void_p()
        pv = self._predicate.view()
        ctx.check(ctx.lib.ah_filter_predicate_build(ctx.handle, C.byref(pv), C.byref(h)))
        return FilterPredicate(ctx, h, self._predicate)


def _term_array(terms, joins):
    ops = [(_CMP_OPS[op] if isinstance(op, str) else int(op), lhs, rhs) for op, lhs, rhs in terms]
    js = [_JOIN_OPS[j] if isinstance(j, str) else int(j) for j in joins]
    if len(js) != max(len(ops) - 1, 0):
        raise InvalidArgumentError("a filter expression of n terms takes n - 1 joins")
    arr = (L.FilterTerm * max(len(ops), 1))()
    keep, ctx = [], None
    for i, (op, lhs, rhs) in enumerate(ops):
        l, l_s = lhs.get()
        r, r_s = rhs.get()
        ctx = ctx or l.ctx
        lv, rv = l.view(), r.view()
        keep += [l, r, lv, rv]
        arr[i].op, arr[i].lhs, arr[i].lhs_is_scalar = op, C.pointer(lv), int(l_s)
        arr[i].rhs, arr[i].rhs_is_scalar = C.pointer(rv), int(r_s)
    return ctx, len(ops), arr, (C.c_int32 * max(len(js), 1))(*js), keep


def filter_expr(values, terms, joins=()):
    """``filter(values, <terms joined by joins>)`` in one call (``ah_filter_expr``): what an engine writes as
    ``filter(&a, &and_kleene(&lt(&a, &x)?, &gt_eq(&b, &y)?)?)`` (cmp.rs:113-164, boolean.rs:60-300, filter.rs:201) with
    nothing materialised: the comparisons run inside the filter's count pass, the scatter reads the packed result."""
    ctx, n, arr, js, _keep = _term_array(terms, joins)
    out = L.ArrayOut()
    vv = values.view()
    ctx.check(ctx.lib.ah_filter_expr(ctx.handle, n, arr, js, C.byref(vv), C.byref(out)))
    return Array._from_out(ctx, out, values.data_type, keepalive=(values,))


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
