"""parquet::arrow::arrow_reader::{RowSelection, RowSelector} with the selection held as a device bitmap —
the mirror of parquet/src/arrow/arrow_reader/selection/mod.rs (bitmap-backed form, `RowSelectionInner::Mask`).

The parquet reader's row-filter loop (arrow_reader/read_plan.rs) evaluates a predicate per batch, turns the
BooleanArrays into a selection (`from_filters`) and chains it onto the previous one (`and_then`): with the
predicate results already in HBM, the selection algebra stays there too.  The run-length form
(`Vec<RowSelector>`) is produced on demand (`selectors()` / `iter()`), as the reference does for mask-backed
selections (boolean.rs:172-190)."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L
from . import array as A
from .compute.kernels.concat import concat


@dataclass(frozen=True)
class RowSelector:
    """selection/selector.rs:33-57"""
    row_count: int
    skip: bool

    @classmethod
    def select(cls, row_count):
        return cls(int(row_count), False)

    @classmethod
    def skip_rows(cls, row_count):  # `RowSelector::skip` (the field of the same name shadows it in Python)
        return cls(int(row_count), True)


def _out_bool(ctx, out, keepalive=()):
    return A.Array._from_out(ctx, out, A.Boolean, keepalive=keepalive)


class RowSelection:
    def __init__(self, mask):
        if mask.data_type != A.Boolean:
            raise A.InvalidArgumentError("a row selection mask must be a BooleanArray")
        if mask.null_count() != 0:  # `assert_eq!(filter.null_count(), 0)` (mod.rs:318)
            raise A.Panic(f"assertion `left == right` failed\n  left: {mask.null_count()}\n right: 0")
        self.mask = mask
        self._count = None

    # ---- constructors
    @classmethod
    def from_boolean_buffer(cls, mask):
        """mod.rs:210"""
        return cls(mask)

    @classmethod
    def from_filters(cls, filters, ctx=None):
        """mod.rs:311: the concatenation of the filters' value bits."""
        filters = list(filters)
        for f in filters:
            if f.null_count() != 0:
                raise A.Panic(f"assertion `left == right` failed\n  left: {f.null_count()}\n right: 0")
        if not filters:
            return cls._empty(ctx or A.default_context())
        return cls(filters[0] if len(filters) == 1 else concat(filters))

    @classmethod
    def _empty(cls, ctx):
        return cls.from_boundaries([], 0, ctx)

    @classmethod
    def from_boundaries(cls, bounds, total_rows, ctx=None):
        ctx = ctx or A.default_context()
        b = A.Array.from_numpy(np.asarray(bounds, dtype=np.int64), ctx=ctx)
        out = L.ArrayOut()
        v = b.view()
        ctx.check(ctx.lib.ah_selection_from_boundaries(ctx.handle, C.byref(v), int(total_rows), C.byref(out)))
        return cls(_out_bool(ctx, out))

    @classmethod
    def from_selectors(cls, selectors, ctx=None):
        """`From<Vec<RowSelector>>` (mod.rs:671-711): empty selectors dropped, neighbours of one kind merged."""
        bounds, pos, state = [], 0, False
        for s in selectors:
            if s.row_count == 0:
                continue
            want = not s.skip
            if want != state:
                bounds.append(pos)
                state = want
            pos += s.row_count
        return cls.from_boundaries(bounds, pos, ctx)

    @classmethod
    def from_consecutive_ranges(cls, ranges, total_rows, ctx=None):
        """mod.rs:326-356"""
        sel, last_end = [], 0
        for start, end in ranges:
            if end - start == 0:
                continue
            if start < last_end:
                raise A.Panic("out of order")
            if start > last_end:
                sel.append(RowSelector.skip_rows(start - last_end))
            sel.append(RowSelector.select(end - start))
            last_end = end
        if last_end != total_rows:
            sel.append(RowSelector.skip_rows(total_rows - last_end))
        return cls.from_selectors(sel, ctx)

    # ---- inspection
    @property
    def ctx(self):
        return self.mask.ctx

    def as_mask(self):
        """mod.rs:229"""
        return self.mask

    def total_row_count(self):
        return self.mask.length

    def row_count(self):
        if self._count is None:
            n = C.c_int64()
            m = self.mask
            if m.length:
                self.ctx.check(self.ctx.lib.ah_count_set_bits(self.ctx.handle, m.values.ptr, m.values_bit_offset, m.length,
                                                              C.byref(n)))
            self._count = n.value
        return self._count

    def skipped_row_count(self):
        return self.total_row_count() - self.row_count()

    def selects_any(self):
        return self.row_count() > 0

    def boundaries(self):
        """Ascending positions where the selection flips (first run is a skip, empty when the first is 0)."""
        out = L.ArrayOut()
        v = self.mask.view()
        self.ctx.check(self.ctx.lib.ah_selection_boundaries(self.ctx.handle, C.byref(v), C.byref(out)))
        return A.Array._from_out(self.ctx, out, A.Int64)

    def selectors(self):
        """`mask_to_selectors` (boolean.rs:172-190)."""
        total = self.total_row_count()
        if total == 0:
            return []
        b = self.boundaries()
        bounds = b.values_numpy().tolist() if b.length else []
        out, prev, skip = [], 0, True
        for p in bounds + [total]:
            if p > prev:
                out.append(RowSelector(p - prev, skip))
            prev, skip = p, not skip
        return out

    def iter(self):
        return iter(self.selectors())

    def __eq__(self, other):
        if not isinstance(other, RowSelection):
            return NotImplemented
        if self.total_row_count() != other.total_row_count():
            return False
        if self.total_row_count() == 0:
            return True
        from .compute.kernels import aggregate, cmp
        return aggregate.min_boolean(cmp.eq(self.mask, other.mask)) is True  # `a.mask() == b.mask()` (mod.rs:154), in HBM

    def __repr__(self):
        return f"RowSelection({self.selectors()})"

    # ---- algebra (algebra.rs)
    def _binary(self, fn, other, *extra):
        ctx = self.ctx
        out = L.ArrayOut()
        a, b = self.mask.view(), other.mask.view()
        ctx.check(fn(ctx.handle, *extra, C.byref(a), C.byref(b), C.byref(out)))
        return RowSelection(_out_bool(ctx, out, keepalive=(self.mask,)))

    def and_then(self, other):
        """mod.rs:462: `other` selects among the rows this selection selects."""
        return self._binary(self.ctx.lib.ah_selection_and_then, other)

    def intersection(self, other):
        return self._binary(self.ctx.lib.ah_selection_combine, other, 0)

    def union(self, other):
        return self._binary(self.ctx.lib.ah_selection_combine, other, 1)

    # ---- transforms (boolean.rs:250-309)
    def _find_nth(self, n, start=0):
        pos = C.c_int64()
        v = self.mask.view()
        self.ctx.check(self.ctx.lib.ah_selection_find_nth_set_bit(self.ctx.handle, C.byref(v), start, n, C.byref(pos)))
        return pos.value

    def split_off(self, row_count):
        """mod.rs:408: returns the first `row_count` rows, keeps the rest."""
        total = self.total_row_count()
        if row_count >= total:
            head, self.mask, self._count = self.mask, self.mask.slice(total, 0), None
            return RowSelection(head)
        head = self.mask.slice(0, row_count)
        self.mask, self._count = self.mask.slice(row_count, total - row_count), None
        return RowSelection(head)

    def offset(self, offset):
        """mod.rs:566 / `offset_mask`: skip the first `offset` selected rows."""
        if offset == 0:
            return self
        if offset >= self.row_count():
            return RowSelection._empty(self.ctx)
        pos = self._find_nth(offset)
        zeros = RowSelection.from_boundaries([], pos, self.ctx).mask
        return RowSelection(concat([zeros, self.mask.slice(pos, self.total_row_count() - pos)]))

    def limit(self, limit):
        """mod.rs:585 / `limit_mask`: keep the first `limit` selected rows, drop everything after."""
        return RowSelection(self.mask.slice(0, self._find_nth(limit)))

    def trim(self):
        """mod.rs:535 / `trim_mask`: drop trailing skips."""
        new_len = self._find_nth(self.row_count())
        return self if new_len == self.total_row_count() else RowSelection(self.mask.slice(0, new_len))
