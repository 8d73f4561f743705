"""Host model of the bitmap-backed RowSelection built on the oracle's restatements
(oracle.cpp: orc_selection_and_then / orc_selection_combine / orc_find_nth_set_bit, orc_set_slices) — the
checker for arrow_rs_amd.selection.  Same surface as the device class so the transcribed reference tests
(tests/selection_cases.py) run against both."""
import ctypes as C

import numpy as np

import arrow_rs_amd as A
from arrow_rs_amd.selection import RowSelector
import orc
from orc import HostArray


class ModelSelection:
    oracle = None  # set by the fixture

    def __init__(self, bits):
        self.bits = np.asarray(bits, dtype=bool)

    @classmethod
    def from_boolean_buffer(cls, bits):
        return cls(bits)

    @classmethod
    def from_filters(cls, filters):
        return cls(np.concatenate([np.asarray(f, dtype=bool) for f in filters]) if filters else np.zeros(0, bool))

    @classmethod
    def from_selectors(cls, selectors):
        return cls(np.concatenate([np.full(s.row_count, not s.skip) for s in selectors] + [np.zeros(0, bool)]))

    @classmethod
    def from_consecutive_ranges(cls, ranges, total_rows):
        bits, last = np.zeros(total_rows, bool), 0
        for s, e in ranges:
            if e - s == 0:
                continue
            if s < last:
                raise A.Panic("out of order")
            bits[s:e] = True
            last = e
        return cls(bits)

    def as_bools(self):
        return self.bits.tolist()

    def total_row_count(self):
        return len(self.bits)

    def row_count(self):
        return int(self.bits.sum())

    def skipped_row_count(self):
        return self.total_row_count() - self.row_count()

    def selects_any(self):
        return bool(self.bits.any())

    def selectors(self):  # mask_to_selectors over set_slices (boolean.rs:172-190)
        total = len(self.bits)
        if total == 0:
            return []
        out, last_end = [], 0
        for start, end in self.oracle.set_slices(self.bits):
            if start > last_end:
                out.append(RowSelector.skip_rows(start - last_end))
            out.append(RowSelector.select(end - start))
            last_end = end
        if last_end != total:
            out.append(RowSelector.skip_rows(total - last_end))
        return out

    def __eq__(self, other):
        return np.array_equal(self.bits, other.bits)

    def _h(self):
        return HostArray(A.Boolean, self.bits)

    def and_then(self, other):
        return ModelSelection(self.oracle.selection_and_then(self._h(), other._h()).values)

    def intersection(self, other):
        return ModelSelection(self.oracle.selection_combine(0, self._h(), other._h()).values)

    def union(self, other):
        return ModelSelection(self.oracle.selection_combine(1, self._h(), other._h()).values)

    def _find_nth(self, n, start=0):
        return self.oracle.find_nth_set_bit(self.bits, start, n)

    def split_off(self, row_count):
        head, self.bits = self.bits[:row_count], self.bits[row_count:]
        return ModelSelection(head)

    def offset(self, offset):
        if offset == 0:
            return self
        if offset >= self.row_count():
            return ModelSelection(np.zeros(0, bool))
        pos = self._find_nth(offset)
        out = self.bits.copy()
        out[:pos] = False
        return ModelSelection(out)

    def limit(self, limit):
        return ModelSelection(self.bits[:self._find_nth(limit)])

    def trim(self):
        return ModelSelection(self.bits[:self._find_nth(self.row_count())])


class DeviceAdapter:
    """Gives arrow_rs_amd.selection.RowSelection the constructors the shared cases use (python bools in)."""

    def __init__(self, ctx):
        from arrow_rs_amd.selection import RowSelection
        self.ctx, self.R = ctx, RowSelection

    def from_boolean_buffer(self, bits):
        return self.R.from_boolean_buffer(A.Array.from_numpy(np.asarray(bits, dtype=bool), ctx=self.ctx, bit_offset=3))

    def from_filters(self, filters):
        return self.R.from_filters([A.Array.from_numpy(np.asarray(f, dtype=bool), ctx=self.ctx, bit_offset=i % 5)
                                    for i, f in enumerate(filters)], ctx=self.ctx)

    def from_selectors(self, selectors):
        return self.R.from_selectors(selectors, self.ctx)

    def from_consecutive_ranges(self, ranges, total_rows):
        return self.R.from_consecutive_ranges(ranges, total_rows, self.ctx)


def bools_of(sel):
    if isinstance(sel, ModelSelection):
        return sel.as_bools()
    m = sel.as_mask()
    return m.values_numpy().tolist() if m.length else []


def bits_of(sel):
    """numpy form of bools_of for the large fuzz cases"""
    if isinstance(sel, ModelSelection):
        return sel.bits
    m = sel.as_mask()
    return m.values_numpy() if m.length else np.zeros(0, bool)
