"""Test-side model of arrow_select::coalesce::BatchCoalescer (arrow-select/src/coalesce.rs:148-700)
on HostArrays: the reference's state machine restated in plain Python, using the CPU oracle for
filter / take and numpy slicing / concatenation for copy_rows.  TEST INFRASTRUCTURE ONLY."""
from collections import deque

import numpy as np

from orc import HostArray


def _slice(cols, off, n):
    return [c.slice(off, n) for c in cols]


def _concat(a, b):
    if a.valid is None and b.valid is None:
        valid = None
    else:
        av = a.valid if a.valid is not None else np.ones(len(a), dtype=bool)
        bv = b.valid if b.valid is not None else np.ones(len(b), dtype=bool)
        valid = np.concatenate([av, bv])
    if isinstance(a.values, list) or isinstance(b.values, list):  # strings
        return HostArray(a.data_type, list(a.values) + list(b.values), valid)
    return HostArray(a.data_type, np.concatenate([a.values, b.values]), valid)


class ModelCoalescer:
    def __init__(self, oracle, data_types, target):
        self.oracle, self.dts, self.target = oracle, data_types, target
        self.buf = None  # list of HostArray per column
        self.buffered = 0
        self.completed = deque()
        self.limit = None

    def _append(self, cols):
        self.buf = cols if self.buf is None else [_concat(a, b) for a, b in zip(self.buf, cols)]

    def finish(self):
        if self.buffered == 0:
            return
        out = []
        for c in self.buf:  # NullBufferBuilder: a null buffer only if some null was appended
            nulls = 0 if c.valid is None else int((~c.valid).sum())
            out.append(HostArray(c.data_type, c.values, c.valid if nulls else None))
        self.completed.append(out)
        self.buf, self.buffered = None, 0

    def push(self, cols):
        n = len(cols[0]) if cols else 0
        if n == 0:
            return
        if self.limit is not None and n > self.limit:
            if self.buffered == 0:
                self.completed.append(cols)
                return
            if self.buffered > self.limit:
                self.finish()
                self.completed.append(cols)
                return
        off = 0
        while n > self.target - self.buffered:
            rem = self.target - self.buffered
            self._append(_slice(cols, off, rem))
            self.buffered += rem
            off += rem
            n -= rem
            self.finish()
        if n > 0:
            self._append(_slice(cols, off, n))
        self.buffered += n
        if self.buffered >= self.target:
            self.finish()

    def push_with_filter(self, cols, filt):
        filtered = [self.oracle.filter(c, filt) for c in cols]
        if len(filtered[0]) == 0:
            return
        self.push(filtered)

    def push_with_indices(self, cols, idx):
        self.push([self.oracle.take(c, idx) for c in cols])
