"""arrow::compute::BatchCoalescer == arrow_select::coalesce::BatchCoalescer
(arrow-select/src/coalesce.rs:148-700): primitive columns (coalesce/primitive.rs) through the fused
filter-into-builder path, every other supported layout (Boolean, Utf8, LargeUtf8) through the reference's
generic buffer-and-concat implementation (coalesce/generic.rs).

The host state machine (exact-size output batches, in input order, optional large-batch bypass) is NATIVE —
``ah_coalescer_*`` (csrc/coalesce.hip): one C call per pushed batch, no wait except the predicate's count, one wait per
finished batch; Boolean / Utf8 / LargeUtf8 columns are the native GenericInProgressArray there (pieces + concat), Utf8View /
BinaryView columns the native InProgressByteViewArray (views as a 16-byte column; the data-buffer LISTS are kept here, the
library shifts buffer indices and reports which inputs make up each output batch).  The Python restatement below is only
the fallback for schemas the native object refuses (> 200 columns); the data movement happens in HBM either way:
  * ``copy_rows``                 -> ``ah_copy_rows_into`` (D2D copy + funnel-shift bitmap merge)
  * ``copy_rows_by_filter_from``  -> ``ah_filter_predicate_apply_into``: the filter scatters
    straight into the in-progress buffers (no intermediate filtered array, no second copy).
The reference only fuses when ``selected <= len/16`` (coalesce.rs:45-54, a CPU cache trade-off);
on the GPU the fused scatter is never slower than filter-then-copy, so it is used whenever the
selected rows fit the in-progress batch — the sequence of completed batches is identical.
"""
import ctypes as C
from collections import deque

from ... import _lib as L
from ...array import (Array, Boolean, DeviceBuffer, RecordBatch, InvalidArgumentError, NotYetImplemented,
                      _RawMem)
from .filter import FilterBuilder
from .take import take_record_batch


class _InProgress:
    """InProgressPrimitiveArray (coalesce/primitive.rs:28-52): values + NullBufferBuilder."""

    def __init__(self, ctx, data_type, batch_size, nulls_acc_ptr=None):
        if not data_type.is_primitive() or data_type.width <= 0:
            raise NotYetImplemented(f"BatchCoalescer column type {data_type}")
        self.ctx, self.data_type, self.batch_size = ctx, data_type, batch_size
        self.values = None
        self.validity = None
        self.nulls = 0
        # device word that accumulates the appended null rows: the per-batch calls never wait on the GPU; the
        # coalescer reads all its columns' words in ONE wait when a batch is finished (ah_read_words)
        self.acc = nulls_acc_ptr

    def ensure_capacity(self):  # allocate on first write (primitive.rs:57-61)
        if self.values is None:
            self.values = DeviceBuffer(self.ctx, max(self.batch_size * self.data_type.width, 8))
            self.validity = DeviceBuffer(self.ctx, ((self.batch_size + 63) // 64) * 8)
            self.ctx.check(self.ctx.lib.ah_memset(self.ctx.handle, self.validity.ptr, 0, self.validity.nbytes))

    def copy_rows(self, source, offset, length, at):
        self.ensure_capacity()
        v = source.view()
        if self.acc is not None:
            self.ctx.check(self.ctx.lib.ah_copy_rows_into_acc(self.ctx.handle, C.byref(v), offset, length,
                                                              self.values.ptr, self.validity.ptr, at, self.acc))
            return
        n = C.c_int64()
        self.ctx.check(self.ctx.lib.ah_copy_rows_into(self.ctx.handle, C.byref(v), offset, length,
                                                      self.values.ptr, self.validity.ptr, at, C.byref(n)))
        self.nulls += n.value

    def copy_rows_by_filter_from(self, source, predicate, at):
        self.ensure_capacity()
        v = source.view()
        if self.acc is not None:
            self.ctx.check(self.ctx.lib.ah_filter_predicate_apply_into_acc(
                self.ctx.handle, predicate._h, C.byref(v), self.values.ptr, self.validity.ptr, at, self.acc))
            return
        n = C.c_int64()
        self.ctx.check(self.ctx.lib.ah_filter_predicate_apply_into(
            self.ctx.handle, predicate._h, C.byref(v), self.values.ptr, self.validity.ptr, at, C.byref(n)))
        self.nulls += n.value

    def finish(self, rows):
        w = self.data_type.width
        vals = _RawMem(self.values.ptr, rows * w, self.values)
        # NullBufferBuilder::finish: a buffer only if a null was ever appended
        nmem = _RawMem(self.validity.ptr, self.validity.nbytes, self.validity) if self.nulls else None
        arr = Array(self.ctx, self.data_type, rows, vals, 0, nmem, 0, self.nulls)
        self.values = self.validity = None
        self.nulls = 0
        return arr


class _InProgressGeneric:
    """GenericInProgressArray (coalesce/generic.rs:32-108): buffers slices / filtered arrays, `concat` on finish."""

    def __init__(self, ctx, data_type, batch_size):
        if data_type.physical not in (L.AH_BOOL, L.AH_UTF8, L.AH_LARGE_UTF8):
            raise NotYetImplemented(f"BatchCoalescer column type {data_type}")
        self.ctx, self.data_type = ctx, data_type
        self.buffered = []

    def copy_rows(self, source, offset, length, at):
        self.buffered.append(source.slice(offset, length))

    def copy_rows_by_filter_from(self, source, predicate, at):
        self.buffered.append(predicate.filter(source))

    def finish(self, rows):
        from .concat import concat
        arr = self.buffered[0] if len(self.buffered) == 1 else concat(self.buffered)
        self.buffered = []
        return arr


def _destroy_native(lib, ctx_handle, handle, pending_fins):
    for f in list(pending_fins):
        if f.alive:
            f()  # ah_coalescer_push_abort, once
    del pending_fins[:]
    lib.ah_coalescer_destroy(ctx_handle, handle)


def _in_progress(ctx, data_type, batch_size, acc=None):  # `create_in_progress_array` (coalesce.rs:673-700)
    if data_type.is_primitive() and data_type.width > 0 and data_type.physical not in (L.AH_UTF8_VIEW, L.AH_BINARY_VIEW):
        return _InProgress(ctx, data_type, batch_size, acc)
    return _InProgressGeneric(ctx, data_type, batch_size)


class BatchCoalescer:
    """``BatchCoalescer::new(schema, target_batch_size)``; schema = (names, data types)."""

    def __init__(self, names, data_types, target_batch_size, ctx=None):
        from ...array import default_context
        self.ctx = ctx or default_context()
        self.names = list(names)
        self.data_types = list(data_types)
        self.target_batch_size = int(target_batch_size)
        self.biggest_coalesce_batch_size = None
        self._native = None
        def native_ok(dt):  # fixed-width (InProgressPrimitiveArray), Boolean / Utf8 / LargeUtf8 (GenericInProgressArray),
            if dt.physical in (L.AH_BOOL, L.AH_UTF8, L.AH_LARGE_UTF8, L.AH_UTF8_VIEW, L.AH_BINARY_VIEW):  # views (InProgressByteViewArray)
                return True
            return dt.is_primitive() and dt.width > 0
        if self.data_types and all(native_ok(dt) for dt in self.data_types) and len(self.data_types) <= 200:
            lib, h = self.ctx.lib, C.c_void_p()
            types = (C.c_int32 * len(self.data_types))(*[dt.physical for dt in self.data_types])
            self.ctx.check(lib.ah_coalescer_create(self.ctx.handle, len(self.data_types), types, self.target_batch_size, C.byref(h)))
            self._native = h
            self._tagged, self._next_tag = {}, 1
            # view columns (InProgressByteViewArray, coalesce/byte_view.rs:39): the variadic data buffers stay on this side
            # of the C ABI.  Every pushed batch is an input with a sequence number (the library counts the same way);
            # `_inputs[seq]` = its view columns' buffer lists until no completed batch can refer to it any more
            self._view_cols = [i for i, dt in enumerate(self.data_types) if dt.physical in (L.AH_UTF8_VIEW, L.AH_BINARY_VIEW)]
            self._seq, self._inputs = 0, {}
            import weakref
            # begun pushes that were never ended hold native state INSIDE the coalescer: they are aborted before it is
            # destroyed, whichever of the two Python objects is collected first
            self._pending_fins = []
            self._fin = weakref.finalize(self, _destroy_native, lib, self.ctx.handle, h, self._pending_fins)
            return
        # one device word per column for the appended-null counts (zeroed once; ah_read_words resets them)
        ncols = max(len(self.data_types), 1)
        self._acc = DeviceBuffer(self.ctx, ncols * 8) if ncols <= 200 else None
        if self._acc is not None:
            self.ctx.check(self.ctx.lib.ah_memset(self.ctx.handle, self._acc.ptr, 0, self._acc.nbytes))
        self.in_progress = [_in_progress(self.ctx, dt, self.target_batch_size,
                                         None if self._acc is None else self._acc.ptr + 8 * i)
                            for i, dt in enumerate(self.data_types)]
        self.buffered_rows = 0
        self.completed = deque()

    @classmethod
    def new(cls, names, data_types, target_batch_size, ctx=None):
        return cls(names, data_types, target_batch_size, ctx)

    def with_biggest_coalesce_batch_size(self, limit):
        self.set_biggest_coalesce_batch_size(limit)
        return self

    def set_biggest_coalesce_batch_size(self, limit):
        self.biggest_coalesce_batch_size = limit
        if self._native is not None:
            self.ctx.lib.ah_coalescer_set_biggest_coalesce_batch_size(self._native, -1 if limit is None else int(limit))

    def get_buffered_rows(self):
        if self._native is not None:
            return self.ctx.lib.ah_coalescer_buffered_rows(self._native)
        return self.buffered_rows

    def is_empty(self):
        return self.get_buffered_rows() == 0 and not self.has_completed_batch()

    def has_completed_batch(self):
        if self._native is not None:
            return self.ctx.lib.ah_coalescer_completed_count(self._native) > 0
        return bool(self.completed)

    def completed_count(self):
        """number of finished batches waiting in the queue"""
        if self._native is not None:
            return int(self.ctx.lib.ah_coalescer_completed_count(self._native))
        return len(self.completed)

    def next_completed_batch(self):
        if self._native is None:
            return self.completed.popleft() if self.completed else None
        n = len(self.data_types)
        outs = (L.ArrayOut * n)()
        rows, tag = C.c_int64(), C.c_uint64()
        self._pending_sources = self._read_sources() if self._view_cols else []
        self.ctx.check(self.ctx.lib.ah_coalescer_next_completed_batch(self.ctx.handle, self._native, outs, C.byref(rows), C.byref(tag)))
        if rows.value < 0:
            return None
        if tag.value:  # the caller's own batch, passed through untouched (large-batch bypass)
            return self._tagged.pop(tag.value)
        cols = []
        for i, dt in enumerate(self.data_types):
            o = L.ArrayOut()
            C.memmove(C.byref(o), C.byref(outs[i]), C.sizeof(L.ArrayOut))
            cols.append(Array._from_out(self.ctx, o, dt))
        if self._view_cols:
            for i in self._view_cols:  # the inputs that contributed rows, in order: their buffer lists, concatenated
                cols[i].data_buffers = [b for sq in self._pending_sources for b in self._inputs[sq][i]]
            for sq in [k for k in self._inputs if self._pending_sources and k < self._pending_sources[0]]:
                del self._inputs[sq]  # inputs contribute in push order: nothing older can show up again
        return RecordBatch(self.names, cols, num_rows=rows.value)

    def next_completed_batches(self, max_batches=1 << 30):
        """Up to ``max_batches`` finished batches with ONE C call (``ah_coalescer_next_completed_batches``): a grouped push of
        8192-row batches — the reference's operating point, coalesce.rs:172-173 — completes thousands of output batches
        at once.  View schemas and bypassed batches keep the one-at-a-time path (their bookkeeping is per batch)."""
        out = []
        if self._native is None or self._view_cols or self._tagged:
            while len(out) < max_batches:
                b = self.next_completed_batch()
                if b is None:
                    break
                out.append(b)
            return out
        nc = len(self.data_types)
        want = min(max_batches, self.completed_count())
        if want <= 0:
            return out
        outs = (L.ArrayOut * (want * nc))()
        rows = (C.c_int64 * want)()
        n = C.c_int32()
        st = self.ctx.lib.ah_coalescer_next_completed_batches(self.ctx.handle, self._native, want, outs, rows, None, C.byref(n))
        # the batches the call popped before it failed are owned HERE now: wrap them first, then raise
        for j in range(n.value):
            cols = []
            for i, dt in enumerate(self.data_types):
                o = L.ArrayOut()
                C.memmove(C.byref(o), C.byref(outs[j * nc + i]), C.sizeof(L.ArrayOut))
                cols.append(Array._from_out(self.ctx, o, dt))
            out.append(RecordBatch(self.names, cols, num_rows=rows[j]))
        self.ctx.check(st)
        return out

    def _slab_schema(self):
        """the library's slab push takes this schema (csrc/coalesce.hip slab_eligible): any number of batches per call"""
        return (self._native is not None and self.biggest_coalesce_batch_size is None and not self._view_cols
                and self.target_batch_size % 64 == 0 and len(self.data_types) <= 8
                and all(dt.is_primitive() and dt.width in (1, 2, 4, 8) and dt.physical != L.AH_BOOL for dt in self.data_types))

    def _read_sources(self):
        """which inputs make up the FRONT completed batch (``ah_coalescer_completed_batch_sources``)"""
        n = C.c_int32()
        self.ctx.check(self.ctx.lib.ah_coalescer_completed_batch_sources(self.ctx.handle, self._native, None, 0, C.byref(n)))
        seqs = (C.c_uint64 * max(n.value, 1))()
        self.ctx.check(self.ctx.lib.ah_coalescer_completed_batch_sources(self.ctx.handle, self._native, seqs, n.value, C.byref(n)))
        return [int(seqs[i]) for i in range(n.value)]

    def _declare_inputs(self, batches):
        """before the native push of `batches` (view schemas): their data-buffer counts go to the library, their buffer
        lists stay here under the sequence numbers the library will give them"""
        if not self._view_cols:
            return
        nc = len(self.data_types)
        counts = (C.c_int32 * (len(batches) * nc))()
        for b_i, b in enumerate(batches):
            bufs = {}
            for i in self._view_cols:
                bufs[i] = list(b.columns[i].data_buffers or [])
                counts[b_i * nc + i] = len(bufs[i])
            self._inputs[self._seq] = bufs
            self._seq += 1
        self.ctx.check(self.ctx.lib.ah_coalescer_declare_view_buffers(self.ctx.handle, self._native, len(batches), counts))

    def _views(self, batch):
        if batch.num_columns() != len(self.data_types):
            raise InvalidArgumentError(
                f"Batch has {batch.num_columns()} columns but BatchCoalescer expects {len(self.data_types)}")
        views = (L.ArrayView * len(self.data_types))()
        for i, c in enumerate(batch.columns):
            views[i] = c.view()
        return views

    def _native_push(self, fn, batch, *args):
        """one C call per pushed batch; a bypassed batch (large-batch cases 1 / 2) is kept alive under its tag until it
        comes back from next_completed_batch"""
        tag = self._next_tag
        self._next_tag += 1
        bypassed = C.c_int32()
        self._declare_inputs([batch])
        self.ctx.check(fn(self.ctx.handle, self._native, *args, tag, C.byref(bypassed)))
        if bypassed.value:
            self._tagged[tag] = batch

    # ---- coalesce.rs:229
    def push_batch_with_filter(self, batch, filter):
        if filter.data_type != Boolean:
            raise InvalidArgumentError(f"filter predicate must be Boolean, got {filter.data_type}")
        filter_len, rows = filter.length, batch.num_rows()
        if filter_len > rows:
            raise InvalidArgumentError(
                f"Filter predicate of length {filter_len} is larger than target array of length {rows}")
        if self._native is not None:
            views, fv = self._views(batch), filter.view()
            return self._native_push(self.ctx.lib.ah_coalescer_push_batch_with_filter, batch, views, rows, C.byref(fv))
        predicate = FilterBuilder.new(filter).optimize().build()  # one count pass for all columns
        selected = predicate.count()
        if selected == 0:
            return
        if selected == rows and filter_len == rows:
            return self.push_batch(batch)
        if batch.num_columns() != len(self.in_progress):
            raise InvalidArgumentError(
                f"Batch has {batch.num_columns()} columns but BatchCoalescer expects {len(self.in_progress)}")
        exceeds = self.biggest_coalesce_batch_size is not None and selected > self.biggest_coalesce_batch_size
        does_not_fit = selected > self.target_batch_size - self.buffered_rows
        if exceeds or does_not_fit:  # materialise, then split across output batches
            return self.push_batch(predicate.filter_record_batch(batch))
        for ip, col in zip(self.in_progress, batch.columns):
            ip.copy_rows_by_filter_from(col, predicate, self.buffered_rows)
        self.buffered_rows += selected
        if self.buffered_rows >= self.target_batch_size:
            self.finish_buffered_batch()

    def push_batches_with_filters(self, pairs):
        """``push_batch_with_filter`` for a list of (batch, filter) pairs — the same output batches in the same order, but
        with ONE host wait for all the predicate counts (``ah_coalescer_push_batches_with_filters``): for a caller that
        has several batches queued.  Non-native schemas fall back to one push per pair."""
        pairs = list(pairs)
        if self._native is None or not pairs:
            for b, f in pairs:
                self.push_batch_with_filter(b, f)
            return
        n, nc = len(pairs), len(self.data_types)
        views = (L.ArrayView * (n * nc))()
        fviews = (L.ArrayView * n)()
        rows = (C.c_int64 * n)()
        tags = (C.c_uint64 * n)()
        bypassed = (C.c_int32 * n)()
        for i, (b, f) in enumerate(pairs):
            if f.data_type != Boolean:
                raise InvalidArgumentError(f"filter predicate must be Boolean, got {f.data_type}")
            if b.num_columns() != nc:
                raise InvalidArgumentError(f"Batch has {b.num_columns()} columns but BatchCoalescer expects {nc}")
            for k, c in enumerate(b.columns):
                views[i * nc + k] = c.view()
            fviews[i] = f.view()
            rows[i] = b.num_rows()
            tags[i] = self._next_tag
            self._next_tag += 1
        self._declare_inputs([b for b, _f in pairs])
        self.ctx.check(self.ctx.lib.ah_coalescer_push_batches_with_filters(self.ctx.handle, self._native, n, views, rows, fviews,
                                                                           tags, bypassed))
        for i, (b, _f) in enumerate(pairs):
            if bypassed[i]:
                self._tagged[tags[i]] = b

    def push_batches_with_filters_begin(self, pairs):
        """First half of ``push_batches_with_filters`` (``ah_coalescer_push_batches_with_filters_begin``): the count passes
        of up to 64 (batch, filter) pairs are ENQUEUED and the call returns; ``.end()`` of the returned handle waits for
        the counts and appends the batches.  An engine that already holds the next group calls ``begin`` for it BEFORE
        ``end`` of the current one, so the GPU never idles for a count round trip:

            pending = co.push_batches_with_filters_begin(groups[0])
            for g in groups[1:]:
                nxt = co.push_batches_with_filters_begin(g)
                pending.end()
                pending = nxt
            pending.end()

        Same output batches, same order as the one-call form.  Non-native schemas: everything happens in ``end()``."""
        pairs = list(pairs)
        if self._native is None or not pairs:  # (any number of pairs: the library counts groups of 64 itself when its slab path declines)
            return _PendingPush(self, pairs, None, None, None)
        n, nc = len(pairs), len(self.data_types)
        views = (L.ArrayView * (n * nc))()
        fviews = (L.ArrayView * n)()
        rows = (C.c_int64 * n)()
        tags = (C.c_uint64 * n)()
        for i, (b, f) in enumerate(pairs):
            if f.data_type != Boolean:
                raise InvalidArgumentError(f"filter predicate must be Boolean, got {f.data_type}")
            if b.num_columns() != nc:
                raise InvalidArgumentError(f"Batch has {b.num_columns()} columns but BatchCoalescer expects {nc}")
            for k, c in enumerate(b.columns):
                views[i * nc + k] = c.view()
            fviews[i] = f.view()
            rows[i] = b.num_rows()
            tags[i] = self._next_tag
            self._next_tag += 1
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.ah_coalescer_push_batches_with_filters_begin(self.ctx.handle, self._native, n, views, rows, fviews,
                                                                                 tags, C.byref(h)))
        return _PendingPush(self, pairs, h, tags, (views, fviews, rows))

    # ---- coalesce.rs:257
    def push_batch_with_indices(self, batch, indices):
        if self._native is not None:
            views, iv = self._views(batch), indices.view()
            self._declare_inputs([batch])
            return self.ctx.check(self.ctx.lib.ah_coalescer_push_batch_with_indices(self.ctx.handle, self._native, views,
                                                                                   batch.num_rows(), C.byref(iv)))
        return self.push_batch(take_record_batch(batch, indices))

    # ---- coalesce.rs:296-525
    def push_batch(self, batch):
        batch_size = batch.num_rows()
        if batch_size == 0:
            return
        if self._native is not None:
            return self._native_push(self.ctx.lib.ah_coalescer_push_batch, batch, self._views(batch), batch_size)
        limit = self.biggest_coalesce_batch_size
        if limit is not None and batch_size > limit:
            if self.buffered_rows == 0:          # case 1: bypass
                self.completed.append(batch)
                return
            if self.buffered_rows > limit:       # case 2: flush, then bypass
                self.finish_buffered_batch()
                self.completed.append(batch)
                return
        if batch.num_columns() != len(self.in_progress):
            raise InvalidArgumentError(
                f"Batch has {batch.num_columns()} columns but BatchCoalescer expects {len(self.in_progress)}")
        num_rows, offset = batch_size, 0
        while num_rows > self.target_batch_size - self.buffered_rows:
            remaining = self.target_batch_size - self.buffered_rows
            for ip, col in zip(self.in_progress, batch.columns):
                ip.copy_rows(col, offset, remaining, self.buffered_rows)
            self.buffered_rows += remaining
            offset += remaining
            num_rows -= remaining
            self.finish_buffered_batch()
        if num_rows > 0:
            for ip, col in zip(self.in_progress, batch.columns):
                ip.copy_rows(col, offset, num_rows, self.buffered_rows)
        self.buffered_rows += num_rows
        if self.buffered_rows >= self.target_batch_size:
            self.finish_buffered_batch()

    # ---- coalesce.rs:536
    def finish_buffered_batch(self):
        if self._native is not None:
            return self.ctx.check(self.ctx.lib.ah_coalescer_finish_buffered_batch(self.ctx.handle, self._native))
        if self.buffered_rows == 0:
            return
        if self._acc is not None:  # the ONE wait of this output batch: every column's appended-null count
            n = len(self.in_progress)
            host = (C.c_uint64 * n)()
            self.ctx.check(self.ctx.lib.ah_read_words(self.ctx.handle, self._acc.ptr, n, host, 1))
            for ip, v in zip(self.in_progress, host):
                if isinstance(ip, _InProgress):
                    ip.nulls += int(v)
        cols = [ip.finish(self.buffered_rows) for ip in self.in_progress]
        self.completed.append(RecordBatch(self.names, cols, num_rows=self.buffered_rows))
        self.buffered_rows = 0


class _PendingPush:
    """A grouped push between ``push_batches_with_filters_begin`` and ``end``; keeps the input batches alive."""

    def __init__(self, co, pairs, handle, tags, keep):
        self.co, self.pairs, self.handle, self.tags, self._keep = co, pairs, handle, tags, keep
        self._fin = None
        if handle is not None:  # a handle that is never ended must not keep its pinned slot and predicates
            import weakref
            self._fin = weakref.finalize(self, co.ctx.lib.ah_coalescer_push_abort, co.ctx.handle, co._native, handle)
            co._pending_fins[:] = [f for f in co._pending_fins if f.alive] + [self._fin]

    def abort(self):
        """give the push up without appending its batches (``ah_coalescer_push_abort``)"""
        self.pairs = None
        if self._fin is not None and self._fin.alive:
            self._fin()  # calls the abort once
        self.handle = self._keep = None

    def end(self):
        co, pairs = self.co, self.pairs
        if pairs is None:
            raise RuntimeError("push already ended")
        self.pairs = None
        if self.handle is None:  # not native (or more than 64 pairs): the one-call form
            return co.push_batches_with_filters(pairs)
        n = len(pairs)
        bypassed = (C.c_int32 * n)()
        h, self.handle = self.handle, None
        if self._fin is not None:
            self._fin.detach()  # _end consumes the handle, also when it fails
        co._declare_inputs([b for b, _f in pairs])  # (the library numbers a group's batches when they are APPENDED: now)
        co.ctx.check(co.ctx.lib.ah_coalescer_push_batches_with_filters_end(co.ctx.handle, co._native, h, bypassed))
        for i, (b, _f) in enumerate(pairs):
            if bypassed[i]:
                co._tagged[self.tags[i]] = b
        self._keep = None
