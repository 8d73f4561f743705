"""Row-range sharding of a column across the GPUs of one node + RCCL reassembly.

The reference has no parallelism of any kind (kernels are synchronous pure functions,
SURVEY.md §2); this module is new capability required by the north star: a RecordBatch is
split by contiguous row range (``RecordBatch::slice`` semantics,
arrow-array/src/record_batch.rs:681), each GPU runs the single-GPU kernels on its range, and
the global result is the in-order concatenation of the shard results — the oracle for which is
``arrow_select::concat`` (arrow-select/src/concat.rs:495, primitives :334-343).

There is ONE exchange step, an all-gatherv with counts known only after the filter:

  1. all-gather of (len, null_count, has_validity) per rank  (3 x i64)
  2. values: every rank receives each peer's piece DIRECTLY at its final offset with grouped
     point-to-point send/recv (``batch_isend_irecv`` -> ncclGroupStart/End): xGMI is a
     point-to-point fabric (7 links per GPU), so a direct exchange drives all links at once
     whereas a ring all-gather is bound by one link.  RCCL has no native all-gatherv.
  3. validity: pieces land at bit offset sum(len_<r), generally not byte aligned, so they are
     gathered into a staging buffer and merged by the funnel-shift kernel
     (``ah_bitmap_set_bits``; reference analogue arrow-buffer/src/util/bit_mask.rs:33).

``Communicator.all_gather_batches`` is the multi-column form: every rank frames its RecordBatch as ONE Arrow
IPC message (``ipc.encode_batch``: all buffers of all columns in one contiguous HBM body, metadata appended),
one grouped exchange moves the messages, and each peer's batch is decoded as zero-copy views of the receive
buffer — a "chunked" result (one batch per rank, in rank order), which is what ``BatchCoalescer`` consumes.

One process per GPU (``torch.distributed``, backend "nccl" == RCCL on ROCm).  torch is only
used for the collective; device memory stays owned by the arrow_hip context.
"""
import ctypes as C

from . import _lib as L
from .array import Array, DeviceBuffer, _RawMem, NotYetImplemented


def shard_range(n_rows, rank, world, align=64):
    """Rows [start, end) owned by ``rank``: contiguous, boundaries rounded to a multiple of
    ``align`` rows so input bitmaps split on word boundaries."""
    per = -(-n_rows // world)
    per = -(-per // align) * align
    start = min(n_rows, rank * per)
    end = min(n_rows, start + per)
    return start, end


def exclusive_offsets(sizes):
    offs, acc = [], 0
    for s in sizes:
        offs.append(acc)
        acc += int(s)
    return offs, acc


def all_gatherv_bytes(dist, local, out, offsets, sizes, group=None):
    """Every rank ends with ``out[offsets[r]:offsets[r]+sizes[r]] == rank r's local`` for all r.
    ``local`` / ``out`` are 1-D uint8 tensors (CUDA with nccl, CPU with gloo)."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    if sizes[rank]:
        out[offsets[rank]:offsets[rank] + sizes[rank]].copy_(local[:sizes[rank]])
    if world == 1:
        return
    ops = []
    for step in range(1, world):  # stagger peers so every link is busy in both directions
        dst = (rank + step) % world
        src = (rank - step) % world
        if sizes[rank]:
            ops.append(dist.P2POp(dist.isend, local[:sizes[rank]], dst, group))
        if sizes[src]:
            ops.append(dist.P2POp(dist.irecv, out[offsets[src]:offsets[src] + sizes[src]], src, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


class _CudaView:
    """Zero-copy hand-off of an arrow_hip device range to torch (__cuda_array_interface__ v2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1",
                                         "data": (int(ptr), False), "version": 2, "strides": None}


class Communicator:
    """RCCL-backed reassembly of row-sharded results for one arrow_hip context."""

    def __init__(self, ctx, dist, group=None):
        import torch
        self.torch, self.dist, self.ctx, self.group = torch, dist, ctx, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = torch.device("cuda", ctx.device)

    def _tensor(self, ptr, nbytes):
        if nbytes == 0:
            return self.torch.empty(0, dtype=self.torch.uint8, device=self.device)
        return self.torch.as_tensor(_CudaView(ptr, nbytes), device=self.device)

    def all_gather_batches(self, batch, alignment=64):
        """Every rank's RecordBatch, in rank order, as zero-copy views into one receive buffer.
        One count exchange + one grouped send/recv per call, however many columns the batch has."""
        from . import ipc
        torch, dist, ctx = self.torch, self.dist, self.ctx
        meta, body = ipc.encode_batch(batch, alignment)
        mlen = len(meta)
        # message = body | metadata (metadata is a multiple of `alignment`, so the next slot stays aligned)
        msg = DeviceBuffer(ctx, max(body.nbytes + mlen, 8))
        lib, h = ctx.lib, ctx.handle
        ctx.check(lib.ah_memcpy_dtod(h, msg.ptr, body.ptr, body.nbytes))
        mbuf = (C.c_char * mlen).from_buffer_copy(meta)
        ctx.check(lib.ah_memcpy_htod(h, msg.ptr + body.nbytes, mbuf, mlen))
        mine = torch.tensor([body.nbytes, mlen], dtype=torch.int64, device=self.device)
        allc = torch.empty(self.world * 2, dtype=torch.int64, device=self.device)
        ctx.synchronize()
        dist.all_gather_into_tensor(allc, mine, group=self.group)
        counts = allc.view(self.world, 2).cpu().tolist()
        sizes = [b + m for b, m in counts]
        offs, total = exclusive_offsets(sizes)
        recv = DeviceBuffer(ctx, max(total, 8))
        all_gatherv_bytes(dist, self._tensor(msg.ptr, sizes[self.rank]), self._tensor(recv.ptr, total), offs, sizes,
                          self.group)
        torch.cuda.synchronize(self.device)
        schema = ipc.Schema.of(batch)
        out = []
        for r in range(self.world):
            blen, ml = counts[r]
            host = (C.c_char * ml)()
            ctx.check(lib.ah_memcpy_dtoh(h, host, recv.ptr + offs[r] + blen, ml))
            out.append(ipc.decode_batch(bytes(host), recv.ptr + offs[r], blen, schema, ctx, keepalive=(recv,)))
        return out

    def all_gatherv(self, array):
        """Concatenation of every rank's ``array`` in rank order, materialised on every rank
        (== arrow_select::concat of the shard results)."""
        import time as _t
        torch, dist, ctx = self.torch, self.dist, self.ctx
        tm = {}
        _t0 = _t.perf_counter()
        dt = array.data_type
        w = dt.width
        if w <= 0:
            raise NotYetImplemented(f"all_gatherv of {dt}")
        has_v = 1 if (array.validity is not None and array.null_count() > 0) else 0
        mine = torch.tensor([array.length, array.null_count(), has_v], dtype=torch.int64, device=self.device)
        allc = torch.empty(self.world * 3, dtype=torch.int64, device=self.device)  # flat: gloo and nccl agree
        ctx.synchronize()  # array's producer kernels ran on the context stream
        dist.all_gather_into_tensor(allc, mine, group=self.group)
        counts = allc.view(self.world, 3).cpu().tolist()
        tm["counts"] = _t.perf_counter() - _t0
        lens = [c[0] for c in counts]
        row_offs, total = exclusive_offsets(lens)
        any_valid = any(c[2] for c in counts)

        # values: straight to their final offsets
        out_vals = DeviceBuffer(ctx, max(total * w, 8))
        local = self._tensor(array.values.ptr, array.length * w) if array.length else self._tensor(0, 0)
        out_t = self._tensor(out_vals.ptr, total * w)
        all_gatherv_bytes(dist, local, out_t, [o * w for o in row_offs], [n * w for n in lens], self.group)
        tm["values"] = _t.perf_counter() - _t0

        vmem, nulls = None, 0
        if any_valid:
            lib, h = ctx.lib, ctx.handle
            pbytes = [((n + 63) // 64) * 8 for n in lens]
            poffs, ptotal = exclusive_offsets(pbytes)
            # local validity packed at bit offset 0 (all ones if this shard carries no nulls)
            mine_bits = DeviceBuffer(ctx, max(pbytes[self.rank], 8))
            ctx.check(lib.ah_memset(h, mine_bits.ptr, 0, mine_bits.nbytes))
            if array.length:
                src = array.validity.ptr if has_v else None
                ctx.check(lib.ah_bitmap_set_bits(h, mine_bits.ptr, 0, src,
                                                 array.validity_bit_offset if has_v else 0, array.length, None))
            ctx.synchronize()
            staging = DeviceBuffer(ctx, max(ptotal, 8))
            all_gatherv_bytes(dist, self._tensor(mine_bits.ptr, pbytes[self.rank]),
                              self._tensor(staging.ptr, ptotal), poffs, pbytes, self.group)
            torch.cuda.synchronize(self.device)
            out_valid = DeviceBuffer(ctx, ((total + 63) // 64) * 8)
            ctx.check(lib.ah_memset(h, out_valid.ptr, 0, out_valid.nbytes))
            for r in range(self.world):
                if lens[r]:
                    ctx.check(lib.ah_bitmap_set_bits(h, out_valid.ptr, row_offs[r], staging.ptr + poffs[r], 0,
                                                     lens[r], None))
            ctx.synchronize()
            nulls = sum(c[1] for c in counts)
            vmem = _RawMem(out_valid.ptr, out_valid.nbytes, out_valid)
        else:
            torch.cuda.synchronize(self.device)
        tm["total"] = _t.perf_counter() - _t0
        self.timings = {k: round(v * 1e3, 3) for k, v in tm.items()}
        # what this rank pushed to EACH peer and pulled in total (values + packed validity), for per-link rates
        sent = array.length * w + (((array.length + 63) // 64) * 8 if any_valid else 0)
        recv = (total - array.length) * w + (sum(((n + 63) // 64) * 8 for n in lens) - ((array.length + 63) // 64) * 8
                                              if any_valid else 0)
        self.last_exchange = {"peers": self.world - 1, "bytes_to_each_peer": int(sent), "bytes_received": int(recv)}
        return Array(ctx, dt, total, _RawMem(out_vals.ptr, total * w, out_vals), 0, vmem, 0, nulls)


class CApiCommunicator:
    """The exchange step through the C ABI (``ah_comm_*`` / ``ah_all_gatherv``, csrc/comm.hip): libarrow_hip.so binds
    RCCL itself, so a Rust (or any FFI) host reassembles shards with the same calls and torch is nowhere in the data
    path.  The only thing a host must provide is a way to ship rank 0's 128-byte unique id to the other ranks:
    ``share_id(payload_or_None) -> payload`` (bench.py uses a gloo broadcast; an engine would use its control plane)."""

    def __init__(self, ctx, rank, world, share_id=None, use_rccl=True):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        lib = ctx.lib
        raw = None
        if use_rccl or world > 1:
            buf = C.create_string_buffer(128)
            if self.rank == 0:
                ctx.check(lib.ah_comm_unique_id(ctx.handle, buf))
            raw = bytes(buf.raw)
            if world > 1:
                if share_id is None:
                    raise ValueError("a multi-rank communicator needs share_id to distribute rank 0's unique id")
                raw = bytes(share_id(raw if self.rank == 0 else None))
        h = C.c_void_p()
        ctx.check(lib.ah_comm_create(ctx.handle, self.rank, self.world, raw, C.byref(h)))
        self._h = h
        import weakref
        self._fin = weakref.finalize(self, lib.ah_comm_destroy, ctx.handle, h)
        self.timings, self.last_exchange = None, None

    def barrier(self):
        self.ctx.check(self.ctx.lib.ah_comm_barrier(self.ctx.handle, self._h))

    def allreduce_max(self, values):
        arr = (C.c_double * len(values))(*[float(v) for v in values])
        self.ctx.check(self.ctx.lib.ah_comm_allreduce_max_f64(self.ctx.handle, self._h, arr, len(values)))
        return list(arr)

    def _note(self, st):
        self.timings = {"counts": round(st.counts_ms, 3), "total": round(st.total_ms, 3)}
        self.last_exchange = {"peers": st.peers, "bytes_to_each_peer": int(st.bytes_to_each_peer),
                              "bytes_received": int(st.bytes_received)}

    def all_gatherv(self, array):
        """concat of every rank's ``array`` in rank order, on every rank (== arrow_select::concat of the shards)."""
        ctx = self.ctx
        out, st, v = L.ArrayOut(), L.ExchangeStats(), array.view()
        ctx.check(ctx.lib.ah_all_gatherv(ctx.handle, self._h, C.byref(v), C.byref(out), C.byref(st)))
        self._note(st)
        return Array._from_out(ctx, out, array.data_type)

    def all_gather_record_batch_begin(self, batch):
        """Start the exchange of ``batch``'s columns (``ah_all_gather_columns_begin``: count exchange, outputs allocated,
        grouped send / recv and merge kernels enqueued) and return a handle whose ``end()`` yields the concatenated
        RecordBatch.  Between the two the host is free to drive OTHER contexts (streams): that is how a C host overlaps
        a take with the reassembly."""
        ctx, n = self.ctx, batch.num_columns()
        views = (L.ArrayView * n)()
        for i, c in enumerate(batch.columns):
            views[i] = c.view()
        h = C.c_void_p()
        ctx.check(ctx.lib.ah_all_gather_columns_begin(ctx.handle, self._h, n, views, C.byref(h)))
        return _PendingExchange(self, batch, views, h)

    def all_gather_record_batch(self, batch):
        """concat_batches of every rank's RecordBatch shard (one count exchange + one grouped exchange for all columns)."""
        return self.all_gather_record_batch_begin(batch).end()

    def all_gather_batches(self, batch, alignment=64):
        """Interface twin of ``Communicator.all_gather_batches``: here the result is already concatenated."""
        return [self.all_gather_record_batch(batch)]


class _PendingExchange:
    """An exchange between ``ah_all_gather_columns_begin`` and ``_end``; keeps the input batch alive."""

    def __init__(self, comm, batch, views, handle):
        self.comm, self.batch, self.views, self.handle = comm, batch, views, handle

    def end(self):
        from .array import RecordBatch
        comm, batch = self.comm, self.batch
        ctx, n = comm.ctx, batch.num_columns()
        if self.handle is None:
            raise RuntimeError("exchange already ended")
        outs = (L.ArrayOut * n)()
        st = L.ExchangeStats()
        h, self.handle = self.handle, None
        ctx.check(ctx.lib.ah_all_gather_columns_end(ctx.handle, comm._h, h, outs, C.byref(st)))
        comm._note(st)
        cols = []
        for i, c in enumerate(batch.columns):
            o = L.ArrayOut()
            C.memmove(C.byref(o), C.byref(outs[i]), C.sizeof(L.ArrayOut))
            cols.append(Array._from_out(ctx, o, c.data_type))
        self.batch = self.views = None
        return RecordBatch(batch.names, cols, num_rows=cols[0].length if cols else 0)
