#!/usr/bin/env python
"""bench.py — the hot-path benchmark (BASELINE.json metric: Mrows/s + achieved HBM GB/s,
filter->take on 1e9-row Int64 with 10% nulls, 1/2/4/8 GPUs).

A "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM: ``filter(values, predicate)`` followed by ``take(values, indices)`` through
the C ABI (libarrow_hip.so), exactly as an engine calling ``arrow::compute::kernels`` would.
Default workload = BASELINE.json configs[1]: Int64 x 1e9 rows, Bernoulli(0.9) validity,
Bernoulli(0.1) predicate, 1e8 uniform UInt32 take indices.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

The ONE JSON line of the default single-GPU run also carries, under "configs" (and "next_rows": SURVEY 8f), a few timed steps of
every other single-GPU configuration of BASELINE.json (configs[2]: add_wrapping and lt on 1e9
Float64 rows; configs[3]: Int64->Float64 and Float64->LargeUtf8 on 2^29 rows), each with its own
roofline object, so that every quoted roofline fraction is driver-run (--no-configs skips them).
"roofline.traffic" is measured in the same invocation: bench.py re-runs two steps of the workload
under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, corrections per
MI355X_MICROARCH.md §HBM; --pmc-traffic off skips it, and without rocprofv3 it is null).

Multi-GPU (weak scaling): every rank owns one 1e9-row shard of an N x 1e9-row column
(row-range sharding), filters/takes locally, then the filtered shard results are
reassembled on every rank with an all-gatherv over RCCL (north_star).  ``value`` includes the
reassembly; ``local_value`` is the same run's rate without it.

Other workloads: --workload arith|cmp|cast|cast_string|coalesce|string_filter_take|aggregate|sort|record_batch.
"""
import argparse
import csv
import ctypes as C
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable copy
# Every L2 miss is a 128-byte line fill (TCC_EA0_RDREQ_128B, profiles/r02_take_ablation.md); a random gather spends one
# fill per 8-byte value and one per validity bit.  The MEASURED maximum for exactly that mix is the `value_bit` probe of
# tools/gather_probe2.hip (the same accesses with nothing else in the kernel): 201.6 M fills in 3.819 ms = 52.8 G/s
# (the bitmap half is served by the Infinity Cache, which is why this is above 6.3 TB/s / 128 B = 49 G/s).
PROBE_MAX_FILLS_G = 52.8
ALL_WORKLOADS = ["filter_take", "arith", "cmp", "cast", "cast_string", "cast_string_utf8", "cast_chain", "coalesce", "string_filter_take", "string_filter",
                 "string_take", "predicate_filter", "predicate_filter_fused", "aggregate", "sort", "record_batch"]
EXTRA_CONFIGS = ["arith", "cmp", "cast", "cast_string", "cast_string_utf8", "cast_chain"]  # BASELINE configs[2] and [3], timed inside the default run
# SURVEY 8f rows 1 / 2 / 3 and configs[4]'s per-GPU shape, ditto
NEXT_ROWS = ["coalesce", "record_batch", "string_filter", "string_take", "predicate_filter", "predicate_filter_fused"]


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--rows", type=int, default=1_000_000_000, help="rows per GPU")
    p.add_argument("--selectivity", type=float, default=0.1)
    p.add_argument("--valid", type=float, default=0.9)
    p.add_argument("--workload", default="filter_take", choices=ALL_WORKLOADS)
    p.add_argument("--batch-rows", type=int, default=1 << 24, help="coalesce workload: rows per pushed batch")
    p.add_argument("--reassemble", default="auto", choices=["auto", "none", "allgatherv"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample-rows", type=int, default=1 << 27)  # SURVEY 8d / BASELINE.md 2: 2^27-2^28 rows
    p.add_argument("--no-configs", action="store_true", help="skip the configs[2]/[3] lines of the default run")
    p.add_argument("--only-narrow", action="store_true", help="print just the configs_narrow block (4-byte and narrower operands)")
    p.add_argument("--only-coalesce-sweep", default=None, nargs="?", const="all",
                   help="print just the coalesce_by_batch_rows block (optionally a comma-separated list of '<batch_rows>x<target>' keys)")
    p.add_argument("--config-steps", type=int, default=5)
    p.add_argument("--pmc-traffic", default="auto", choices=["auto", "off"])
    p.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the run rocprofv3 wraps
    # test hook (tests/test_gpu_comm.py): rank 0 writes the LAST reassembled column(s) of the headline workload to this
    # .npz so a checker can compare them with the oracle's un-sharded filter
    p.add_argument("--dump-gathered", default=None, help=argparse.SUPPRESS)
    p.add_argument("--detail-json", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                   help="where the full-detail object goes (stdout carries the compact line); '' = stderr only")
    p.add_argument("--transport", default="capi", choices=["capi", "torch"],
                   help="exchange step: ah_comm_* (libarrow_hip.so drives RCCL itself) or torch.distributed")
    return p.parse_args()


# ------------------------------------------------------------------------------------------ inputs
def mk_array(A, ctx, dt, n, vals, valid=None, nulls=0):
    R = A.array._RawMem
    return A.Array(ctx, dt, n, R(vals.ptr, vals.nbytes, vals), 0,
                   R(valid.ptr, valid.nbytes, valid) if valid is not None else None, 0, nulls)


def count_bits(ctx, buf, n):
    c = C.c_int64()
    ctx.check(ctx.lib.ah_count_set_bits(ctx.handle, buf.ptr, 0, n, C.byref(c)))
    return c.value


def gen_i64_column(A, ctx, n, seed, p_valid, row0, lo=-2**63, hi=2**63 - 1):
    lib, h = ctx.lib, ctx.handle
    vals = ctx.alloc(n * 8)
    valid = ctx.alloc(((n + 63) // 64) * 8)
    ctx.check(lib.ah_gen_uniform_i64(h, vals.ptr, n, seed, lo, hi, row0))
    ctx.check(lib.ah_gen_bernoulli_bits(h, valid.ptr, n, seed + 1, p_valid, row0))
    ctx.check(lib.ah_zero_null_slots(h, vals.ptr, 8, valid.ptr, n))
    return mk_array(A, ctx, A.Int64, n, vals, valid, n - count_bits(ctx, valid, n))


def gen_f64_column(A, ctx, n, seed, p_valid, row0):
    lib, h = ctx.lib, ctx.handle
    vals = ctx.alloc(n * 8)
    valid = ctx.alloc(((n + 63) // 64) * 8)
    ctx.check(lib.ah_gen_uniform_f64(h, vals.ptr, n, seed, -1e6, 1e6, row0))
    ctx.check(lib.ah_gen_bernoulli_bits(h, valid.ptr, n, seed + 1, p_valid, row0))
    ctx.check(lib.ah_zero_null_slots(h, vals.ptr, 8, valid.ptr, n))
    return mk_array(A, ctx, A.Float64, n, vals, valid, n - count_bits(ctx, valid, n))


def gen_cast_source(A, K, ctx, n, p_valid, row0):
    """Int64 uniform in [-1e6, 1e6] with 1 % of the rows full-range (so the >= 2^53 rounding of
    Int64 -> Float64 and the exponent forms of Float64 -> Utf8 are on the timed path), validity p_valid, null slots 0.
    Built from the plain counter-based generators so the oracle side can rebuild any chunk: rows where
    Bernoulli(seed 79, 0.01) is set take the full-range stream (seed 78), the others the bounded one (seed 42)."""
    base = gen_i64_column(A, ctx, n, 42, p_valid, row0, -10**6, 10**6)
    fv = ctx.alloc(n * 8)
    ctx.check(ctx.lib.ah_gen_uniform_i64(ctx.handle, fv.ptr, n, 78, -2**63, 2**63 - 1, row0))
    ctx.check(ctx.lib.ah_zero_null_slots(ctx.handle, fv.ptr, 8, base.validity.ptr, n))
    full = A.Array(ctx, A.Int64, n, A.array._RawMem(fv.ptr, fv.nbytes, fv), 0, base.validity, 0, base.null_count())
    pick = gen_predicate(A, ctx, n, 79, 0.01, row0)
    return K.zip(pick, full, base)


def gen_predicate(A, ctx, n, seed, p_true, row0):
    bits = ctx.alloc(((n + 63) // 64) * 8)
    ctx.check(ctx.lib.ah_gen_bernoulli_bits(ctx.handle, bits.ptr, n, seed, p_true, row0))
    return mk_array(A, ctx, A.Boolean, n, bits)


def gen_column(A, ctx, dt, n, seed, p_valid, row0=0):
    """A fixed-width column of any primitive width with Bernoulli validity, null slots zeroed (the reference's bench arrays,
    arrow/src/util/bench_util.rs:45-60): Int32 / Int16 / Int8 full-range bit patterns, Float32 uniform in [-1e6, 1e6)."""
    lib, h = ctx.lib, ctx.handle
    width = {A.Int8: 1, A.Int16: 2, A.Int32: 4, A.Float32: 4}[dt]
    vals = ctx.alloc(n * width)
    if dt is A.Int32:
        ctx.check(lib.ah_gen_uniform_i32(h, vals.ptr, n, seed, row0))
    elif dt is A.Float32:
        ctx.check(lib.ah_gen_uniform_f32(h, vals.ptr, n, seed, -1e6, 1e6, row0))
    else:
        ctx.check(lib.ah_gen_uniform_small(h, vals.ptr, width, n, seed, row0))
    if p_valid >= 1.0:
        return mk_array(A, ctx, dt, n, vals)
    valid = ctx.alloc(((n + 63) // 64) * 8)
    ctx.check(lib.ah_gen_bernoulli_bits(h, valid.ptr, n, seed + 1, p_valid, row0))
    ctx.check(lib.ah_zero_null_slots(h, vals.ptr, width, valid.ptr, n))
    return mk_array(A, ctx, dt, n, vals, valid, n - count_bits(ctx, valid, n))


# The reference's own criterion shapes are 4-byte and narrower (arrow/benches/filter_kernels.rs:39-45 i32/u8/f32,
# arithmetic_kernels.rs:26-33 f32, comparison_kernels.rs:33,83-84 f32/i32, cast_kernels.rs:34-40 i32->f64/i64, take_kernels.rs i32);
# BASELINE configs[0] is Int32.  Each entry: the same kernels on a narrow operand at a size that moves ~8-25 GB, with the
# SURVEY 8d accounting (every input buffer once in full, every output buffer once).
NARROW_CONFIGS = ["filter_i32", "take_i32", "add_wrapping_f32", "lt_f32", "lt_i32_scalar", "cast_i32_f64", "cast_i32_i64",
                  "filter_i16", "filter_i8"]


def build_narrow(env, name):
    A, K, ctx, args = env.A, env.K, env.ctx, env.args
    st = {}
    W = {"state": st}
    wb = lambda m: (m + 7) // 8  # noqa: E731
    if name.startswith("filter_"):
        dt, width, n = {"filter_i32": (A.Int32, 4, 2_000_000_000), "filter_i16": (A.Int16, 2, 4_000_000_000),
                        "filter_i8": (A.Int8, 1, 4_000_000_000)}[name]
        col = gen_column(A, ctx, dt, n, 42, args.valid)
        pred = gen_predicate(A, ctx, n, 44, args.selectivity, 0)

        def step(_r):
            f = K.filter(col, pred)
            st["k"], st["fn"] = f.length, f.null_count()
            return f

        def alg():
            k = st["k"]
            return n * width + 2 * wb(n) + k * width + (wb(k) if st["fn"] > 0 else 0)
        W.update(step=step, kernels=["filter_count", "filter_scatter"], dominant="filter_scatter", alg=alg,
                 text=f"filter {dt.name}, {n} rows, {args.valid:.0%} valid, {args.selectivity:.0%} selected (filter_kernels.rs:39-45)")
    elif name == "take_i32":
        n, nidx = 2_000_000_000, 100_000_000
        col = gen_column(A, ctx, A.Int32, n, 42, args.valid)
        ib = ctx.alloc(nidx * 4)
        ctx.check(ctx.lib.ah_gen_uniform_u32(ctx.handle, ib.ptr, nidx, 45, n, 0))
        idx = mk_array(A, ctx, A.UInt32, nidx, ib)
        W.update(step=lambda _r: K.take(col, idx), kernels=["take_gather"], dominant="take_gather",
                 alg=lambda: nidx * (4 + 4 + 4) + 2 * wb(nidx), rows=nidx,
                 text=f"take Int32 by {nidx} uniform UInt32 indices from {n} rows, {args.valid:.0%} valid (take_kernels.rs:32-80); "
                      "a random gather: bound by 128-byte line fills, not bytes")
    elif name in ("add_wrapping_f32", "lt_f32"):
        n = 2_000_000_000
        a = gen_column(A, ctx, A.Float32, n, 52, args.valid)
        b = gen_column(A, ctx, A.Float32, n, 62, args.valid)
        if name == "lt_f32":
            W.update(step=lambda _r: K.lt(a, b), kernels=["compare"], dominant="compare", alg=lambda: 8 * n + 4 * wb(n),
                     text=f"lt Float32 < Float32, {n} rows, both with NullBuffers (comparison_kernels.rs:33)")
        else:
            W.update(step=lambda _r: K.add_wrapping(a, b), kernels=["arith_binary"], dominant="arith_binary", alg=lambda: 12 * n + 3 * wb(n),
                     text=f"add_wrapping Float32 + Float32, {n} rows, both with NullBuffers (arithmetic_kernels.rs:26-33)")
    elif name == "lt_i32_scalar":
        n = 2_000_000_000
        a = gen_column(A, ctx, A.Int32, n, 42, args.valid)
        s0 = A.Scalar.new(0, A.Int32, ctx)
        W.update(step=lambda _r: K.lt(a, s0), kernels=["compare"], dominant="compare", alg=lambda: 4 * n + 3 * wb(n),
                 text=f"lt Int32 < scalar, {n} rows with NullBuffer (comparison_kernels.rs:83-84)")
    elif name in ("cast_i32_f64", "cast_i32_i64"):
        n = 2_000_000_000
        a = gen_column(A, ctx, A.Int32, n, 42, args.valid)
        to = A.Float64 if name == "cast_i32_f64" else A.Int64
        W.update(step=lambda _r: K.cast(a, to), kernels=["cast_numeric"], dominant="cast_numeric", alg=lambda: 12 * n + 2 * wb(n),
                 text=f"cast Int32 -> {to.name}, {n} rows with NullBuffer (cast_kernels.rs:34-40)")
    else:
        raise ValueError(name)
    W.setdefault("rows", n)
    W["n"] = n
    return W


def narrow_configs(env):
    """-> {name: {...}}: a few timed steps of every NARROW_CONFIGS entry, each with alg_bytes / avg_launch_ms / frac."""
    ctx, args = env.ctx, env.args
    res = {}
    only = os.environ.get("AH_NARROW_ONLY")  # A/B runs: a comma-separated subset
    for name in (NARROW_CONFIGS if not only else [x for x in NARROW_CONFIGS if x in only.split(",")]):
        try:
            ctx.lib.ah_pool_trim(ctx.handle)
            W = build_narrow(env, name)
            el, prof, out = run_timed(env, W, args.config_steps, 2, False, settle=True)
            ms = el / args.config_steps * 1e3
            dom_ms, dom_n = prof[W["dominant"]]
            rf = roofline_obj(W["dominant"], W["alg"](), dom_ms / max(dom_n, 1), dom_n)
            res[name] = {"workload": W["text"], "rows": W["n"], "steps": args.config_steps, "ms": round(ms, 4),
                         "value": round(W["rows"] / (ms * 1e-3) / 1e6, 1), "unit": "Mrows/s", "roofline": rf,
                         "kernel_avg_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()},
                         "host_gap_ms": round(ms - sum(v[0] for v in prof.values()) / args.config_steps, 4)}
            W = out = None
        except Exception as ex:  # noqa: BLE001 - never lose the headline to a secondary config
            res[name] = {"error": repr(ex)[:300]}
    ctx.lib.ah_pool_trim(ctx.handle)
    return res


# ------------------------------------------------------------------------------- BatchCoalescer by input batch size
# the reference's operating point is 8192-row input batches (arrow-select/src/coalesce.rs:172-173 "Typical
# values are 4096 or 8192 rows"; arrow/benches/coalesce_kernels.rs:34 batch_size = 8192), not the 2^24-row batches of the
# `coalesce` row.  The stream is the same 1e9 rows {Int64, Float64, both with NullBuffers} + 10 %-selected predicate, handed over
# as zero-copy slices of `batch_rows` rows; the host side is what an engine in C++ / Rust would do — the batch descriptors
# (ah_array_view) are prepared ahead, the timed loop is C-ABI calls only: grouped pushes (begin of group g + 1 before end of
# group g), bulk fetch + bulk release of the completed batches after every push.
COALESCE_SWEEP_BATCH_ROWS = [8192, 65536, 1 << 20, 1 << 24]


def _np_views(n):
    import numpy as np
    dt = np.dtype({"names": ["type", "length", "null_count", "values", "values_bit_offset", "validity", "validity_bit_offset", "offsets"],
                   "formats": ["<i4", "<i8", "<i8", "<u8", "<i8", "<u8", "<i8", "<u8"], "offsets": [0, 8, 16, 24, 32, 40, 48, 56],
                   "itemsize": 64})
    return np.zeros(n, dtype=dt)


class CoalesceStream:
    """`n` rows of (col_a, col_b, pred) as n / batch_rows pushes through ah_coalescer_push_batches_with_filters_begin / _end."""

    def __init__(self, env, cols, pred, n, batch_rows, target, group):
        import numpy as np
        from arrow_rs_amd import _lib as L
        self.env, self.lib, self.h = env, env.ctx.lib, env.ctx.handle
        self.ncols, self.target = len(cols), target
        nb = n // batch_rows
        self.nb, self.rows_streamed = nb, nb * batch_rows
        i = np.arange(nb, dtype=np.int64)
        cv = _np_views(nb * self.ncols).reshape(nb, self.ncols)
        for k, c in enumerate(cols):
            w = c.data_type.width
            cv["type"][:, k] = c.data_type.physical
            cv["length"][:, k] = batch_rows
            cv["null_count"][:, k] = -1
            cv["values"][:, k] = c.values.ptr + i * batch_rows * w
            cv["validity"][:, k] = c.validity.ptr
            cv["validity_bit_offset"][:, k] = i * batch_rows
        fv = _np_views(nb)
        fv["type"], fv["length"], fv["null_count"] = L.AH_BOOL, batch_rows, 0
        fv["values"], fv["values_bit_offset"] = pred.values.ptr, i * batch_rows
        self._cv, self._fv = np.ascontiguousarray(cv.reshape(-1)), fv
        self._rows = np.full(nb, batch_rows, dtype=np.int64)
        self._keep = (cols, pred)
        VP = C.POINTER(L.ArrayView)
        self.groups = []
        for g0 in range(0, nb, group):
            m = min(group, nb - g0)
            self.groups.append((m, C.cast(self._cv.ctypes.data + g0 * self.ncols * 64, VP), C.cast(self._rows.ctypes.data + g0 * 8, C.POINTER(C.c_int64)),
                                C.cast(self._fv.ctypes.data + g0 * 64, VP)))
        self.types = (C.c_int32 * self.ncols)(*[c.data_type.physical for c in cols])
        cap = int(group * batch_rows / max(target, 1)) + 8
        self.cap = cap
        self.outs = (L.ArrayOut * (cap * self.ncols))()
        self.out_rows = np.zeros(cap, dtype=np.int64)
        self.out_rows_p = C.cast(self.out_rows.ctypes.data, C.POINTER(C.c_int64))
        self.pushes = len(self.groups)

    def run(self):
        lib, h, ctx = self.lib, self.h, self.env.ctx
        co = C.c_void_p()
        ctx.check(lib.ah_coalescer_create(h, self.ncols, self.types, self.target, C.byref(co)))
        total = batches = 0
        n = C.c_int32()

        def drain(limit=1 << 30):
            nonlocal total, batches
            while limit > 0:
                ctx.check(lib.ah_coalescer_next_completed_batches(h, co, min(self.cap, limit), self.outs, self.out_rows_p, None, C.byref(n)))
                if n.value == 0:
                    return
                total += int(self.out_rows[:n.value].sum())
                batches += n.value
                limit -= n.value
                lib.ah_arrays_release(h, self.outs, n.value * self.ncols)
        try:
            pending = None
            for m, cv, rows, fv in self.groups:
                nxt = C.c_void_p()
                ctx.check(lib.ah_coalescer_push_batches_with_filters_begin(h, co, m, cv, rows, fv, None, C.byref(nxt)))
                if pending is not None:
                    # the batches finished by EARLIER pushes go downstream after this push has been enqueued: fetching a batch
                    # waits for its scatter and null counts, and those of the push just ended have only now been queued
                    ready = lib.ah_coalescer_completed_count(co)
                    ctx.check(lib.ah_coalescer_push_batches_with_filters_end(h, co, pending, None))
                    drain(ready)
                pending = nxt
            if pending is not None:
                ctx.check(lib.ah_coalescer_push_batches_with_filters_end(h, co, pending, None))
            ctx.check(lib.ah_coalescer_finish_buffered_batch(h, co))
            drain()
        finally:
            lib.ah_coalescer_destroy(h, co)
        return total, batches


def cpu_coalesce_1core(batch_rows, sample_rows, sel, valid):
    """One CPU core on the same stream shape: the oracle's filter of both columns, batch by batch (the reference's fused
    `copy_rows_by_filter_from` is a filter into the builder: coalesce/primitive.rs:95-140) -> Mrows/s."""
    import numpy as np
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))

    class View(C.Structure):
        _fields_ = [("type", C.c_int32), ("length", C.c_int64), ("null_count", C.c_int64), ("values", C.c_void_p),
                    ("values_bit_offset", C.c_int64), ("validity", C.c_void_p), ("validity_bit_offset", C.c_int64),
                    ("offsets", C.c_void_p)]

    class Out(C.Structure):
        _fields_ = [("type", C.c_int32), ("length", C.c_int64), ("null_count", C.c_int64), ("values", C.c_void_p),
                    ("values_bytes", C.c_int64), ("values_bit_offset", C.c_int64), ("validity", C.c_void_p),
                    ("validity_bytes", C.c_int64), ("validity_bit_offset", C.c_int64), ("offsets", C.c_void_p),
                    ("offsets_bytes", C.c_int64), ("flags", C.c_int32)]
    lib.orc_gen_uniform_i64.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_int64, C.c_int64, C.c_int64]
    lib.orc_gen_bernoulli_bits.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_int64]
    per = sample_rows
    v = np.empty(per, dtype=np.int64)
    vb = np.zeros(per // 8 + 8, dtype=np.uint8)
    mb = np.zeros(per // 8 + 8, dtype=np.uint8)
    lib.orc_gen_uniform_i64(v.ctypes.data, per, 42, -2**63, 2**63 - 1, 0)
    lib.orc_gen_bernoulli_bits(vb.ctypes.data, per, 43, valid, 0)
    lib.orc_gen_bernoulli_bits(mb.ctypes.data, per, 44, sel, 0)
    nb = per // batch_rows
    views = []
    for i in range(nb):
        vv, mv = View(), View()
        vv.type, vv.length, vv.null_count, vv.values, vv.validity, vv.validity_bit_offset = 5, batch_rows, -1, v.ctypes.data + i * batch_rows * 8, vb.ctypes.data, i * batch_rows
        mv.type, mv.length, mv.null_count, mv.values, mv.values_bit_offset = 1, batch_rows, 0, mb.ctypes.data, i * batch_rows
        views.append((vv, mv))

    def once():
        for vv, mv in views:
            for _col in range(2):  # two 8-byte columns share the predicate
                o = Out()
                lib.orc_filter(C.byref(vv), C.byref(mv), C.byref(o))
                lib.orc_release(C.byref(o))
    once()
    t0 = time.perf_counter()
    once()
    dt = time.perf_counter() - t0
    return nb * batch_rows / dt / 1e6


def coalesce_by_batch_rows(env, only=None):
    """-> {"<batch_rows>x<target>": {...}}: ms per 1e9 rows streamed, Mrows/s, frac (SURVEY 8d bytes of the `coalesce` row), pushes and
    launches per step, output batches, and the one-core CPU figure for the same input batch size."""
    A, K, ctx, args = env.A, env.K, env.ctx, env.args
    n = args.rows
    ctx.lib.ah_pool_trim(ctx.handle)
    col = gen_i64_column(A, ctx, n, 42, args.valid, 0)
    col2 = gen_f64_column(A, ctx, n, 52, args.valid, 0)
    pred = gen_predicate(A, ctx, n, 44, args.selectivity, 0)
    res = {}
    cpu = {}
    for br in COALESCE_SWEEP_BATCH_ROWS:
        if br > n:
            continue
        rule4 = max(64, int(br * args.selectivity * 4) // 64 * 64)
        for tname, target in (("8192", 8192), ("2^20", 1 << 20), ("4x", rule4)):
            key = f"{br}x{tname}"
            if only and key not in only:
                continue
            try:
                # batches per grouped push: ~2^27 rows' worth (at most 16 384 batches) — a push pays ~30 us of fixed GPU latency
                # (table upload, count, scan, read-back), 2^25-row pushes spent 0.9 of 5.5 ms per 1e9 rows on it at 8192-row batches
                gr = int(os.environ.get("AH_SWEEP_PUSH_ROWS", str(1 << 27)))
                group = max(2, min(16384, gr // br)) if br < (1 << 24) else 8
                stream = CoalesceStream(env, [col, col2], pred, n, br, target, group)
                stream.run()
                env.sync_all()
                times = []
                for _ in range(5):  # median of five whole streams (an occasional stream pays a pool miss: +2-3 ms)
                    t0 = time.perf_counter()
                    out_rows, out_batches = stream.run()
                    env.sync_all()
                    times.append((time.perf_counter() - t0) * 1e3)
                ms = sorted(times)[len(times) // 2]
                ctx.profile(True)
                ctx.profile_reset()
                stream.run()
                env.sync_all()
                prof = {k: ctx.profile_get(k) for k in ("filter_count", "filter_scatter", "copy_rows")}
                ctx.profile(False)
                rows = stream.rows_streamed
                alg = 2 * (rows * 8 + (rows + 7) // 8) + (rows + 7) // 8 + 2 * (out_rows * 8 + (out_rows + 7) // 8)
                res[key] = {"batch_rows": br, "target": target, "rows": rows, "ms": round(ms, 3), "value": round(rows / (ms * 1e-3) / 1e6, 1),
                            "unit": "Mrows/s", "alg_bytes": alg, "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                            "pushes": stream.pushes, "batches_per_push": group, "output_batches": out_batches, "output_rows": out_rows,
                            "kernel_launches": {k: v[1] for k, v in prof.items()},
                            "kernel_ms": {k: round(v[0], 3) for k, v in prof.items()}, "stream_ms_all": [round(t, 3) for t in times]}
                stream = None
            except Exception as ex:  # noqa: BLE001
                res[key] = {"error": repr(ex)[:300]}
        if not args.no_cpu_baseline and not only:
            try:
                cpu[str(br)] = round(cpu_coalesce_1core(br, min(1 << 25, n), args.selectivity, args.valid), 1)
            except Exception as ex:  # noqa: BLE001
                cpu[str(br)] = repr(ex)[:120]
    ids, _why = _granted_cores()
    out = {"points": res, "cpu_1core_Mrows_per_s_by_batch_rows": cpu, "cpu_cores_granted": len(ids),
           "what": "1e9 rows {Int64, Float64, NullBuffers}, 10 % selected, streamed as batch_rows-row pushes into BatchCoalescer(target): ms per "
                   "whole stream incl. fetching and releasing every output batch; C-ABI calls only in the timed loop (descriptors prepared ahead); "
                   "cpu = the oracle's filter of both columns batch by batch on one core (x cores_granted = a linear-scaling upper bound)"}
    col = col2 = pred = None
    ctx.lib.ah_pool_trim(ctx.handle)
    return out


# ------------------------------------------------------------------------------------ CPU baseline
def _host_threads_for(bytes_per_thread):
    """All host cores, unless the memory this container may use cannot hold one shard per core."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    avail = None
    try:
        for l in open("/proc/meminfo"):
            if l.startswith("MemAvailable:"):
                avail = int(l.split()[1]) * 1024
        for f in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
            if os.path.exists(f):
                t = open(f).read().strip()
                if t.isdigit():
                    avail = min(avail or int(t), int(t))
    except Exception:
        pass
    if avail:
        cores = max(1, min(cores, int(avail * 0.5 // bytes_per_thread)))
    return cores


def _granted_cores():
    """(core ids this process may run on, clipped by the cgroup CPU quota; what limited it)."""
    try:
        ids = sorted(os.sched_getaffinity(0))
    except Exception:
        ids = list(range(os.cpu_count() or 1))
    why = f"affinity mask: {len(ids)} cores"
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(f).read().split()
            if f.endswith("cpu.max"):
                quota, period = t[0], int(t[1])
            else:
                quota, period = t[0], int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = max(1, int(int(quota) / period))
                if q < len(ids):
                    ids, why = ids[:q], f"cgroup cpu quota: {q} cores of {len(ids)} in the affinity mask"
            break
        except Exception:
            continue
    return ids, why


def cpu_worker_main(argv):
    """`bench.py --cpu-worker <k> <core> <rows> <reps> <sel> <valid>`: one pinned process of the all-core baseline.
    Raw ctypes on oracle/liboracle.so (no HIP runtime in 256 processes).  Prints READY, waits for a line on stdin,
    runs, prints its elapsed seconds."""
    import numpy as np
    k, core, per, reps, sel, valid = int(argv[0]), int(argv[1]), int(argv[2]), int(argv[3]), float(argv[4]), float(argv[5])
    try:
        os.sched_setaffinity(0, {core})
    except Exception:
        pass
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))

    class View(C.Structure):
        _fields_ = [("type", C.c_int32), ("length", C.c_int64), ("null_count", C.c_int64), ("values", C.c_void_p),
                    ("values_bit_offset", C.c_int64), ("validity", C.c_void_p), ("validity_bit_offset", C.c_int64),
                    ("offsets", C.c_void_p)]

    class Out(C.Structure):
        _fields_ = [("type", C.c_int32), ("length", C.c_int64), ("null_count", C.c_int64), ("values", C.c_void_p),
                    ("values_bytes", C.c_int64), ("values_bit_offset", C.c_int64), ("validity", C.c_void_p),
                    ("validity_bytes", C.c_int64), ("validity_bit_offset", C.c_int64), ("offsets", C.c_void_p),
                    ("offsets_bytes", C.c_int64), ("flags", C.c_int32)]
    lib.orc_gen_uniform_i64.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_int64, C.c_int64, C.c_int64]
    lib.orc_gen_uniform_u32.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint32, C.c_int64]
    lib.orc_gen_bernoulli_bits.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_int64]
    nidx = int(per * sel)
    v = np.empty(per, dtype=np.int64)
    vb = np.zeros(per // 8 + 8, dtype=np.uint8)
    mb = np.zeros(per // 8 + 8, dtype=np.uint8)
    ix = np.empty(nidx, dtype=np.uint32)
    lib.orc_gen_uniform_i64(v.ctypes.data, per, 42, -2**63, 2**63 - 1, k * per)
    lib.orc_gen_bernoulli_bits(vb.ctypes.data, per, 43, valid, k * per)
    lib.orc_gen_bernoulli_bits(mb.ctypes.data, per, 44, sel, k * per)
    lib.orc_gen_uniform_u32(ix.ctypes.data, nidx, 45 + k, per, 0)
    AH_BOOL, AH_INT64, AH_UINT32 = 1, 5, 8
    vv, mv, iv = View(), View(), View()
    vv.type, vv.length, vv.null_count, vv.values, vv.validity = AH_INT64, per, -1, v.ctypes.data, vb.ctypes.data
    mv.type, mv.length, mv.null_count, mv.values = AH_BOOL, per, 0, mb.ctypes.data
    iv.type, iv.length, iv.null_count, iv.values = AH_UINT32, nidx, 0, ix.ctypes.data

    def once():
        o = Out()
        lib.orc_filter(C.byref(vv), C.byref(mv), C.byref(o))
        lib.orc_release(C.byref(o))
        o = Out()
        lib.orc_take(C.byref(vv), C.byref(iv), 0, C.byref(o))
        lib.orc_release(C.byref(o))
    once()  # heap warmed: the outputs of the timed calls reuse these chunks
    print("READY", flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    print(f"DONE {time.perf_counter() - t0:.6f}", flush=True)


def cpu_baseline_all_cores(args, one_core_mrows):
    per, reps = 1 << 24, 8
    ids, why = _granted_cores()
    T = min(len(ids), _host_threads_for(per * 14))
    if T < len(ids):
        why += f"; host memory holds {T} shards of {per} rows"
    env = dict(os.environ, MALLOC_MMAP_THRESHOLD_=str(1 << 30), MALLOC_TRIM_THRESHOLD_=str(1 << 34), MALLOC_TOP_PAD_=str(64 << 20),
               OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(k), str(ids[k]), str(per), str(reps),
                               str(args.selectivity), str(args.valid)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, text=True, env=env) for k in range(T)]
    try:
        for p in procs:
            if p.stdout.readline().strip() != "READY":
                raise RuntimeError("a CPU baseline worker did not start")
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        each = [float(p.stdout.readline().split()[1]) for p in procs]
        wall = time.perf_counter() - t0
    finally:
        for p in procs:
            try:
                p.stdin.close()
            except Exception:
                pass
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()
    nidx = int(per * args.selectivity)
    mrows = per * T * reps / wall / 1e6
    # bytes the port moves per (filter + take): values + 2 bitmaps in, K values + bits out; idx + gathered values in, values + bits out
    k_sel = int(per * args.selectivity)
    alg = per * 8 + 2 * (per // 8) + k_sel * 8 + k_sel // 8 + nidx * (4 + 8 + 8) + 2 * (nidx // 8)
    out = {"value": round(mrows, 1), "unit": "Mrows/s", "cores": T, "processes": T, "granted": why,
           "speedup_over_one_core": round(mrows / one_core_mrows, 1) if one_core_mrows else None,
           "algorithmic_GBps": round(alg * T * reps / wall / 1e9, 1),
           "slowest_worker_s": round(max(each), 3), "fastest_worker_s": round(min(each), 3), "wall_s": round(wall, 3),
           "sample": f"{reps} x (filter + take), {T} pinned processes x {per} rows (128 MiB of values each, "
                     f"{per * T * 8 / 2**30:.0f} GiB in all: DRAM-resident), {nidx} u32 indices per process, outputs reused from the heap"}
    if one_core_mrows and mrows < 20 * one_core_mrows:
        ratio = mrows / one_core_mrows
        if T < 20:
            out["limit"] = (f"{T} cores are all this container is granted ({why}): {ratio:.1f}x one core = {100 * ratio / T:.0f} % per-core "
                            f"efficiency ({out['algorithmic_GBps']} GB/s algorithmic; the random 8-byte gathers of take fetch whole "
                            "lines on top); 20x one core is out of reach by construction")
        else:
            out["limit"] = (f"{T} processes reach {ratio:.1f}x one core: the port is DRAM-bound ({out['algorithmic_GBps']} GB/s "
                            "algorithmic, random 8-byte gathers fetch whole lines on top), not core-bound")
    return out


def cpu_baseline_filter_take(args):
    """The oracle (a scalar port of the reference's algorithm) timed on the GPU box's host cores over a
    bounded sample of the same workload.  Reported baseline, never the thing shipped."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import arrow_rs_amd as A
    import orc
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        return None
    oracle = orc.load(so)
    n = args.cpu_sample_rows
    vals = oracle.gen_i64(n, 42, -2**63, 2**63 - 1)
    valid = oracle.gen_bits(n, 43, args.valid)
    mask = oracle.gen_bits(n, 44, args.selectivity)
    vals[~valid] = 0
    idx = oracle.gen_u32(max(1, int(n * args.selectivity)), 45, n)
    hv = orc.HostArray(A.Int64, vals, valid)
    hm = orc.HostArray(A.Boolean, mask)
    hi = orc.HostArray(A.UInt32, idx)
    # pre-pack once (packing is harness cost, not kernel cost)
    hvh, hmh, hih = orc._Held(hv), orc._Held(hm), orc._Held(hi)
    reps, t_total = 0, 0.0
    while t_total < 8.0 and reps < 50:
        out = orc.Out()
        t0 = time.perf_counter()
        st = oracle.lib.orc_filter(C.byref(hvh.view), C.byref(hmh.view), C.byref(out))
        t1 = time.perf_counter()
        assert st == 0
        oracle.lib.orc_release(C.byref(out))
        out = orc.Out()
        t2 = time.perf_counter()
        st = oracle.lib.orc_take(C.byref(hvh.view), C.byref(hih.view), 0, C.byref(out))
        t3 = time.perf_counter()
        assert st == 0
        oracle.lib.orc_release(C.byref(out))
        t_total += (t1 - t0) + (t3 - t2)
        reps += 1
    mrows = n * reps / t_total / 1e6
    res = {"value": round(mrows, 2), "unit": "Mrows/s", "cores": 1, "kind": "port",
           "sample": f"{reps} x (filter + take) on {n} Int64 rows, {int(n * args.selectivity)} u32 indices, "
                     f"same generators/densities; oracle/liboracle.so single thread; "
                     f"host has {os.cpu_count()} cores"}
    del hvh, hmh, hih, hv, hm, hi, vals, valid, mask, idx
    # secondary: the same port row-sharded over EVERY host core this container is granted (how an engine parallelises
    # the single-threaded reference kernels): ONE PROCESS per core, pinned, 2^24 rows = 128 MiB of values each so the
    # shards stream from DRAM, glibc told to keep the 13 MB outputs on its heap instead of mmap/munmap-ing them per call
    # (256 threads of one process fought over the address-space lock and reported 4x one core).
    try:
        res["all_cores"] = cpu_baseline_all_cores(args, mrows)
    except Exception as ex:  # noqa: BLE001
        res["all_cores"] = {"error": repr(ex)}
    # independent sanity line: Arrow C++ through pyarrow on the same sample.  A different
    # implementation (filter/take semantics coincide) — NOT the reference and not the baseline.
    try:
        import pyarrow as pa
        import pyarrow.compute as pc
        vals = oracle.gen_i64(n, 42, -2**63, 2**63 - 1)
        valid = oracle.gen_bits(n, 43, args.valid)
        mask = oracle.gen_bits(n, 44, args.selectivity)
        idx = oracle.gen_u32(max(1, int(n * args.selectivity)), 45, n)
        pv = pa.array(vals, mask=~valid)
        pm = pa.array(mask)
        pi = pa.array(idx)
        pc.take(pc.filter(pv, pm), pa.array([0], type=pa.uint32()))  # warm
        reps2, t2 = 0, 0.0
        while t2 < 3.0 and reps2 < 20:
            t0 = time.perf_counter()
            pc.filter(pv, pm)
            pc.take(pv, pi)
            t2 += time.perf_counter() - t0
            reps2 += 1
        res["arrow_cpp_sanity"] = {"value": round(n * reps2 / t2 / 1e6, 2), "unit": "Mrows/s",
                                   "what": f"pyarrow {pa.__version__} pc.filter + pc.take on the same {n}-row sample: "
                                           "Arrow C++, not the reference", "threads": pa.cpu_count()}
    except Exception as ex:
        res["arrow_cpp_sanity"] = {"error": repr(ex)}
    return res


def crossover_rows(env):
    """Smallest batch size (rows) at which one synchronous C-ABI `ah_filter` call (Int64, 10 % nulls, 10 % selectivity,
    inputs resident in HBM) is faster than the 1-core oracle on the same rows: below it a batch is better left on the
    CPU (SURVEY section 7 'batch-size mismatch').  -> dict(rows, gpu_us, cpu_us per probed size)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import arrow_rs_amd as A
    from arrow_rs_amd import _lib as L
    import orc
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        return None
    oracle = orc.load(so)
    ctx, lib, h = env.ctx, env.ctx.lib, env.ctx.handle
    probes, cross = {}, None
    for n in (4096, 8192, 16384, 32768, 65536, 131072, 262144, 1048576):
        col = gen_i64_column(A, ctx, n, 42, 0.9, 0)
        pred = gen_predicate(A, ctx, n, 44, 0.1, 0)
        vc, vp = col.view(), pred.view()

        def call():
            out = L.ArrayOut()
            assert lib.ah_filter(h, C.byref(vc), C.byref(vp), C.byref(out)) == 0
            lib.ah_array_release(h, C.byref(out))
        for _ in range(5):
            call()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            call()
        ctx.synchronize()
        gpu_us = (time.perf_counter() - t0) / 200 * 1e6
        vals = oracle.gen_i64(n, 42, -2**63, 2**63 - 1)
        hv = orc._Held(orc.HostArray(A.Int64, vals, oracle.gen_bits(n, 43, 0.9)))
        hm = orc._Held(orc.HostArray(A.Boolean, oracle.gen_bits(n, 44, 0.1)))
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.05:
            o = orc.Out()
            oracle.lib.orc_filter(C.byref(hv.view), C.byref(hm.view), C.byref(o))
            oracle.lib.orc_release(C.byref(o))
            k += 1
        cpu_us = (time.perf_counter() - t0) / k * 1e6
        probes[str(n)] = {"gpu_us": round(gpu_us, 1), "cpu_1core_us": round(cpu_us, 1)}
        if cross is None and gpu_us < cpu_us:
            cross = n
    return {"rows": cross, "probes": probes,
            "what": "first probed size at which one synchronous C-ABI ah_filter call (Int64, 10 % nulls, 10 % selected) beats the "
                    "1-core oracle on the same rows"}


def reference_bench_shapes(env, batch=64):
    """The drop-in at the reference's OWN criterion shapes: 65 536-row filters at the three
    selectivities of arrow/benches/filter_kernels.rs:39-45,96-120 (with a fresh predicate and with a prebuilt
    FilterPredicate, without and with 50 % nulls), take of 512 / 1 024 Int32 rows (take_kernels.rs:32-80), add_wrapping
    and lt on 65 536 Float32 (arithmetic_kernels.rs:26-33, comparison_kernels.rs:33).  Per shape, through the raw C ABI:
      sync_us     one synchronous call (result complete and null_count known at return) — the reference's contract;
      batched_us  the same call amortised over `batch` calls handed over together: predicates for the whole batch built
                  with ONE wait (ah_filter_predicates_build), every apply / arith / compare only ENQUEUED (deferred mode),
                  ONE ah_synchronize at the end — what an engine that has several batches queued can do;
      cpu_1core_us the oracle (scalar port of the reference's kernel) on one host core, same data.
    take (round 5) has the deferred form too: the out-of-bounds panic is then raised by the next ah_synchronize."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import arrow_rs_amd as A
    from arrow_rs_amd import _lib as L
    import orc
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        return None
    oracle = orc.load(so)
    ctx, lib, h = env.ctx, env.ctx.lib, env.ctx.handle
    rng = np.random.default_rng(42)
    n = 65536

    def timed(fn, reps):
        for _ in range(3):
            fn()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    def cpu_timed(fn):
        fn()
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.05:
            fn()
            k += 1
        return (time.perf_counter() - t0) / k * 1e6

    def release(out):
        lib.ah_array_release(h, C.byref(out))

    def graph_timed(record_one, replays=40):
        """`batch` calls RECORDED into one hipGraph (ah_graph_begin / _end), then replayed: us per call.  None if the
        capture is refused (a shape whose call must wait on the device)."""
        outs = [L.ArrayOut() for _ in range(batch)]
        g = C.c_void_p()
        if lib.ah_graph_begin(h) != 0:
            return None
        ok = all(record_one(outs[i]) == 0 for i in range(batch))
        st = lib.ah_graph_end(h, C.byref(g))
        if not ok or st != 0:
            if st == 0:
                lib.ah_graph_destroy(h, g)
            for o in outs:
                release(o)
            return None
        try:
            for _ in range(3):
                assert lib.ah_graph_launch(h, g) == 0
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(replays):
                assert lib.ah_graph_launch(h, g) == 0
            ctx.synchronize()
            return (time.perf_counter() - t0) / (replays * batch) * 1e6
        finally:
            lib.ah_graph_destroy(h, g)
            for o in outs:
                release(o)

    shapes = {}

    def row(name, rows, sync_us, batched_us, cpu_us, note=None, graph_us=None):
        best = min(x for x in (batched_us, graph_us) if x is not None) if (batched_us is not None or graph_us is not None) else None
        r = {"rows": rows, "sync_us": round(sync_us, 2), "batched_us": None if batched_us is None else round(batched_us, 2),
             "graph_us": None if graph_us is None else round(graph_us, 2), "cpu_1core_us": round(cpu_us, 2),
             "sync_beats_cpu": bool(sync_us < cpu_us), "batched_beats_cpu": None if best is None else bool(best < cpu_us)}
        if note:
            r["note"] = note
        shapes[name] = r

    # ---- filter (filter_kernels.rs): Int32, three selectivities; fresh predicate (no nulls) and prebuilt predicate (50 % nulls)
    sels = (("kept 1/2", 0.5), ("high selectivity (kept 1023/1024)", 1.0 - 1.0 / 1024.0), ("low selectivity (kept 1/1024)", 1.0 / 1024.0))
    for nulls, pv in (("", None), (" w NULLs", 0.5)):
        hv = orc.HostArray(A.Int32, rng.integers(-2**31, 2**31 - 1, n, dtype=np.int32), None if pv is None else rng.random(n) < pv)
        dv = hv.to_device(ctx)
        vv = dv.view()
        for label, sel in sels:
            hm = orc.HostArray(A.Boolean, rng.random(n) < sel)
            dm = hm.to_device(ctx)
            mv = dm.view()
            hvh, hmh = orc._Held(hv), orc._Held(hm)

            def cpu():
                o = orc.Out()
                oracle.lib.orc_filter(C.byref(hvh.view), C.byref(hmh.view), C.byref(o))
                oracle.lib.orc_release(C.byref(o))
            cpu_us = cpu_timed(cpu)
            if pv is None:  # "filter i32 (...)": FilterBuilder::new + filter per call
                def sync():
                    out = L.ArrayOut()
                    assert lib.ah_filter(h, C.byref(vv), C.byref(mv), C.byref(out)) == 0
                    release(out)
                masks = (L.ArrayView * batch)(*[mv] * batch)

                def batched():
                    preds = (C.c_void_p * batch)()
                    assert lib.ah_filter_predicates_build(h, batch, masks, preds) == 0  # ONE wait for `batch` counts
                    lib.ah_context_set_deferred(h, 1)
                    try:
                        outs = [L.ArrayOut() for _ in range(batch)]
                        for i in range(batch):
                            assert lib.ah_filter_predicate_apply(h, preds[i], C.byref(vv), C.byref(outs[i])) == 0
                    finally:
                        lib.ah_context_set_deferred(h, 0)
                    assert lib.ah_synchronize(h) == 0
                    for i in range(batch):
                        release(outs[i])
                        lib.ah_filter_predicate_free(h, preds[i])
                row(f"filter i32 ({label})" if label.startswith("kept") else f"filter i32 {label}", n, timed(sync, 200),
                    timed(batched, 8) / batch, cpu_us)
            else:  # "filter context i32 w NULLs (...)": the FilterPredicate is built once, outside the timed call
                pred = C.c_void_p()
                assert lib.ah_filter_predicate_build(h, C.byref(mv), C.byref(pred)) == 0

                def sync():
                    out = L.ArrayOut()
                    assert lib.ah_filter_predicate_apply(h, pred, C.byref(vv), C.byref(out)) == 0
                    release(out)

                def batched():
                    lib.ah_context_set_deferred(h, 1)
                    try:
                        outs = [L.ArrayOut() for _ in range(batch)]
                        for i in range(batch):
                            assert lib.ah_filter_predicate_apply(h, pred, C.byref(vv), C.byref(outs[i])) == 0
                    finally:
                        lib.ah_context_set_deferred(h, 0)
                    assert lib.ah_synchronize(h) == 0
                    for o in outs:
                        release(o)
                gus = graph_timed(lambda o: lib.ah_filter_predicate_apply(h, pred, C.byref(vv), C.byref(o)))
                row(f"filter context i32 w NULLs ({label})" if label.startswith("kept") else f"filter context i32 w NULLs {label}", n,
                    timed(sync, 200), timed(batched, 8) / batch, cpu_us,
                    "cpu_1core_us includes the oracle's predicate build (it has no prebuilt-predicate entry point)", graph_us=gus)
                lib.ah_filter_predicate_free(h, pred)

    # ---- take (take_kernels.rs): Int32 values, random u32 indices, no nulls
    for m in (512, 1024):
        hv = orc.HostArray(A.Int32, rng.integers(-2**31, 2**31 - 1, m, dtype=np.int32))
        hi = orc.HostArray(A.UInt32, rng.integers(0, m, m).astype(np.uint32))
        dv, di = hv.to_device(ctx), hi.to_device(ctx)
        vv, iv = dv.view(), di.view()
        hvh, hih = orc._Held(hv), orc._Held(hi)

        def sync():
            out = L.ArrayOut()
            assert lib.ah_take(h, C.byref(vv), C.byref(iv), 0, C.byref(out)) == 0
            release(out)

        def cpu():
            o = orc.Out()
            oracle.lib.orc_take(C.byref(hvh.view), C.byref(hih.view), 0, C.byref(o))
            oracle.lib.orc_release(C.byref(o))
        def batched():  # deferred mode (round 5): the gathers are only enqueued; an out-of-bounds index would be raised by the synchronize
            lib.ah_context_set_deferred(h, 1)
            try:
                outs = [L.ArrayOut() for _ in range(batch)]
                for i in range(batch):
                    assert lib.ah_take(h, C.byref(vv), C.byref(iv), 0, C.byref(outs[i])) == 0
            finally:
                lib.ah_context_set_deferred(h, 0)
            assert lib.ah_synchronize(h) == 0
            for o in outs:
                release(o)
        gus = graph_timed(lambda o: lib.ah_take(h, C.byref(vv), C.byref(iv), 0, C.byref(o)))
        row(f"take i32 {m}", m, timed(sync, 200), timed(batched, 8) / batch, cpu_timed(cpu),
            "sync: an out-of-bounds index (the reference's panic) is reported at return; batched / graph: deferred mode, the panic "
            "is raised by the next ah_synchronize", graph_us=gus)

    # ---- add_wrapping / lt on 65 536 Float32, no nulls (arithmetic_kernels.rs add(0), comparison_kernels.rs lt)
    ha = orc.HostArray(A.Float32, rng.random(n, dtype=np.float32))
    hb = orc.HostArray(A.Float32, rng.random(n, dtype=np.float32))
    da, db = ha.to_device(ctx), hb.to_device(ctx)
    av, bv = da.view(), db.view()
    hah, hbh = orc._Held(ha), orc._Held(hb)
    for name, fn_name, op, ofn in (("add(0) f32", "ah_arith_binary", 1, "orc_arith"), ("lt f32", "ah_compare", 2, "orc_compare")):
        fn = getattr(lib, fn_name)

        def sync():
            out = L.ArrayOut()
            assert fn(h, op, C.byref(av), 0, C.byref(bv), 0, C.byref(out)) == 0
            release(out)

        def batched():
            lib.ah_context_set_deferred(h, 1)
            try:
                outs = [L.ArrayOut() for _ in range(batch)]
                for i in range(batch):
                    assert fn(h, op, C.byref(av), 0, C.byref(bv), 0, C.byref(outs[i])) == 0
            finally:
                lib.ah_context_set_deferred(h, 0)
            assert lib.ah_synchronize(h) == 0
            for o in outs:
                release(o)

        def cpu():
            o = orc.Out()
            getattr(oracle.lib, ofn)(op, C.byref(hah.view), 0, C.byref(hbh.view), 0, C.byref(o))
            oracle.lib.orc_release(C.byref(o))
        gus = graph_timed(lambda o: fn(h, op, C.byref(av), 0, C.byref(bv), 0, C.byref(o)))
        row(name, n, timed(sync, 200), timed(batched, 8) / batch, cpu_timed(cpu), graph_us=gus)
    lost = [k for k, v in shapes.items() if v["batched_beats_cpu"] is False or (v["batched_beats_cpu"] is None and not v["sync_beats_cpu"])]
    return {"batch": batch, "shapes": shapes, "shapes_where_one_cpu_core_wins": lost,
            "what": "the reference's own criterion shapes through the raw C ABI: sync_us = one synchronous call; batched_us = per call "
                    "inside a batch of %d (predicates built with one wait, applies / arith / compare enqueued in deferred mode, one "
                    "ah_synchronize); graph_us = per call when those %d deferred calls are RECORDED once into a hipGraph "
                    "(ah_graph_begin / _end) and replayed (shapes whose output size is fixed: a prebuilt predicate, add, lt); "
                    "cpu_1core_us = the oracle on one host core.  A one-call-at-a-time drop-in loses to one core at "
                    "these sizes; the batched / graph forms are what an engine with several batches queued gets" % (batch, batch)}


# ------------------------------------------------------------------------------- in-run PMC traffic
# kernel-name patterns of the kernels a workload's roofline objects talk about, and how FETCH_SIZE is corrected.
# This rocprofv3 computes FETCH_SIZE = TCC_EA0_RDREQ x 64 B, but on gfx950 the L2's read requests are 128-byte line
# fills (MI355X_MICROARCH.md §HBM: double it).  That holds for the wide streaming kernels AND — calibrated in
# profiles/r02_take_ablation.md with TCC_EA0_RDREQ_128B — for the random gather: 1.03e8 (value) + 1.00e8 (validity
# byte) requests per 1e8 indices, every one a 128-byte request.  So FETCH_SIZE is doubled for every kernel here.
# WRITE_SIZE is calibrated 1:1 (gen_i64 writes 8.0e9 B and reports 7 812 500 KB).
PMC_KERNELS = {
    "filter_take": {"take_gather": (r"take_kernel<", 2.0), "filter_scatter": (r"filter_scatter_kernel<8", 2.0),
                    "filter_count": (r"filter_count_kernel", 2.0)},
    "arith": {"arith_binary": (r"arith_kernel<", 2.0)},
    "cmp": {"compare": (r"compare_kernel<", 2.0)},
    "cast": {"cast_numeric": (r"cast_(stream_)?kernel<", 2.0)},
    "cast_string": {"cast_string_len": (r"string_len_kernel<", 2.0), "cast_string_write": (r"string_write_kernel<", 2.0)},
}


# workloads whose roofline object covers ALL launches of one step: every dispatch whose name matches is summed and
# divided by the steps the child ran (setup kernels — generators, the casts that build the input — do not match)
PMC_STEP_KERNELS = {
    "string_filter": r"filter_(scatter|count|count_small|group_scan|finish)\w*_kernel|"
                     r"string_filter_\w+_kernel|chained_scan_kernel",
    "string_take": r"chained_scan_kernel|take_gather_rows_kernel|take_ranges_kernel",
    "string_filter_take": r"filter_(scatter|count|count_small|group_scan|finish)\w*_kernel|string_filter_\w+_kernel|"
                          r"chained_scan_kernel|take_gather_rows_kernel|take_ranges_kernel",
    "coalesce": r"filter_(scatter|count|count_small|group_scan|finish|finish_acc)\w*_kernel|copy_rows\w*_kernel|bm_acc_kernel|coalesce_finish_kernel",
    "predicate_filter_fused": r"filter_expr_\w+_kernel|filter_(scatter|group_scan|finish)\w*_kernel",
    "predicate_filter": r"compare_kernel|bitmap_op_kernel|filter_(scatter|count|count_small|group_scan|finish)\w*_kernel|popcount_partial_kernel",
}
PMC_CHILD_STEPS, PMC_CHILD_WARMUP = 2, 1


def pmc_traffic_inrun(args, wl):
    """HBM bytes per launch of the workload's kernels, measured NOW: two extra runs of two steps each under
    rocprofv3 (FETCH_SIZE and WRITE_SIZE in separate passes, never combined with other tracing)."""
    if args.pmc_traffic == "off" or (wl not in PMC_KERNELS and wl not in PMC_STEP_KERNELS) or shutil.which("rocprofv3") is None:
        return None
    pats = PMC_KERNELS.get(wl, {})
    step_pat = PMC_STEP_KERNELS.get(wl)
    got = {k: {} for k in pats}
    step_sum = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ah_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", wl, "--steps", str(PMC_CHILD_STEPS),
               "--warmup", str(PMC_CHILD_WARMUP),
               "--rows", str(args.rows), "--selectivity", str(args.selectivity), "--valid", str(args.valid)]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=150, check=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            for f in files:
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name", counter) != counter:
                        continue
                    for k, (pat, _) in pats.items():
                        if re.search(pat, r["Kernel_Name"]):
                            got[k].setdefault(counter, []).append(float(r["Counter_Value"]))
                    if step_pat and re.search(step_pat, r["Kernel_Name"]):
                        step_sum[counter] = step_sum.get(counter, 0.0) + float(r["Counter_Value"])
        except Exception as ex:  # noqa: BLE001 - a profiler hiccup must never cost the bench line
            return {"error": repr(ex)[:200]}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {}
    for k, (pat, fetch_scale) in pats.items():
        f, w = got[k].get("FETCH_SIZE"), got[k].get("WRITE_SIZE")
        if f and w:
            out[k] = int(sum(f) / len(f) * 1024 * fetch_scale + sum(w) / len(w) * 1024)
    if step_pat and "FETCH_SIZE" in step_sum and "WRITE_SIZE" in step_sum:
        child_steps = PMC_CHILD_STEPS + PMC_CHILD_WARMUP
        out["step_total"] = int((step_sum["FETCH_SIZE"] * 2.0 + step_sum["WRITE_SIZE"]) * 1024 / child_steps)
    return out or None


# --------------------------------------------------------------------------------------- workloads
class Env:
    pass


def build_workload(env, wl):
    """-> dict(step, kernels, dominant, state, n, ...) for one workload; inputs are generated in HBM here."""
    A, K, ctx, args = env.A, env.K, env.ctx, env.args
    n, row0, rank = args.rows, env.rank * args.rows, env.rank
    st = {}
    W = {"state": st, "n": n}
    if wl == "filter_take":
        col = gen_i64_column(A, ctx, n, 42, args.valid, row0)
        pred = gen_predicate(A, ctx, n, 44, args.selectivity, row0)
        nidx = max(1, int(n * args.selectivity))
        ib = ctx.alloc(nidx * 4)
        ctx.check(ctx.lib.ah_gen_uniform_u32(ctx.handle, ib.ptr, nidx, 45 + rank, n if n < 2**32 else 0, 0))
        idx = mk_array(A, ctx, A.UInt32, nidx, ib)
        W["idx"], W["col"], W["pred"] = idx, col, pred
        # the take is independent of the filter output, so at N>1 it runs on a SECOND context (own HIP
        # stream, own thread; ctypes drops the GIL) while the filtered shard is being all-gathered over
        # RCCL: collective and compute overlap on separate streams
        ctx2 = A.Context(env.local_rank) if env.comm is not None else None

        def rebind(a, c):
            return A.Array(c, a.data_type, a.length, a.values, a.values_bit_offset, a.validity,
                           a.validity_bit_offset, a.null_count(), a.offsets)

        col_b, idx_b = (rebind(col, ctx2), rebind(idx, ctx2)) if ctx2 is not None else (None, None)

        def step(with_reassembly):
            f = K.filter(col, pred)
            if with_reassembly and hasattr(env.comm, "all_gather_record_batch_begin"):
                # the C-host form: start the exchange on this context's stream (ah_all_gather_columns_begin), run the
                # take on the second context = second stream while RCCL moves the shards, come back for the result
                pending = env.comm.all_gather_record_batch_begin(A.RecordBatch(["f"], [f], f.length))
                t = K.take(col_b, idx_b)
                g = pending.end()
                st["gk"] = g.num_rows()
                if args.dump_gathered:
                    st["gathered"] = g
            elif with_reassembly and not st.get("reassemble_error"):
                import threading
                box = {}

                def side():
                    try:
                        box["t"] = K.take(col_b, idx_b)
                    except Exception as ex:  # noqa: BLE001
                        box["err"] = ex
                th = threading.Thread(target=side)
                th.start()
                try:
                    g = env.comm.all_gatherv(f)
                    st["gk"] = g.length
                except Exception as ex:  # keep the run alive: report local-only numbers + the error
                    st["reassemble_error"] = repr(ex)[:300]
                th.join()
                if "err" in box:
                    raise box["err"]
                t = box["t"]
            else:
                t = K.take(col, idx)
            st["k"], st["fn"], st["tn"] = f.length, f.null_count(), t.null_count()
            return f, t

        W.update(step=step, kernels=["filter_count", "filter_scatter", "take_gather"], dominant="filter_scatter")
    elif wl == "coalesce":
        # BatchCoalescer::push_batch_with_filter over a stream of batches —
        # the filter scatters straight into the in-progress output batch (no intermediate array)
        col = gen_i64_column(A, ctx, n, 42, args.valid, row0)
        col2 = gen_f64_column(A, ctx, n, 52, args.valid, row0)
        pred = gen_predicate(A, ctx, n, 44, args.selectivity, row0)
        br = min(args.batch_rows, n)
        nb = n // br
        # (AH_COALESCE_TARGET_BATCHES: output batch size in pushed batches' worth of selected rows; experiments only)
        target = max(1, int(br * args.selectivity * float(os.environ.get("AH_COALESCE_TARGET_BATCHES", "4"))))
        if target >= 64 and os.environ.get("AH_COALESCE_TARGET_ANY") != "1":
            target = target // 64 * 64  # whole validity words per output batch: what the slab push (csrc/coalesce.hip) takes
        batches = [(A.RecordBatch(["a", "b"], [col.slice(i * br, br), col2.slice(i * br, br)]), pred.slice(i * br, br))
                   for i in range(nb)]

        # how many batches the engine hands over per call: 1 = push_batch_with_filter per batch; > 1 = the grouped push
        # (ah_coalescer_push_batches_with_filters: one count read-back for the group, the batches of one output window
        # scattered by one launch).  The default run reports the grouped form (8) and, beside it, the single-push time.
        group = max(1, int(os.environ.get("AH_COALESCE_GROUP", "8")))
        pipelined = os.environ.get("AH_COALESCE_PIPELINE", "1") != "0"
        W["group"], W["pipelined"] = group, pipelined

        def step(_r):
            co = K.BatchCoalescer.new(["a", "b"], [A.Int64, A.Float64], target, ctx)
            out_rows = 0
            pending = None
            for i in range(0, len(batches), group):
                # batches finished by EARLIER pushes are handed downstream after this push has been enqueued: fetching a
                # batch waits for its scatters, and a push's own count read-back has already waited for everything the
                # earlier pushes enqueued — so the fetch costs nothing and the GPU has this push's work queued meanwhile
                ready = co.completed_count()
                if pipelined:
                    # the engine holds the NEXT group already: its count passes are enqueued before this group is appended
                    # (ah_coalescer_push_batches_with_filters_begin / _end), so no count round trip leaves the GPU idle
                    nxt = co.push_batches_with_filters_begin(batches[i:i + group])
                    if pending is not None:
                        pending.end()
                    pending = nxt
                elif group == 1:
                    co.push_batch_with_filter(*batches[i])
                else:
                    co.push_batches_with_filters(batches[i:i + group])
                for _ in range(ready):
                    out_rows += co.next_completed_batch().num_rows()
            if pending is not None:
                pending.end()
            co.finish_buffered_batch()
            while co.has_completed_batch():
                out_rows += co.next_completed_batch().num_rows()
            st["out_rows"] = out_rows
            return out_rows

        W.update(step=step, kernels=["filter_count", "filter_scatter", "copy_rows"], dominant="filter_scatter")
    elif wl in ("string_filter_take", "string_filter", "string_take"):
        # LargeUtf8 column (the config-4 cast output) through filter and take — as two rows, because
        # the two halves are different machines: the filter is order-preserving (contiguous selected runs stream), the
        # take is a random row gather (bound by line fills)
        n = min(n, 1 << 29) if args.rows == 1_000_000_000 else n  # (2^29 rows: the size of the cast that produces the column, configs[3])
        src = gen_i64_column(A, ctx, n, 42, args.valid, row0, -10**6, 10**6)
        scol = K.cast(K.cast(src, A.Float64), A.LargeUtf8)
        st["text_bytes"] = scol.values.nbytes
        src = None
        pred = gen_predicate(A, ctx, n, 44, args.selectivity, row0)
        nidx = max(1, int(n * args.selectivity))
        ib = ctx.alloc(nidx * 4)
        ctx.check(ctx.lib.ah_gen_uniform_u32(ctx.handle, ib.ptr, nidx, 45, n, 0))
        idx = mk_array(A, ctx, A.UInt32, nidx, ib)
        W["idx"] = idx

        do_f, do_t = wl != "string_take", wl != "string_filter"

        def step(_r):
            f = K.filter(scol, pred) if do_f else None
            t = K.take(scol, idx) if do_t else None
            st["fbytes"], st["tbytes"] = (f.values.nbytes if do_f else 0), (t.values.nbytes if do_t else 0)
            st["k"] = f.length if do_f else 0
            return f, t

        W.update(step=step, dominant="string_gather_bytes",
                 kernels=(["filter_count", "filter_scatter", "string_filter_ranges"] if do_f else []) +
                         ["string_ranges_scan", "string_gather_bytes"] + (["string_take_ranges", "take_gather"] if do_t else []))
    elif wl == "predicate_filter":
        # the predicate is BUILT on the device and consumed by filter without leaving it —
        # WHERE a < 0 AND b >= 0: lt(a, scalar), gt_eq(b, scalar), and_kleene, filter(a, mask with nulls)
        col = gen_i64_column(A, ctx, n, 42, args.valid, row0)
        colb = gen_f64_column(A, ctx, n, 52, args.valid, row0)
        s0 = A.Scalar.new(0, A.Int64, ctx)
        s1 = A.Scalar.new(0.0, A.Float64, ctx)

        def step(_r):
            m = K.and_kleene(K.lt(col, s0), K.gt_eq(colb, s1))
            f = K.filter(col, m)
            st["k"], st["fn"] = f.length, f.null_count()
            return f

        W.update(step=step, kernels=["compare", "boolean", "filter_count", "filter_scatter"], dominant="filter_scatter")
    elif wl == "predicate_filter_fused":
        # the same WHERE a < 0 AND b >= 0 handed over as TERMS (ah_filter_predicate_build_expr): the comparisons become
        # ballots inside the filter's count pass; nothing but the 1-bit-per-row selection is materialised
        col = gen_i64_column(A, ctx, n, 42, args.valid, row0)
        colb = gen_f64_column(A, ctx, n, 52, args.valid, row0)
        s0 = A.Scalar.new(0, A.Int64, ctx)
        s1 = A.Scalar.new(0.0, A.Float64, ctx)

        def step(_r):
            f = K.filter_expr(col, [("lt", col, s0), ("gt_eq", colb, s1)], ["and_kleene"])
            st["k"], st["fn"] = f.length, f.null_count()
            return f

        W.update(step=step, kernels=["filter_expr_count", "filter_scatter"], dominant="filter_expr_count")
    elif wl == "aggregate":
        # sum + min + max of the Int64 column (three streaming reads per step)
        col = gen_i64_column(A, ctx, n, 42, args.valid, row0)
        G = K.aggregate
        W.update(step=lambda _r: (G.sum(col), G.min(col), G.max(col)), kernels=["aggregate"], dominant="aggregate")
    elif wl == "record_batch":
        # BASELINE configs[4] shape: RecordBatch {Int64, Float64, each with validity} + one mask per shard;
        # filter_record_batch (one count pass, one two-column scatter launch), then at N>1 ONE exchange of both columns
        cola = gen_i64_column(A, ctx, n, 42, args.valid, row0)
        colb = gen_f64_column(A, ctx, n, 52, args.valid, row0)
        pred = gen_predicate(A, ctx, n, 44, args.selectivity, row0)
        rb = A.RecordBatch(["a", "b"], [cola, colb], n)

        def step(with_reassembly):
            f = K.filter_record_batch(rb, pred)
            st["k"] = f.num_rows()
            if with_reassembly:
                parts = env.comm.all_gather_batches(f)
                st["gk"] = sum(p.num_rows() for p in parts)
                if args.dump_gathered and len(parts) == 1:
                    st["gathered"] = parts[0]
            return f

        W.update(step=step, kernels=["filter_count", "filter_scatter"], dominant="filter_scatter")
    elif wl == "sort":
        # the producer of take's indices: sort_to_indices of the full-range Int64 column (8 radix passes)
        n = min(n, 1 << 29) if args.rows == 1_000_000_000 else n
        col = gen_i64_column(A, ctx, n, 42, args.valid, row0)
        W["col"] = col
        W.update(step=lambda _r: K.sort_to_indices(col), kernels=["sort_radix_pass"], dominant="sort_radix_pass")
    elif wl in ("arith", "cmp"):
        a = gen_f64_column(A, ctx, n, 52, args.valid, row0)
        b = gen_f64_column(A, ctx, n, 62, args.valid, row0)
        fn = K.add_wrapping if wl == "arith" else K.lt
        k = "arith_binary" if wl == "arith" else "compare"
        W.update(step=lambda _r: fn(a, b), kernels=[k], dominant=k)
    else:
        n = min(n, 1 << 29) if args.rows == 1_000_000_000 else n
        if os.environ.get("AH_BENCH_CAST_PURE") == "1":  # ablation: no full-range rows (round 1's input)
            src = gen_i64_column(A, ctx, n, 42, args.valid, row0, -10**6, 10**6)
        else:
            src = gen_cast_source(A, K, ctx, n, args.valid, row0)
        if wl == "cast":
            W.update(step=lambda _r: K.cast(src, A.Float64), kernels=["cast_numeric"])
        elif wl == "cast_chain":
            # configs[3] as ONE call (ah_cast_chain): text straight from the Int64 column, the Float64 array is never built
            W.update(step=lambda _r: K.cast_chain(src, [A.Float64, A.LargeUtf8]), kernels=["cast_string_len", "cast_string_write"])
        elif wl == "cast_string_utf8":
            # SURVEY 8d cfg 4: "Utf8 in 64 Mi-row batches to show i32 behaviour" — the same 2^29 rows, 8 casts per step
            f64 = K.cast(src, A.Float64)
            br = min(n, 1 << 26)
            parts = [f64.slice(i * br, min(br, n - i * br)) for i in range((n + br - 1) // br)]

            def step(_r):
                outs = [K.cast(p_, A.Utf8) for p_ in parts]
                st["out_bytes"] = sum(o.values.nbytes for o in outs)
                st["batches"] = len(outs)
                return outs[-1]
            W.update(step=step, kernels=["cast_string_len", "cast_string_write"])
        else:
            f64 = K.cast(src, A.Float64)
            W.update(step=lambda _r: K.cast(f64, A.LargeUtf8), kernels=["cast_string_len", "cast_string_write"])
        W["dominant"] = W["kernels"][-1]
    W["n"] = n
    return W


def describe(env, wl, W, prof, out, steps):
    """-> (kernel the roofline object is about, its avg ms, its launches, algorithmic bytes per launch, workload text,
    metric, dtype) — SURVEY §8d: every input buffer read once in full + every output buffer written once."""
    args, n, st = env.args, W["n"], W["state"]
    dominant = W["dominant"]
    dom_ms, dom_n = prof[dominant]
    dom_avg = dom_ms / max(dom_n, 1)
    if wl == "filter_take":
        k = st["k"]
        alg = n * 8 + 2 * ((n + 7) // 8) + k * 8 + (((k + 7) // 8) if st["fn"] > 0 else 0)
        text = (f"configs[1]: filter()+take() on {n}-row Int64 per GPU, {args.valid:.0%} valid, "
                f"{args.selectivity:.0%} selectivity, {W['idx'].length} uniform UInt32 take indices")
        return dominant, dom_avg, dom_n, alg, text, "filter_take_Mrows_per_s", "int64"
    per_row = {"arith": 24.375, "cmp": 16.5, "cast": 16.25}.get(wl)
    kernels = W["kernels"]
    if wl in ("string_filter_take", "string_filter", "string_take"):
        k, idx = st["k"], W["idx"]
        # filter: offsets + validity + mask in, K+1 offsets + K validity bits + bytes out (selected bytes read once, written once);
        # take: idx + one offsets pair and one validity bit per index in, offsets + bits + bytes out
        alg = 0
        if wl != "string_take":
            alg += (n + 1) * 8 + 2 * ((n + 7) // 8) + (k + 1) * 8 + (k + 7) // 8 + 2 * st["fbytes"]
        if wl != "string_filter":
            alg += idx.length * (4 + 16 + 8) + 2 * ((idx.length + 7) // 8) + 2 * st["tbytes"]
        dom_avg, dom_n = sum(prof[kk][0] for kk in kernels) / max(steps, 1), steps
        dominant = f"{wl}_step"
        if wl == "string_filter":
            # SURVEY 8d prices "every input buffer read once in full": the WHOLE text buffer, not only the selected rows' bytes
            # (whose 128-byte lines the kernel has to fetch almost all of anyway: ~14 rows per line, 10 % selected)
            W["alg_survey_rule"] = alg - st["fbytes"] + st["text_bytes"]
    elif wl == "coalesce":
        k = st["out_rows"]
        # two columns share one predicate: 2 x (values + validity) + mask in, 2 x (K values + K bits) out
        alg = 2 * (n * 8 + (n + 7) // 8) + (n + 7) // 8 + 2 * (k * 8 + (k + 7) // 8)
        dom_avg, dom_n = sum(prof[kk][0] for kk in kernels) / max(steps, 1), steps  # all launches of one step
        dominant = "coalesce_step"
    elif wl == "predicate_filter":
        k = st["k"]
        wb = (n + 7) // 8
        # two scalar compares (values + validity in, bits + validity out), and_kleene (4 bitmaps in, 2 out),
        # filter (values + validity + mask + mask validity in, K values + K bits out)
        alg = 2 * (n * 8 + wb + 2 * wb) + 6 * wb + (n * 8 + 3 * wb + k * 8 + (k + 7) // 8)
        dom_avg, dom_n = sum(prof[kk][0] for kk in kernels) / max(steps, 1), steps  # all launches of one step
        dominant = "predicate_filter_step"
    elif wl == "predicate_filter_fused":
        k = st["k"]
        wb = (n + 7) // 8
        # every input once (a, b values + their validity), every output once (K values + K bits).  The
        # selection words and the second pass's re-read of `a` are the implementation's own traffic (r03's first
        # collection counted the two-pass plan's 26.3 GB here, which flattered the fraction: 0.68 instead of 0.47).
        alg = 2 * n * 8 + 2 * wb + k * 8 + (k + 7) // 8
        dom_avg, dom_n = sum(prof[kk][0] for kk in kernels) / max(steps, 1), steps
        dominant = "predicate_filter_fused_step"
    elif wl == "record_batch":
        k = st["k"]  # ONE scatter launch for both columns: 2 x (values + validity) + the mask in, 2 x (K values + K bits) out
        alg = 2 * (n * 8 + (n + 7) // 8) + (n + 7) // 8 + 2 * (k * 8 + (k + 7) // 8)
    elif wl == "sort":
        alg = (n - W["col"].null_count()) * 32  # per radix pass: keys read twice, (key, index) pairs written once
    elif wl == "aggregate":
        alg = n * 8 + (n + 7) // 8  # per launch: values + validity in, 8 bytes out
    elif wl == "cast_string_utf8":  # all batches of one step: i32 offsets
        alg = n * 8 + (n + 7) // 8 + (n + st["batches"]) * 4 + st["out_bytes"] + (n + 7) // 8
        dom_avg, dom_n = sum(prof[k][0] for k in kernels) / max(steps, 1), steps
        dominant = "cast_string_utf8_step"
    elif per_row is None:  # cast_string / cast_chain: input + validity in, offsets + bytes + validity out; both passes of one cast
        alg = n * 8 + (n + 7) // 8 + (n + 1) * 8 + out.values.nbytes + (n + 7) // 8
        dom_avg = sum(prof[k][0] for k in kernels) / max(prof[kernels[0]][1], 1)
        dominant = "cast_string_len+cast_string_write"
    else:
        alg = int(per_row * n)
    text = {"arith": "configs[2]: add_wrapping Float64+Float64 with NullBuffers",
            "cmp": "configs[2]: lt Float64<Float64 with NullBuffers",
            "cast": "configs[3]: cast Int64->Float64 (values in [-1e6, 1e6], 1 % full-range)",
            "cast_string": "configs[3]: cast Float64->LargeUtf8 (integer-valued doubles in [-1e6, 1e6], 1 % full-range; parity beyond the reference's five literals is pinned to "
                           "the oracle's restated ryu, unpinned by the reference itself)",
            "record_batch": "configs[4] shape: filter_record_batch on {Int64, Float64} with NullBuffers",
            "sort": "arrow_ord sort_to_indices of a full-range Int64 column with NullBuffer (stable LSD radix)",
            "aggregate": "SURVEY 8f-4: sum + min + max of an Int64 column with NullBuffer",
            "string_filter_take": "SURVEY 8f-3: filter + take on a LargeUtf8 column (cast output)",
            "string_filter": "SURVEY 8f-3: filter on a LargeUtf8 column (cast output): order-preserving, selected runs stream",
            "string_take": "SURVEY 8f-3: take with uniform random UInt32 indices on a LargeUtf8 column (cast output): one row gather per index",
            "cast_string_utf8": "configs[3]: cast Float64->Utf8 (i32 offsets) in 64 Mi-row batches of the same column",
            "cast_chain": "configs[3] as ONE call: ah_cast_chain(Int64 -> Float64 -> LargeUtf8), text formatted straight from the Int64 column (no Float64 "
                          "array); compare with configs.cast.ms + configs.cast_string.ms for the two separate calls",
            "predicate_filter_fused": "SURVEY 8f-2: the same WHERE a < 0 AND b >= 0 handed over as terms (ah_filter_expr): the comparisons are ballots inside the filter's count pass, only the 1-bit-per-row selection is materialised",
            "predicate_filter": "SURVEY 8f-2: filter(a, and_kleene(lt(a, 0), gt_eq(b, 0.0))) on Int64 a, Float64 b with NullBuffers",
            "coalesce": f"SURVEY 8f-1: BatchCoalescer.push_batch_with_filter, Int64+Float64, "
                        f"{args.batch_rows}-row batches"
                        + (f", pushed {W.get('group', 1)} at a time (ah_coalescer_push_batches_with_filters"
                           + ("_begin / _end: the next group's counts are enqueued before this group is appended)" if W.get("pipelined") else ")")
                           if W.get("group", 1) > 1 else ", one push_batch_with_filter per batch"
                           + (" (pipelined: begin of batch i + 1 before end of batch i)" if W.get("pipelined") else ""))}[wl] + f", {n} rows per GPU"
    dtype = "int64" if wl in ("aggregate", "sort") else "int64+f64" if wl == "record_batch" else "f64"
    return dominant, dom_avg, dom_n, alg, text, f"{wl}_Mrows_per_s", dtype


def roofline_obj(kernel, alg, avg_ms, launches, traffic=None):
    ach = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    r = {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic, "algorithmic_bytes_per_launch": alg,
         "avg_launch_ms": round(avg_ms, 4), "launches": launches}
    if traffic and avg_ms > 0:
        # what the memory system moved for this kernel (PMC bytes per launch / measured launch time)
        r["traffic_GBps"] = round(traffic / (avg_ms * 1e-3) / 1e9, 1)
        r["traffic_frac"] = round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
    return r


def run_timed(env, W, steps, warmup, reassemble, settle=False):
    ctx = env.ctx
    step = W["step"]
    out = None
    for _ in range(warmup):
        out = step(reassemble)
    if settle:
        # The configs[2..] / 8f lines run a few steps each, behind CPU-side phases of this same process (oracle timings) that
        # leave the GPU idle for seconds, and on freshly allocated buffers (the pool is trimmed between configs): one
        # collection showed the FIRST config after such a phase at 1.7x its usual time for all of its five steps.  Extra
        # untimed steps until two in a row agree within 3 % (at most 12): clocks ramped, pages touched.
        prev, W["settle_ms"] = None, []
        for i in range(12):
            env.sync_all()
            t0 = time.perf_counter()
            out = step(reassemble)
            env.sync_all()
            dt = time.perf_counter() - t0
            W["settle_ms"].append(round(dt * 1e3, 3))
            if prev is not None and abs(dt - prev) <= 0.03 * prev:
                break
            prev = dt
    ctx.profile(True)
    ctx.profile_reset()
    env.sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step(reassemble)
    env.sync_all()
    elapsed = time.perf_counter() - t0
    prof = {k: ctx.profile_get(k) for k in W["kernels"]}
    ctx.profile(False)
    return elapsed, prof, out


def _compact_roofline(rf):
    if not isinstance(rf, dict):
        return rf
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "launches",
            "traffic_GBps", "traffic_frac")
    return {k: rf[k] for k in keep if k in rf}


def _compact_config(c):
    """One per-config entry of the compact line: what the judge recomputes fractions from, nothing else."""
    if not isinstance(c, dict) or "error" in c:
        return c
    out = {k: c[k] for k in ("rows", "rows_per_gpu", "ms", "value", "local_ms", "local_value", "gathered_rows", "host_gap_ms",
                              "ms_without_kernel_events") if k in c}
    if "roofline" in c:
        rf = c["roofline"]
        out.update(kernel=rf.get("kernel"), avg_launch_ms=rf.get("avg_launch_ms"), alg_bytes=rf.get("algorithmic_bytes_per_launch"),
                   frac=rf.get("frac"))
        if "frac_survey_rule" in rf:
            out.update(alg_bytes_survey_rule=rf.get("algorithmic_bytes_survey_rule"), frac_survey_rule=rf.get("frac_survey_rule"))
    for k in ("single_push", "single_push_pipelined", "grouped_not_pipelined"):
        if isinstance(c.get(k), dict):
            out[k + "_ms"] = c[k].get("ms_without_kernel_events")
    if isinstance(c.get("exchange"), dict):
        out["exchange"] = {k: c["exchange"][k] for k in ("peers", "bytes_to_each_peer", "total_ms", "per_link_GBps") if k in c["exchange"]}
    return out


def emit(line, args):
    """stdout gets ONE compact JSON line, LAST: the contract keys, `roofline`, `cpu_baseline` and a few numbers per config
    (enough to recompute every fraction) — small enough that a log tail keeps all of it (r03's single 14 KB line lost
    `configs.arith` off the front of the driver's stdout tail).  The full-detail object (every roofline object, notes,
    probes) goes to stderr as one JSON line and to --detail-json (default gpurun_out/bench_detail.json)."""
    detail = json.dumps(line)
    path = args.detail_json
    try:
        if path:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "w") as f:
                f.write(detail + "\n")
    except OSError:
        path = None
    print("BENCH_DETAIL " + detail, file=sys.stderr, flush=True)
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "kernel_avg_ms", "host_gap_ms", "local_value", "local_ms_per_step", "exchange", "reassemble_last_ms",
            "step_algorithmic_GBps", "hbm_pool", "filter_selected_rows", "take_indices", "take_sorted_indices_ms", "take_null_indices_ms",
            "pmc_traffic_bytes_per_launch")
    compact = {k: line[k] for k in keep if k in line}
    compact["roofline"] = _compact_roofline(line.get("roofline"))
    if "requests" in (line.get("roofline") or {}):
        rq = line["roofline"]["requests"]
        compact["roofline"]["requests"] = {k: rq[k] for k in ("per_launch", "achieved_G_per_s", "probe_max_G_per_s", "frac_of_probe_max") if k in rq}
    if "roofline_filter_scatter" in line:
        compact["roofline_filter_scatter"] = _compact_roofline(line["roofline_filter_scatter"])
    # every per-config roofline of this invocation INSIDE `roofline` (the driver's record keeps that object whole and only the
    # names of the other keys: VERDICT r05 item 3): {frac, avg_launch_ms, alg_bytes} per configuration (kernel names: `configs`)
    by_config = {}

    def _bc(name, rf):
        if isinstance(rf, dict) and "frac" in rf:
            by_config[name] = {"frac": rf.get("frac"), "avg_launch_ms": rf.get("avg_launch_ms"),
                               "alg_bytes": rf.get("algorithmic_bytes_per_launch")}
            if "frac_survey_rule" in rf:
                by_config[name]["frac_survey_rule"] = rf["frac_survey_rule"]
    _bc("filter", line.get("roofline_filter_scatter"))
    if (line.get("roofline") or {}).get("kernel") == "take_gather":
        _bc("take", line["roofline"])
    for grp in ("configs", "next_rows", "configs_narrow"):
        for k, v in (line.get(grp) or {}).items():
            if isinstance(v, dict):
                _bc(k, v.get("roofline"))
    if by_config and isinstance(compact["roofline"], dict):
        compact["roofline"]["by_config"] = by_config
    for grp in ("configs", "next_rows", "configs_narrow"):
        if grp in line:
            compact[grp] = {k: _compact_config(v) for k, v in line[grp].items()}
    if isinstance(line.get("filter_by_selectivity"), dict):
        compact["filter_scatter_ms_by_selectivity"] = {k: (v.get("filter_scatter_ms") if isinstance(v, dict) else v)
                                                       for k, v in line["filter_by_selectivity"].items()}
    cs = line.get("coalesce_by_batch_rows")
    if isinstance(cs, dict) and "points" in cs:
        compact["coalesce_by_batch_rows"] = {
            "columns": ["ms", "Mrows_per_s", "frac", "pushes", "output_batches", "launches"],
            "points": {k: ([v.get("ms"), v.get("value"), v.get("frac"), v.get("pushes"), v.get("output_batches"),
                            sum((v.get("kernel_launches") or {}).values())] if "error" not in v else v) for k, v in cs["points"].items()},
            "alg_bytes_per_stream": next((v.get("alg_bytes") for v in cs["points"].values() if "alg_bytes" in v), None),
            "cpu_1core_Mrows_per_s_by_batch_rows": cs.get("cpu_1core_Mrows_per_s_by_batch_rows"), "cpu_cores_granted": cs.get("cpu_cores_granted")}
    elif cs is not None:
        compact["coalesce_by_batch_rows"] = cs
    rs = line.get("reference_bench_shapes")
    if isinstance(rs, dict) and "shapes" in rs:
        compact["reference_bench_shapes"] = {
            "batch": rs["batch"], "columns": ["sync_us", "batched_us", "graph_us", "cpu_1core_us"],
            "shapes": {k: [v["sync_us"], v["batched_us"], v.get("graph_us"), v["cpu_1core_us"]] for k, v in rs["shapes"].items()},
            "one_cpu_core_wins": rs["shapes_where_one_cpu_core_wins"]}
    elif rs is not None:
        compact["reference_bench_shapes"] = rs
    cr = line.get("crossover_rows")
    if isinstance(cr, dict):
        compact["crossover_rows"] = cr.get("rows", cr)
    cb = line.get("cpu_baseline")
    if isinstance(cb, dict):
        compact["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "error") if k in cb}
        if isinstance(cb.get("all_cores"), dict):
            compact["cpu_baseline"]["all_cores"] = {k: cb["all_cores"][k] for k in ("value", "cores", "unit") if k in cb["all_cores"]}
    compact["detail"] = "stderr line 'BENCH_DETAIL {...}'" + (f" and {path}" if path else "")
    print(json.dumps(compact), flush=True)


# ------------------------------------------------------------------------------------------ transport probe (N > 1)
PROBE_ROWS = 1 << 22  # rows per rank in the probe exchange (Int64 + Float64 with validity: ~68 MB per rank)


def probe_child_main(argv):
    """`python bench.py --probe-transport capi|torch`: ONE exchange of a {Int64, Float64, validity} shard per rank over
    the transport named, in a process of its own.  The parent rank (below) runs it BEFORE it touches the transport: a
    transport that hangs or aborts on this node (first contact with real RCCL over xGMI happens on the driver's scaling
    run) takes this child down, not the benchmark — the parent then switches transports or reports shard-local numbers.
    Exit 0 + 'PROBE_OK' = the concatenation arrived with the right row count on this rank."""
    transport = argv[0] if argv else "capi"
    if os.environ.get("AH_BENCH_PROBE_TEST_SLEEP"):  # test hook (tests/test_bench_probe_cpu.py): a child that never answers
        time.sleep(float(os.environ["AH_BENCH_PROBE_TEST_SLEEP"]))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = 0 if os.environ.get("AH_BENCH_SHARED_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("AH_BENCH_BACKEND", "nccl")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import arrow_rs_amd as A
    from arrow_rs_amd import distributed as D
    ctx = A.Context(local_rank)
    A.set_default_context(ctx)
    n = PROBE_ROWS + 4096 * rank  # ragged on purpose: the counts differ per rank
    a = gen_i64_column(A, ctx, n, 7, 0.9, rank * PROBE_ROWS)
    b = gen_f64_column(A, ctx, n, 9, 0.9, rank * PROBE_ROWS)
    total = sum(PROBE_ROWS + 4096 * r for r in range(world))
    if transport == "capi":
        def share_id(payload):
            box = [payload]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        comm = D.CApiCommunicator(ctx, rank, world, share_id)
        g = comm.all_gather_record_batch(A.RecordBatch(["a", "b"], [a, b], n))
        rows, nulls = g.num_rows(), [c.null_count() for c in g.columns]
        comm.barrier()
    else:
        grp = dist.new_group(backend="nccl") if backend == "nccl" else None
        comm = D.Communicator(ctx, dist, group=grp)
        ga, gb = comm.all_gatherv(a), comm.all_gatherv(b)
        rows, nulls = ga.length, [ga.null_count(), gb.null_count()]
        if gb.length != rows:
            raise SystemExit(f"probe: columns disagree ({rows} vs {gb.length} rows)")
        dist.barrier()
    ctx.synchronize()
    if rows != total:
        raise SystemExit(f"probe: {rows} rows arrived, {total} expected")
    # every rank must see the same NULL counts (the bitmaps travelled and were merged at the right bit offsets)
    seen = [None] * world
    dist.all_gather_object(seen, nulls)
    if any(x != seen[0] for x in seen):
        raise SystemExit(f"probe: ranks disagree on the gathered NULL counts: {seen}")
    print("PROBE_OK", flush=True)
    sys.stdout.flush()
    os._exit(0)  # no destructors: a transport that tears down badly must not turn a good probe into a bad one


def probe_transport(dist, rank, world, transport):
    """Run the probe child of every rank (collectively) and agree on the verdict.  -> (ok, reason)"""
    import tempfile
    box = [_free_port() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    timeout = float(os.environ.get("AH_BENCH_PROBE_TIMEOUT", "150"))
    # (under torch.distributed.run the ranks rendezvous through the AGENT's store — TORCHELASTIC_USE_AGENT_STORE — which
    # does not listen on the probe's port: without those variables the child of rank 0 hosts the store itself)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(MASTER_PORT=str(box[0]), AH_BENCH_PROBE_CHILD="1")
    log = tempfile.NamedTemporaryFile(prefix=f"ah_probe_{transport}_{rank}_", suffix=".log", delete=False)
    t0 = time.perf_counter()
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--probe-transport", transport], env=env,
                         stdout=log, stderr=subprocess.STDOUT)
    why = None
    try:
        rc = p.wait(timeout=timeout)
        if rc != 0:
            why = f"rank {rank}: probe exited {rc}"
    except subprocess.TimeoutExpired:
        p.kill()  # (this exact child)
        p.wait()
        why = f"rank {rank}: no answer within {timeout:.0f} s (hung)"
    log.close()
    try:
        text = open(log.name, errors="replace").read()
        os.unlink(log.name)
    except OSError:
        text = ""
    if why is None and "PROBE_OK" not in text:
        why = f"rank {rank}: probe ended without PROBE_OK"
    if why is not None:
        tail = " | ".join(l.strip() for l in text.strip().splitlines()[-3:])
        why = (why + (": " + tail if tail else ""))[:300]
    verdicts = [None] * world
    dist.all_gather_object(verdicts, why)
    bad = [v for v in verdicts if v]
    return (not bad), (bad[0] if bad else f"ok in {time.perf_counter() - t0:.1f} s")


class ShardLocalOnly:
    """What is left when no transport passed the probe: barriers and the MAX over ranks through the CPU process group;
    no exchange (the line says so).  Never chosen silently: `config.transport_note` carries both probes' verdicts."""

    def __init__(self, dist, world):
        self.dist, self.world = dist, world
        self.timings, self.last_exchange = None, None


def _free_port():
    import socket
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    return port


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: spawn N ranks HERE (one per visible GPU, rendezvous on
    127.0.0.1), relay rank 0's ONE JSON line.  Fewer than N GPUs visible: say so in the line and exit non-zero —
    a 1-GPU number is never reported as n_gpus N."""
    import arrow_rs_amd as A
    visible = int(A._lib.load().ah_device_count())
    shared = os.environ.get("AH_BENCH_SHARED_GPU") == "1"
    if visible < args.gpus and not shared:
        print(json.dumps({"metric": "filter_take_Mrows_per_s", "value": None, "unit": "Mrows/s", "n_gpus": visible,
                          "requested_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "data": "synthetic",
                          "error": f"--gpus {args.gpus} requested but only {visible} GPU(s) are visible to this process "
                                   "(hipGetDeviceCount); refusing to report a multi-GPU number from fewer devices"}), flush=True)
        return 3
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(args.gpus), AH_BENCH_SPAWNED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        if shared:
            env.setdefault("GLOO_SOCKET_IFNAME", "lo")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = procs[0].communicate()[0].decode()
    rc = procs[0].returncode
    for p in procs[1:]:
        try:
            rc = rc or p.wait(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            rc = rc or 4
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    for l in out0.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return rc if rc else (0 if lines else 5)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker_main(sys.argv[2:])
    if len(sys.argv) > 1 and sys.argv[1] == "--probe-transport":
        return probe_child_main(sys.argv[2:])
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.pmc_child:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (tests/test_gpu_parity.py::test_bench_two_ranks_one_gpu): every rank on GPU 0 and a transport
    # that allows it.  The driver never sets these: one rank per GPU over RCCL ("nccl") is the product path.
    backend = os.environ.get("AH_BENCH_BACKEND", "nccl")
    if os.environ.get("AH_BENCH_SHARED_GPU") == "1":
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    dist = None
    torch = None
    use_dist = world > 1 or args.reassemble == "allgatherv"  # single-rank smoke of the exchange branch
    transport = args.transport
    if backend != "nccl":
        transport = "torch"  # the gloo test hook moves device tensors through torch
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if world == 1 and "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29533")
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)  # gloo / RCCL print banners on the C stdout: keep stdout for the ONE JSON line
        try:
            if transport == "capi":
                # rendezvous only: a CPU (gloo) group ships rank 0's RCCL unique id; the data path never sees torch
                dist.init_process_group("gloo", rank=rank, world_size=world)
            elif backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        finally:
            os.dup2(keep, 1)
            os.close(keep)

    # N > 1: every transport is tried in a child process first (probe_transport above); one that hangs or aborts costs
    # that child, and the run goes on with the next transport — or shard-local, saying so — instead of ending there
    probe_notes, exchange_ok = [], True
    if (world > 1 and use_dist and transport == "capi" and not args.pmc_child and os.environ.get("AH_BENCH_PROBE", "1") != "0"):
        ok, why = probe_transport(dist, rank, world, "capi")
        if not ok:
            probe_notes.append("ah_comm transport failed its probe (" + why + ")")
            ok, why = probe_transport(dist, rank, world, "torch")
            if ok:
                transport = "torch"
            else:
                probe_notes.append("torch.distributed transport failed its probe (" + why + ")")
                exchange_ok = False

    import arrow_rs_amd as A
    from arrow_rs_amd import compute as K
    ctx = A.Context(local_rank)
    A.set_default_context(ctx)
    env = Env()
    env.A, env.K, env.ctx, env.args, env.rank, env.local_rank, env.world = A, K, ctx, args, rank, local_rank, world
    env.comm = None
    transport_note = "; ".join(probe_notes) if probe_notes else None
    if use_dist and not exchange_ok:
        env.comm = ShardLocalOnly(dist, world)
        transport = "torch"  # (barriers and the MAX over ranks go through the CPU process group)
    elif use_dist:
        from arrow_rs_amd import distributed as D
        if transport == "capi":
            def share_id(payload):
                box = [payload]
                dist.broadcast_object_list(box, src=0)
                return box[0]
            err = None
            try:
                env.comm = D.CApiCommunicator(ctx, rank, world, share_id)
            except Exception as ex:  # noqa: BLE001 - keep the run alive on the torch transport, say so in the line
                err = repr(ex)[:200]
            flags = [None] * world
            dist.all_gather_object(flags, err)
            if any(flags):  # every rank switches together
                transport = "torch"
                transport_note = "; ".join(probe_notes + ["capi transport failed: " + str(next(f for f in flags if f))])
                env.comm = None
        if env.comm is None:
            g = dist.new_group(backend="nccl") if (dist.get_backend() != "nccl" and backend == "nccl") else None
            env.comm = D.Communicator(ctx, dist, group=g)
            env.torch_group = g
    reassemble = exchange_ok and ((world > 1 and args.reassemble == "auto") or args.reassemble == "allgatherv")

    def sync_all():
        ctx.synchronize()
        if use_dist:
            if transport == "capi":
                env.comm.barrier()
            else:
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
    env.sync_all = sync_all

    wl = args.workload
    if args.only_narrow:
        print(json.dumps({"configs_narrow": {k: _compact_config(v) for k, v in narrow_configs(env).items()}}), flush=True)
        return
    if args.only_coalesce_sweep:
        only = None if args.only_coalesce_sweep == "all" else args.only_coalesce_sweep.split(",")
        print(json.dumps({"coalesce_by_batch_rows": coalesce_by_batch_rows(env, only)}), flush=True)
        return
    W = build_workload(env, wl)
    n = W["n"]
    if args.pmc_child:  # the profiled re-run: just the steps, no reporting
        run_timed(env, W, args.steps, args.warmup, False)
        return
    elapsed, prof, out = run_timed(env, W, args.steps, args.warmup, reassemble)
    comm = env.comm
    # the same loop without the per-kernel HIP events: what the measurement itself costs (nothing for three big
    # kernels per step, ~20 % for the coalescer's 500 small launches per step)
    no_events_ms = None
    if world == 1:
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = W["step"](reassemble)
        sync_all()
        no_events_ms = (time.perf_counter() - t0) / args.steps * 1e3
    if wl == "filter_take" and reassemble and prof["take_gather"][1] == 0:
        # the take ran on the side context during the timed loop: time it alone for the kernel table
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(3):
            K.take(W["col"], W["idx"])
        prof["take_gather"] = ctx.profile_get("take_gather")
        ctx.profile(False)

    if args.dump_gathered and rank == 0 and W["state"].get("gathered") is not None:
        import numpy as np
        g = W["state"].pop("gathered")
        dump = {"selected_local": np.int64(W["state"]["k"]), "gathered_rows": np.int64(g.num_rows())}
        for i, c in enumerate(g.columns):
            dump[f"values{i}"] = c.values_numpy()
            dump[f"valid{i}"] = c.valid_mask()
        np.savez(args.dump_gathered, **dump)
        del g, dump
    W["state"].pop("gathered", None)
    local_elapsed = None
    if reassemble:  # same run, same data, without the exchange step
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            W["step"](False)
        sync_all()
        local_elapsed = time.perf_counter() - t0

    if use_dist:  # MAX over ranks
        if transport == "capi":
            mx = env.comm.allreduce_max([elapsed, local_elapsed or 0.0])
        else:
            tt = torch.tensor([elapsed, local_elapsed or 0.0], dtype=torch.float64,
                              device=f"cuda:{local_rank}" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            mx = [float(tt[0]), float(tt[1])]
        elapsed, local_elapsed = mx[0], (mx[1] if reassemble else None)

    extra = {}
    if wl == "filter_take" and n < 2**32:
        # SURVEY §8d 2b variants, timed outside the step: (ii) sorted indices = the positions the predicate selects
        # (what a filter->indices->take pipeline feeds take); (iii) the same random indices with 10 % null indices
        col, idx, pred = W["col"], W["idx"], W["pred"]
        iota = ctx.alloc(n * 4)
        ctx.check(ctx.lib.ah_gen_iota_u32(ctx.handle, iota.ptr, n, 0))
        sorted_idx = K.filter(mk_array(A, ctx, A.UInt32, n, iota), pred)
        del iota
        ivalid = ctx.alloc(((idx.length + 63) // 64) * 8)
        ctx.check(ctx.lib.ah_gen_bernoulli_bits(ctx.handle, ivalid.ptr, idx.length, 47, 0.9, 0))
        R = A.array._RawMem
        null_idx = A.Array(ctx, A.UInt32, idx.length, idx.values, 0, R(ivalid.ptr, ivalid.nbytes, ivalid), 0,
                           idx.length - count_bits(ctx, ivalid, idx.length))
        for name, ix in (("take_sorted_indices_ms", sorted_idx), ("take_null_indices_ms", null_idx)):
            K.take(col, ix)
            ctx.profile(True)
            ctx.profile_reset()
            for _ in range(3):
                K.take(col, ix)
            ms, cnt = ctx.profile_get("take_gather")
            extra[name] = round(ms / max(cnt, 1), 4)
            ctx.profile(False)
            if name == "take_sorted_indices_ms":
                # what filter -> indices -> take feeds the kernel; same algorithmic bytes per index
                tb = ix.length * (4 + 8 + 8) + 2 * ((ix.length + 7) // 8)
                extra["roofline_take_sorted"] = roofline_obj("take_gather (sorted indices = positions of the predicate)", tb,
                                                             ms / max(cnt, 1), cnt)
        del sorted_idx, null_idx
        # the selectivity axis of the reference's own bench (arrow/benches/filter_kernels.rs: kept 1/1024 ... 1023/1024):
        # the same column through filter at other densities, scatter kernel time per call.  Sparse selections take the
        # tile-per-wave kernel; the full sweep is profiles/r03_selectivity_sweep.md
        if args.rows == 1_000_000_000 and not args.no_configs:
            by_sel = {}
            for i, sel in enumerate((0.001, 0.01, 0.5)):
                try:
                    pr = gen_predicate(A, ctx, n, 61 + i, sel, row0=0 if world == 1 else rank * n)
                    K.filter(col, pr)
                    ctx.profile(True)
                    ctx.profile_reset()
                    for _ in range(3):
                        f = K.filter(col, pr)
                    ms, cnt = ctx.profile_get("filter_scatter")
                    ctx.profile(False)
                    by_sel[str(sel)] = {"filter_scatter_ms": round(ms / max(cnt, 1), 4), "selected_rows": f.length}
                    del pr, f
                except Exception as ex:  # noqa: BLE001 - a side measurement never costs the headline
                    by_sel[str(sel)] = {"error": repr(ex)[:200]}
            extra["filter_by_selectivity"] = by_sel

    # ranks the transport really initialised (ah_comm_world / the process group), never the --gpus argument
    n_ranks = (int(getattr(env.comm, "world", world)) if env.comm is not None else 1) if use_dist else 1
    if n_ranks != world:
        raise SystemExit(f"bench.py: launcher says {world} ranks but the transport initialised {n_ranks}")
    devices = {local_rank}
    if use_dist:
        got = [None] * world
        dist.all_gather_object(got, (local_rank, os.environ.get("AH_BENCH_SHARED_GPU") == "1"))
        devices = {d for d, _ in got}
    line = None
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = n * world * args.steps / elapsed / 1e6
        kern, avg_ms, launches, alg, workload, metric, dtype = describe(env, wl, W, prof, out, args.steps)
        traffic = pmc_traffic_inrun(args, wl) if world == 1 else None
        tr = traffic if isinstance(traffic, dict) and "error" not in traffic else {}
        rf = roofline_obj(kern, alg, avg_ms, launches, tr.get("step_total") if wl in PMC_STEP_KERNELS else tr.get(kern))
        if wl in PMC_STEP_KERNELS:
            rf["note"] = ("all launches of one step; traffic = PMC bytes of those launches per step.  Random row access moves one "
                          "128-byte line per offsets pair / validity bit / short value, so traffic_frac, not frac, says how "
                          "close to the memory system's limit the step runs")
        if wl == "filter_take":
            idx, st = W["idx"], W["state"]
            take_bytes = idx.length * (4 + 8 + 8) + 2 * ((idx.length + 7) // 8)
            tk_ms, tk_n = prof["take_gather"]
            tk_avg = tk_ms / max(tk_n, 1)
            rf["note"] = ("frac is in ALGORITHMIC bytes (SURVEY §8d); the kernel skips 128-byte lines that hold no selected "
                          "row, so its PMC traffic is below the algorithmic figure")
            extra.update({
                "roofline_filter_scatter": rf,
                "filter_selected_rows": st["k"], "filter_null_count": st["fn"], "take_indices": idx.length,
                "step_algorithmic_GBps": round((alg + take_bytes) / (ms_step * 1e-3) / 1e9, 1),
            })
            if tk_avg > avg_ms:
                # the time-dominant kernel of the step is the random gather: report IT as `roofline`.  Its binding
                # resource is the fabric's request rate, not bytes: one request per value + one per validity byte.
                rf = roofline_obj("take_gather", take_bytes, tk_avg, tk_n, tr.get("take_gather"))
                reqs = idx.length * (2 if st["tn"] > 0 else 1)
                rf["requests"] = {"per_launch": reqs, "achieved_G_per_s": round(reqs / (tk_avg * 1e-3) / 1e9, 1),
                                  "probe_max_G_per_s": PROBE_MAX_FILLS_G,
                                  "frac_of_probe_max": round(reqs / (tk_avg * 1e-3) / 1e9 / PROBE_MAX_FILLS_G, 3),
                                  "what": "128-byte L2 line fills (TCC_EA0_RDREQ_128B): 1 per gathered value + 1 per validity "
                                          "bit; probe_max = the measured fill rate of the bare value+bit gather probe "
                                          "(profiles/r02_take_ablation.md, value_bit: 201.6 M fills / 3.819 ms), not a derived ceiling"}
        line = {
            "metric": metric, "value": round(value, 1), "unit": "Mrows/s", "n_gpus": n_ranks,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": workload, "rows_per_gpu": n, "parallelism": f"row-sharded x{world}",
                       "distinct_devices": len(devices),
                       "reassemble": ("allgatherv" if reassemble else ("none" if exchange_ok else "skipped: no transport passed its probe")),
                       "transport": (("ah_comm (RCCL bound by libarrow_hip.so)" if transport == "capi" else "torch.distributed")
                                     if use_dist else "none")},
            "roofline": rf,
        }
        if isinstance(traffic, dict):
            line["pmc_traffic_bytes_per_launch"] = traffic
            line["pmc_traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two steps re-run inside this invocation"
        if transport_note:
            line["config"]["transport_note"] = transport_note
        if wl == "filter_take" and W["state"].get("reassemble_error"):
            line["config"]["reassemble"] = "failed: " + W["state"]["reassemble_error"]
        if comm is not None and getattr(comm, "timings", None):
            line["reassemble_last_ms"] = comm.timings
            ex = getattr(comm, "last_exchange", None)
            if ex and ex["peers"] and comm.timings.get("total"):
                # GB/s per xGMI link = what one rank pushes to ONE peer over the exchange time of the
                # last step (each peer sits on its own link; 7 links x ~153 GB/s per MI355X); egress = all peers
                secs = comm.timings["total"] * 1e-3
                line["exchange"] = dict(ex, per_link_GBps=round(ex["bytes_to_each_peer"] / secs / 1e9, 2),
                                        egress_GBps=round(ex["bytes_to_each_peer"] * ex["peers"] / secs / 1e9, 2),
                                        ingress_GBps=round(ex["bytes_received"] / secs / 1e9, 2))
        line["kernel_avg_ms"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()}
        # what the step spends outside its profiled kernels (host waits, launches, small helper kernels)
        line["host_gap_ms"] = round(ms_step - sum(v[0] for v in prof.values()) / max(args.steps, 1), 4)
        if no_events_ms is not None:
            line["ms_per_step_without_kernel_events"] = round(no_events_ms, 4)
        if local_elapsed:
            line["local_value"] = round(n * world * args.steps / local_elapsed / 1e6, 1)
            line["local_ms_per_step"] = round(local_elapsed / args.steps * 1e3, 4)
        line.update(extra)
        try:  # the pooled allocator's accounting for this run (ah_context_stats): peak HBM held by results + scratch
            ms_ = ctx.memory_stats()
            line["hbm_pool"] = {"high_water_GB": round(ms_["high_water_bytes"] / 1e9, 3), "live_GB": round(ms_["live_bytes"] / 1e9, 3),
                                "reserved_high_water_GB": round(ms_["reserved_high_water_bytes"] / 1e9, 3),
                                "alloc_calls": ms_["alloc_calls"], "device_malloc_calls": ms_["device_malloc_calls"]}
        except Exception:  # noqa: BLE001
            pass

    # the other single-GPU configurations of BASELINE.json, a few steps each, inside the same line
    if wl == "filter_take" and world == 1 and not args.no_configs and args.rows == 1_000_000_000:
        W = out = None
        configs, next_rows = {}, {}
        for w2 in EXTRA_CONFIGS + NEXT_ROWS:
            dest = configs if w2 in EXTRA_CONFIGS else next_rows
            try:
                ctx.lib.ah_pool_trim(ctx.handle)
                W2 = build_workload(env, w2)
                el2, prof2, out2 = run_timed(env, W2, args.config_steps, 2, False, settle=True)
                kern, avg_ms, launches, alg, workload, metric, dtype = describe(env, w2, W2, prof2, out2, args.config_steps)
                ms2 = el2 / args.config_steps * 1e3
                dest[w2] = {"workload": workload, "rows": W2["n"], "steps": args.config_steps,
                            "ms": round(ms2, 4), "value": round(W2["n"] / (ms2 * 1e-3) / 1e6, 1), "unit": "Mrows/s",
                            "dtype": dtype, "roofline": roofline_obj(kern, alg, avg_ms, launches),
                            "kernel_avg_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in prof2.items()},
                            "host_gap_ms": round(ms2 - sum(v[0] for v in prof2.values()) / args.config_steps, 4),
                            "settle_ms": W2.get("settle_ms")}  # the untimed steps before the timed ones (run_timed)
                if w2 == "string_filter" and W2.get("alg_survey_rule"):
                    a2 = W2["alg_survey_rule"]
                    dest[w2]["roofline"]["frac_selected_bytes_only"] = dest[w2]["roofline"]["frac"]
                    dest[w2]["roofline"]["algorithmic_bytes_survey_rule"] = a2
                    dest[w2]["roofline"]["frac_survey_rule"] = round(a2 / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                    dest[w2]["roofline"]["note"] = ("frac prices the selected rows' bytes only (read once, written once); frac_survey_rule prices the "
                                                    "whole text buffer as an input read once in full (SURVEY 8d)")
                if w2 == "coalesce":  # the single-push forms of the same step, beside the grouped one
                    for key, pipe, what in (("single_push", "0", "the same batches through one synchronous ah_coalescer_push_batch_with_filter call each"),
                                            ("single_push_pipelined", "1", "one batch per push, begin of batch i + 1 before end of batch i"),
                                            ("grouped_not_pipelined", None, "8 batches per synchronous ah_coalescer_push_batches_with_filters call (round 3's form)")):
                        try:
                            os.environ["AH_COALESCE_GROUP"] = "1" if pipe is not None else "8"
                            os.environ["AH_COALESCE_PIPELINE"] = pipe or "0"
                            W3 = build_workload(env, "coalesce")
                            for _ in range(2):
                                W3["step"](False)
                            env.sync_all()
                            t0 = time.perf_counter()
                            for _ in range(args.config_steps):
                                W3["step"](False)
                            env.sync_all()
                            ms1 = (time.perf_counter() - t0) / args.config_steps * 1e3
                            dest[w2][key] = {"ms_without_kernel_events": round(ms1, 4),
                                             "frac_without_kernel_events": round(alg / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "what": what}
                            W3 = None
                        finally:
                            os.environ.pop("AH_COALESCE_GROUP", None)
                            os.environ.pop("AH_COALESCE_PIPELINE", None)
                if w2 == "coalesce":  # many small launches per step: the per-kernel HIP events themselves cost time
                    env.sync_all()
                    t0 = time.perf_counter()
                    for _ in range(args.config_steps):
                        W2["step"](False)
                    env.sync_all()
                    ms3 = (time.perf_counter() - t0) / args.config_steps * 1e3
                    dest[w2]["ms_without_kernel_events"] = round(ms3, 4)
                    dest[w2]["frac_without_kernel_events"] = round(alg / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                W2 = out2 = None
            except Exception as ex:  # noqa: BLE001 - never lose the headline to a secondary config
                dest[w2] = {"error": repr(ex)[:300]}
        ctx.lib.ah_pool_trim(ctx.handle)
        line["configs"] = configs
        line["next_rows"] = next_rows
        line["configs_narrow"] = narrow_configs(env)
        try:
            line["coalesce_by_batch_rows"] = coalesce_by_batch_rows(env)
        except Exception as ex:  # noqa: BLE001
            line["coalesce_by_batch_rows"] = {"error": repr(ex)[:300]}

    if wl == "filter_take" and (world > 1 or args.reassemble == "allgatherv") and use_dist and exchange_ok and not args.no_configs:
        # BASELINE configs[4]: {Int64, Float64, bitmaps} per shard through filter_record_batch, then ONE exchange of both
        # columns (ah_all_gather_columns) — every rank runs it, rank 0 reports
        W = out = None
        try:
            ctx.lib.ah_pool_trim(ctx.handle)
            W2 = build_workload(env, "record_batch")
            el2, prof2, out2 = run_timed(env, W2, args.config_steps, 2, True)
            sync_all()
            t0 = time.perf_counter()
            for _ in range(args.config_steps):
                W2["step"](False)
            sync_all()
            loc2 = time.perf_counter() - t0
            if transport == "capi":
                el2, loc2 = env.comm.allreduce_max([el2, loc2])
            else:
                tt = torch.tensor([el2, loc2], dtype=torch.float64, device=f"cuda:{local_rank}" if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el2, loc2 = float(tt[0]), float(tt[1])
            if rank == 0:
                kern, avg_ms, launches, alg, workload, metric, dtype = describe(env, "record_batch", W2, prof2, out2, args.config_steps)
                ms2 = el2 / args.config_steps * 1e3
                rb = {"workload": workload.replace("configs[4] shape", "configs[4]") + f", all-gatherv of both columns over {world} ranks",
                      "rows_per_gpu": W2["n"], "steps": args.config_steps, "ms": round(ms2, 4),
                      "value": round(W2["n"] * world / (ms2 * 1e-3) / 1e6, 1), "unit": "Mrows/s", "dtype": dtype,
                      "local_ms": round(loc2 / args.config_steps * 1e3, 4),
                      "local_value": round(W2["n"] * world * args.config_steps / loc2 / 1e6, 1),
                      "roofline": roofline_obj(kern, alg, avg_ms, launches), "gathered_rows": W2["state"].get("gk")}
                ex, tm = getattr(env.comm, "last_exchange", None), getattr(env.comm, "timings", None)
                if ex and tm and tm.get("total") and ex["peers"]:
                    secs = tm["total"] * 1e-3
                    rb["exchange"] = dict(ex, total_ms=tm["total"], counts_ms=tm.get("counts"),
                                          per_link_GBps=round(ex["bytes_to_each_peer"] / secs / 1e9, 2),
                                          egress_GBps=round(ex["bytes_to_each_peer"] * ex["peers"] / secs / 1e9, 2),
                                          ingress_GBps=round(ex["bytes_received"] / secs / 1e9, 2))
                line.setdefault("configs", {})["record_batch_allgather"] = rb
            W2 = out2 = None
        except Exception as exc:  # noqa: BLE001 - never lose the headline to a secondary config
            if rank == 0:
                line.setdefault("configs", {})["record_batch_allgather"] = {"error": repr(exc)[:300]}

    if rank == 0:
        if not args.no_cpu_baseline and world == 1 and wl == "filter_take":
            try:
                line["crossover_rows"] = crossover_rows(env)
            except Exception as ex:  # noqa: BLE001
                line["crossover_rows"] = {"error": repr(ex)[:200]}
            try:
                line["reference_bench_shapes"] = reference_bench_shapes(env)
            except Exception as ex:  # noqa: BLE001
                line["reference_bench_shapes"] = {"error": repr(ex)[:200]}
            try:
                cb = cpu_baseline_filter_take(args)
                if cb:
                    line["cpu_baseline"] = cb
            except Exception as ex:  # never lose the GPU line to a baseline hiccup
                line["cpu_baseline"] = {"error": repr(ex)}
        emit(line, args)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
