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

Multi-GPU (weak scaling): every rank owns one 1e9-row shard of an N x 1e9-row column
(row-range sharding), filters/takes locally, then the filtered shard results are
reassembled on every rank with an all-gatherv over RCCL (north_star).  ``value`` includes the
reassembly; ``local_value`` is the same run's rate without it.

Other workloads (for the per-kernel roofline table in DESIGN.md): --workload arith|cmp|cast|cast_string.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable copy


def pmc_traffic(kernel, args):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_traffic.json,
    produced by tools/collect_profiles.sh + tools/profile_summary.py: FETCH_SIZE and WRITE_SIZE in
    separate passes, FETCH doubled for wide coalesced reads as MI355X_MICROARCH.md prescribes).
    Only valid for the exact default workload the counters were collected on; else null."""
    if not (args.workload == "filter_take" and args.rows == 1_000_000_000 and args.selectivity == 0.1
            and args.valid == 0.9):
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    try:
        return json.load(open(files[-1]))["hbm_bytes_per_launch"].get(kernel)
    except Exception:
        return None


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--rows", type=int, default=1_000_000_000, help="rows per GPU")
    p.add_argument("--selectivity", type=float, default=0.1)
    p.add_argument("--valid", type=float, default=0.9)
    p.add_argument("--workload", default="filter_take",
                   choices=["filter_take", "arith", "cmp", "cast", "cast_string", "coalesce", "string_filter_take", "aggregate", "sort", "record_batch"])
    p.add_argument("--batch-rows", type=int, default=1 << 24, help="coalesce workload: rows per pushed batch")
    p.add_argument("--reassemble", default="auto", choices=["auto", "none", "allgatherv"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample-rows", type=int, default=1 << 26)
    return p.parse_args()


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


def gen_predicate(A, ctx, n, seed, p_true, row0):
    bits = ctx.alloc(((n + 63) // 64) * 8)
    ctx.check(ctx.lib.ah_gen_bernoulli_bits(ctx.handle, bits.ptr, n, seed, p_true, row0))
    return mk_array(A, ctx, A.Boolean, n, bits)


def cpu_baseline_filter_take(args):
    """The oracle (a scalar port of the reference's algorithm) timed on ONE host core over a
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
    # secondary: the same port row-sharded over host threads (how an engine would parallelise the
    # single-threaded reference kernels); ctypes releases the GIL during the calls
    try:
        import threading
        T = max(1, min(os.cpu_count() or 1, 64))
        per = (n // T) // 64 * 64
        if T > 1 and per > 0:
            shards = []
            for k in range(T):
                sv = orc.HostArray(A.Int64, vals[k * per:(k + 1) * per], valid[k * per:(k + 1) * per])
                sm = orc.HostArray(A.Boolean, mask[k * per:(k + 1) * per])
                si = orc.HostArray(A.UInt32, (idx[k * (len(idx) // T):(k + 1) * (len(idx) // T)] % per).astype(np.uint32))
                shards.append((orc._Held(sv), orc._Held(sm), orc._Held(si)))

            def work(sh):
                for _ in range(3):
                    o = orc.Out()
                    oracle.lib.orc_filter(C.byref(sh[0].view), C.byref(sh[1].view), C.byref(o))
                    oracle.lib.orc_release(C.byref(o))
                    o = orc.Out()
                    oracle.lib.orc_take(C.byref(sh[0].view), C.byref(sh[2].view), 0, C.byref(o))
                    oracle.lib.orc_release(C.byref(o))
            ths = [threading.Thread(target=work, args=(sh,)) for sh in shards]
            t0 = time.perf_counter()
            [th.start() for th in ths]
            [th.join() for th in ths]
            dt = time.perf_counter() - t0
            res["all_cores"] = {"value": round(per * T * 3 / dt / 1e6, 1), "unit": "Mrows/s", "cores": T,
                                "sample": f"3 x (filter + take), {T} threads x {per} rows"}
    except Exception as ex:
        res["all_cores"] = {"error": repr(ex)}
    # independent sanity line: Arrow C++ through pyarrow on the same sample.  A different
    # implementation (filter/take semantics coincide) — NOT the reference and not the baseline.
    try:
        import pyarrow as pa
        import pyarrow.compute as pc
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


def main():
    args = parse()
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
    use_dist = world > 1 or args.reassemble == "allgatherv"  # single-rank smoke of the RCCL branch
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if world == 1 and "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import arrow_rs_amd as A
    from arrow_rs_amd import compute as K
    ctx = A.Context(local_rank)
    A.set_default_context(ctx)
    n = args.rows
    row0 = rank * n  # this rank's row range of the global column
    reassemble = (world > 1 and args.reassemble == "auto") or args.reassemble == "allgatherv"
    comm = None
    if use_dist:
        from arrow_rs_amd import distributed as D
        comm = D.Communicator(ctx, dist)

    def sync_all():
        ctx.synchronize()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    wl = args.workload
    extra = {}
    if wl == "filter_take":
        col = gen_i64_column(A, ctx, n, 42, args.valid, row0)
        pred = gen_predicate(A, ctx, n, 44, args.selectivity, row0)
        nidx = max(1, int(n * args.selectivity))
        ib = ctx.alloc(nidx * 4)
        ctx.check(ctx.lib.ah_gen_uniform_u32(ctx.handle, ib.ptr, nidx, 45 + rank, n if n < 2**32 else 0, 0))
        idx = mk_array(A, ctx, A.UInt32, nidx, ib)
        # variant (ii) of SURVEY §8d 2b: sorted indices = positions selected by the predicate
        # (what a filter->indices->take pipeline feeds take); timed outside the step, reported as extra
        iota = ctx.alloc(n * 4)
        ctx.check(ctx.lib.ah_gen_iota_u32(ctx.handle, iota.ptr, n, 0)) if n < 2**32 else None
        sorted_idx = K.filter(mk_array(A, ctx, A.UInt32, n, iota), pred) if n < 2**32 else None
        del iota
        state = {}
        # the take is independent of the filter output, so at N>1 it runs on a SECOND context (own HIP
        # stream, own thread; ctypes drops the GIL) while the filtered shard is being all-gathered over
        # RCCL on torch's stream: collective and compute overlap on separate streams
        ctx2 = A.Context(local_rank) if use_dist else None

        def rebind(a, c):
            return A.Array(c, a.data_type, a.length, a.values, a.values_bit_offset, a.validity,
                           a.validity_bit_offset, a.null_count(), a.offsets)

        col_b, idx_b = (rebind(col, ctx2), rebind(idx, ctx2)) if use_dist else (None, None)

        def step(with_reassembly):
            f = K.filter(col, pred)
            if with_reassembly and not state.get("reassemble_error"):
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
                    g = comm.all_gatherv(f)
                    state["gk"] = g.length
                except Exception as ex:  # keep the run alive: report local-only numbers + the error
                    state["reassemble_error"] = repr(ex)[:300]
                th.join()
                if "err" in box:
                    raise box["err"]
                t = box["t"]
            else:
                t = K.take(col, idx)
            state["k"], state["fn"], state["tn"] = f.length, f.null_count(), t.null_count()
            return f, t

        kernels = ["filter_count", "filter_scatter", "take_gather"]
        dominant = "filter_scatter"
    elif wl == "coalesce":
        # BatchCoalescer::push_batch_with_filter over a stream of batches —
        # the filter scatters straight into the in-progress output batch (no intermediate array)
        col = gen_i64_column(A, ctx, n, 42, args.valid, row0)
        col2 = gen_f64_column(A, ctx, n, 52, args.valid, row0)
        pred = gen_predicate(A, ctx, n, 44, args.selectivity, row0)
        br = min(args.batch_rows, n)
        nb = n // br
        target = max(1, int(br * args.selectivity * 4))
        batches = [(A.RecordBatch(["a", "b"], [col.slice(i * br, br), col2.slice(i * br, br)]), pred.slice(i * br, br))
                   for i in range(nb)]
        state = {}

        def step(_r):
            co = K.BatchCoalescer.new(["a", "b"], [A.Int64, A.Float64], target, ctx)
            out_rows = 0
            for rb, f in batches:
                co.push_batch_with_filter(rb, f)
                while co.has_completed_batch():
                    out_rows += co.next_completed_batch().num_rows()
            co.finish_buffered_batch()
            while co.has_completed_batch():
                out_rows += co.next_completed_batch().num_rows()
            state["out_rows"] = out_rows
            return out_rows

        kernels = ["filter_count", "filter_scatter", "copy_rows"]
        dominant = "filter_scatter"
    elif wl == "string_filter_take":
        # LargeUtf8 column (the config-4 cast output) through filter and take
        n = min(n, 1 << 27) if args.rows == 1_000_000_000 else n
        src = gen_i64_column(A, ctx, n, 42, args.valid, row0, -10**6, 10**6)
        scol = K.cast(K.cast(src, A.Float64), A.LargeUtf8)
        pred = gen_predicate(A, ctx, n, 44, args.selectivity, row0)
        nidx = max(1, int(n * args.selectivity))
        ib = ctx.alloc(nidx * 4)
        ctx.check(ctx.lib.ah_gen_uniform_u32(ctx.handle, ib.ptr, nidx, 45, n, 0))
        idx = mk_array(A, ctx, A.UInt32, nidx, ib)
        state = {}

        def step(_r):
            f = K.filter(scol, pred)
            t = K.take(scol, idx)
            state["fbytes"], state["tbytes"], state["k"] = f.values.nbytes, t.values.nbytes, f.length
            return f, t

        kernels = ["filter_count", "filter_scatter", "string_ranges_scan", "string_gather_bytes", "string_take_ranges",
                   "take_gather"]
        dominant = "string_gather_bytes"
    elif wl == "aggregate":
        # sum + min + max of the Int64 column (three streaming reads per step)
        col = gen_i64_column(A, ctx, n, 42, args.valid, row0)
        G = K.aggregate
        step = lambda _r: (G.sum(col), G.min(col), G.max(col))
        kernels = ["aggregate"]
        dominant = "aggregate"
    elif wl == "record_batch":
        # BASELINE configs[4] shape: RecordBatch {Int64, Float64, each with validity} + one mask per shard;
        # filter_record_batch (one count pass, two scatters), then at N>1 ONE IPC-framed exchange of both columns
        cola = gen_i64_column(A, ctx, n, 42, args.valid, row0)
        colb = gen_f64_column(A, ctx, n, 52, args.valid, row0)
        pred = gen_predicate(A, ctx, n, 44, args.selectivity, row0)
        rb = A.RecordBatch(["a", "b"], [cola, colb], n)
        state = {}

        def step(with_reassembly):
            f = K.filter_record_batch(rb, pred)
            state["k"] = f.num_rows()
            if with_reassembly:
                parts = comm.all_gather_batches(f)
                state["gk"] = sum(p.num_rows() for p in parts)
            return f

        kernels = ["filter_count", "filter_scatter"]
        dominant = "filter_scatter"
    elif wl == "sort":
        # the producer of take's indices: sort_to_indices of the full-range Int64 column (8 radix passes)
        n = min(n, 1 << 29) if args.rows == 1_000_000_000 else n
        col = gen_i64_column(A, ctx, n, 42, args.valid, row0)
        step = lambda _r: K.sort_to_indices(col)
        kernels = ["sort_radix_pass"]
        dominant = "sort_radix_pass"
    elif wl in ("arith", "cmp"):
        a = gen_f64_column(A, ctx, n, 52, args.valid, row0)
        b = gen_f64_column(A, ctx, n, 62, args.valid, row0)
        fn = K.add_wrapping if wl == "arith" else K.lt
        step = lambda _r: fn(a, b)
        kernels = ["arith_binary" if wl == "arith" else "compare"]
        dominant = kernels[0]
    else:
        n = min(n, 1 << 29) if args.rows == 1_000_000_000 else n
        src = gen_i64_column(A, ctx, n, 42, args.valid, row0, -10**6, 10**6)
        if wl == "cast":
            step = lambda _r: K.cast(src, A.Float64)
            kernels = ["cast_numeric"]
        else:
            f64 = K.cast(src, A.Float64)
            step = lambda _r: K.cast(f64, A.LargeUtf8)
            kernels = ["cast_string_len", "cast_string_write"]
        dominant = kernels[-1]

    for _ in range(args.warmup):
        step(reassemble)
    ctx.profile(True)
    ctx.profile_reset()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(reassemble)
    sync_all()
    elapsed = time.perf_counter() - t0
    prof = {k: ctx.profile_get(k) for k in kernels}
    ctx.profile(False)
    if wl == "filter_take" and reassemble and prof["take_gather"][1] == 0:
        # the take ran on the side context during the timed loop: time it alone for the kernel table
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(3):
            K.take(col, idx)
        prof["take_gather"] = ctx.profile_get("take_gather")
        ctx.profile(False)

    local_elapsed = None
    if reassemble:  # same run, same data, without the exchange step
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(False)
        sync_all()
        local_elapsed = time.perf_counter() - t0

    if use_dist:
        tt = torch.tensor([elapsed, local_elapsed or 0.0], device=f"cuda:{local_rank}", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, local_elapsed = float(tt[0]), (float(tt[1]) if reassemble else None)

    sorted_ms = None
    if wl == "filter_take" and sorted_idx is not None:
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(3):
            ts = K.take(col, sorted_idx)
        sorted_ms = ctx.profile_get("take_gather")
        sorted_ms = sorted_ms[0] / max(sorted_ms[1], 1)
        ctx.profile(False)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = n * world * args.steps / elapsed / 1e6
        dom_ms, dom_n = prof[dominant]
        dom_avg_ms = dom_ms / max(dom_n, 1)
        if wl == "filter_take":
            k = state["k"]
            has_valid_out = state["fn"] > 0
            # values + validity + mask read once, K values (+ K bits) written once
            alg_bytes = n * 8 + 2 * ((n + 7) // 8) + k * 8 + (((k + 7) // 8) if has_valid_out else 0)
            take_bytes = idx.length * (4 + 8 + 8) + 2 * ((idx.length + 7) // 8)
            step_alg = alg_bytes + take_bytes
            tk_ms, tk_n = prof["take_gather"]
            tk_avg = tk_ms / max(tk_n, 1)
            # random 8-byte gathers pull one 128-byte L2 line each (rocprofv3 FETCH_SIZE, profiles/):
            # line traffic is the physical bound of this kernel, algorithmic bytes are 8/128 of it
            take_line_bytes = idx.length * 128 + idx.length * 4 + idx.length * 8
            rf_filter = {"bound": "hbm", "kernel": "filter_scatter", "achieved": round(alg_bytes / (dom_avg_ms * 1e-3) / 1e9, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(alg_bytes / (dom_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "traffic": pmc_traffic("filter_scatter", args), "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": round(dom_avg_ms, 4), "launches": dom_n}
            extra = {
                "roofline_filter_scatter": rf_filter,
                "take_line_traffic_GBps_model": round(take_line_bytes / (tk_avg * 1e-3) / 1e9, 1) if tk_n else None,
                "take_sorted_indices_ms": round(sorted_ms, 4) if sorted_ms else None,
                "filter_selected_rows": k, "filter_null_count": state["fn"],
                "take_indices": idx.length,
                "filter_scatter_ms": round(dom_avg_ms, 4),
                "filter_count_ms": round(prof["filter_count"][0] / max(prof["filter_count"][1], 1), 4),
                "take_gather_ms": round(tk_ms / max(tk_n, 1), 4),
                "take_algorithmic_GBps": round(take_bytes / (tk_ms / max(tk_n, 1) * 1e-3) / 1e9, 1) if tk_n else None,
                "step_algorithmic_GBps": round(step_alg / (ms_step * 1e-3) / 1e9, 1),
            }
            workload = (f"configs[1]: filter()+take() on {n}-row Int64 per GPU, {args.valid:.0%} valid, "
                        f"{args.selectivity:.0%} selectivity, {idx.length} uniform UInt32 take indices")
            metric = "filter_take_Mrows_per_s"
            dtype = "int64"
        else:
            per_row = {"arith": 24.375, "cmp": 16.5, "cast": 16.25}.get(wl)
            if wl == "string_filter_take":
                k = state["k"]
                # filter: offsets + validity + mask in, K+1 offsets + bytes out; take: idx + ranges in, offsets + bytes out
                alg_bytes = (n + 1) * 8 + 2 * ((n + 7) // 8) + (k + 1) * 8 + 2 * state["fbytes"] + \
                    idx.length * (4 + 16 + 8) + 2 * state["tbytes"]
                dom_avg_ms = sum(prof[kk][0] for kk in kernels) / max(args.steps, 1)
                dom_n = args.steps
            elif wl == "coalesce":
                k = state["out_rows"]
                # two columns share one predicate: 2 x (values + validity) + mask in, 2 x (K values + K bits) out
                alg_bytes = 2 * (n * 8 + (n + 7) // 8) + (n + 7) // 8 + 2 * (k * 8 + (k + 7) // 8)
                dom_avg_ms = sum(prof[kk][0] for kk in kernels) / max(args.steps, 1)  # all launches of one step
                dom_n = args.steps
            elif wl == "record_batch":
                k = state["k"]  # per scatter launch: one column + its validity + the mask in, K values + K bits out
                alg_bytes = n * 8 + 2 * ((n + 7) // 8) + k * 8 + (k + 7) // 8
            elif wl == "sort":
                m_valid = n - col.null_count()
                alg_bytes = m_valid * 32  # per radix pass: keys read twice, (key, index) pairs written once
            elif wl == "aggregate":
                alg_bytes = n * 8 + (n + 7) // 8  # per launch: values + validity in, 8 bytes out
            elif per_row is None:  # cast_string: input + offsets + bytes + validity
                o = out
                alg_bytes = n * 8 + (n + 7) // 8 + (n + 1) * 8 + o.values.nbytes + (n + 7) // 8
                dom_ms_all = sum(prof[k][0] for k in kernels) / max(prof[kernels[0]][1], 1)
                dom_avg_ms = dom_ms_all
            else:
                alg_bytes = int(per_row * n)
            workload = {"arith": "configs[2]: add_wrapping Float64+Float64 with NullBuffers",
                        "cmp": "configs[2]: lt Float64<Float64 with NullBuffers",
                        "cast": "configs[3]: cast Int64->Float64",
                        "cast_string": "configs[3]: cast Float64->LargeUtf8",
                        "record_batch": "configs[4] shape: filter_record_batch on {Int64, Float64} with NullBuffers"
                                        + (" + IPC-framed all_gather_batches" if reassemble else ""),
                        "sort": "arrow_ord sort_to_indices of a full-range Int64 column with NullBuffer (stable LSD radix)",
                        "aggregate": "SURVEY 8f-4: sum + min + max of an Int64 column with NullBuffer",
                        "string_filter_take": "SURVEY 8f-3: filter + take on a LargeUtf8 column (cast output)",
                        "coalesce": f"SURVEY 8f-1: BatchCoalescer.push_batch_with_filter, Int64+Float64, "
                                    f"{args.batch_rows}-row batches"}[wl] + f", {n} rows per GPU"
            metric = f"{wl}_Mrows_per_s"
            dtype = "int64" if wl in ("aggregate", "sort") else "int64+f64" if wl == "record_batch" else "f64"
        if wl == "filter_take" and tk_avg > dom_avg_ms:
            # the time-dominant kernel of the step is the random gather: report IT as `roofline`
            dominant, dom_avg_ms, dom_n, alg_bytes = "take_gather", tk_avg, tk_n, take_bytes
        achieved = alg_bytes / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
        line = {
            "metric": metric, "value": round(value, 1), "unit": "Mrows/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": workload, "rows_per_gpu": n, "parallelism": f"row-sharded x{world}",
                       "reassemble": ("allgatherv" if reassemble else "none")},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": pmc_traffic(dominant, args), "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": round(dom_avg_ms, 4), "launches": dom_n},
        }
        tr = line["roofline"]["traffic"]
        if tr and dom_avg_ms > 0:
            # what the memory system actually moved for this kernel (PMC bytes per launch / measured launch time): for the
            # random gather this is the physically binding number — one 128-byte line per 8-byte value
            line["roofline"]["traffic_GBps"] = round(tr / (dom_avg_ms * 1e-3) / 1e9, 1)
            line["roofline"]["traffic_frac"] = round(tr / (dom_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        if wl == "filter_take" and state.get("reassemble_error"):
            line["config"]["reassemble"] = "failed: " + state["reassemble_error"]
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
        if local_elapsed:
            line["local_value"] = round(n * world * args.steps / local_elapsed / 1e6, 1)
            line["local_ms_per_step"] = round(local_elapsed / args.steps * 1e3, 4)
        line.update(extra)
        if not args.no_cpu_baseline and world == 1 and wl == "filter_take":
            try:
                cb = cpu_baseline_filter_take(args)
                if cb:
                    line["cpu_baseline"] = cb
            except Exception as ex:  # never lose the GPU line to a baseline hiccup
                line["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
