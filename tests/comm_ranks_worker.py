"""N > 1 through the REAL exchange code of libarrow_hip.so on ONE GPU: every rank is a host thread with its own context
and communicator, and RCCL's eleven entry points are served by tests/cpp/fake_rccl.cpp (AH_RCCL_LIBRARY), where a
send / recv pair is a matched device-to-device copy.  Everything else — the count all-gather, offsets, the grouped
exchange into final positions, the staged validity pieces and the one-launch merge, null counts, the record-batch form,
barrier and max-reduce — is the product path, checked against the oracle's un-sharded result.
usage: comm_ranks_worker.py <world>.  TEST INFRASTRUCTURE (uses the oracle)."""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import arrow_rs_amd as A  # noqa: E402
from arrow_rs_amd import compute as K  # noqa: E402
from arrow_rs_amd import distributed as D  # noqa: E402
import orc  # noqa: E402
from orc import HostArray, assert_logical_eq  # noqa: E402

world = int(sys.argv[1])
oracle = orc.load(os.path.join(ROOT, "oracle", "liboracle.so"))
rng = np.random.default_rng(100 + world)
N = 1_000_003
cases = []
for dt, vals in ((A.Int64, rng.integers(-2**62, 2**62, N)), (A.Float64, rng.standard_normal(N)),
                 (A.Int16, rng.integers(-2**15, 2**15 - 1, N).astype(np.int16))):
    for valid in (rng.random(N) < 0.9, None):
        cases.append((HostArray(dt, np.asarray(vals, dtype=dt.np_dtype), valid), HostArray(A.Boolean, rng.random(N) < 0.13)))
# a case where only SOME shards carry nulls (the others contribute all-ones pieces), and one with an empty shard result
v = rng.random(N) < 0.999
v[N // world:] = True
cases.append((HostArray(A.Int64, rng.integers(0, 9, N), v), HostArray(A.Boolean, rng.random(N) < 0.5)))
m = rng.random(N) < 0.2
m[:D.shard_range(N, 0, world)[1]] = False  # rank 0 selects nothing
cases.append((HostArray(A.Int32, rng.integers(0, 9, N).astype(np.int32), rng.random(N) < 0.7), HostArray(A.Boolean, m)))

box, lock, bar = {}, threading.Lock(), threading.Barrier(world)
errors = []


def share(payload):
    if payload is not None:
        box["id"] = payload
    bar.wait()
    return box["id"]


def rank_main(r):
    try:
        ctx = A.Context(0)
        comm = D.CApiCommunicator(ctx, r, world, share)
        s, e = D.shard_range(N, r, world)
        for ci, (h, mask) in enumerate(cases):
            hs, ms = h.slice(s, e - s), mask.slice(s, e - s)
            f = K.filter(hs.to_device(ctx), ms.to_device(ctx))
            g = comm.all_gatherv(f)
            exp = oracle.filter(h, mask)
            assert_logical_eq(HostArray.from_device(g), exp, f"rank {r} case {ci}")
            assert g.null_count() == exp.null_count, f"rank {r} case {ci} null_count"
            assert (g.validity is None) == (exp.null_count == 0), f"rank {r} case {ci} null buffer presence"
            assert comm.last_exchange["peers"] == world - 1
        # record-batch form: two columns, one count exchange + one group
        (ha, ma), (hb, _) = cases[0], cases[2]
        rb = A.RecordBatch(["a", "b"], [ha.slice(s, e - s).to_device(ctx), hb.slice(s, e - s).to_device(ctx)], e - s)
        fb = K.filter_record_batch(rb, ma.slice(s, e - s).to_device(ctx))
        out = comm.all_gather_record_batch(fb)
        assert_logical_eq(HostArray.from_device(out.columns[0]), oracle.filter(ha, ma), f"rank {r} batch col a")
        assert_logical_eq(HostArray.from_device(out.columns[1]), oracle.filter(hb, ma), f"rank {r} batch col b")
        comm.barrier()
        assert comm.allreduce_max([float(r), -float(r)]) == [float(world - 1), 0.0]
    except BaseException as ex:  # noqa: BLE001
        with lock:
            errors.append(f"rank {r}: {ex!r}")
        try:
            bar.abort()
        except Exception:
            pass


ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
[t.start() for t in ths]
[t.join(timeout=600) for t in ths]
if errors or any(t.is_alive() for t in ths):
    print("FAILED", errors, [t.is_alive() for t in ths], flush=True)
    os._exit(1)
print("COMM_RANKS_OK", world, flush=True)
os._exit(0)
