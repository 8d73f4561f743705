"""N > 1 through the REAL exchange code of libarrow_hip.so on ONE GPU: every rank is a host thread with its own context
and communicator, and RCCL's entry points are served by tests/cpp/fake_rccl.cpp (AH_RCCL_LIBRARY), where a
send / recv pair is a matched device-to-device copy.  Everything else — the count all-gather with its status / schema
words, offsets, the grouped exchange into final positions, the staged bit pieces and the one-launch merge, the string
offset rebase, null counts, the record-batch form through begin / end, barrier and max-reduce — is the product path,
checked against the oracle's un-sharded result (tests/comm_cases.py).
usage: comm_ranks_worker.py <world>.  TEST INFRASTRUCTURE (uses the oracle)."""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import comm_cases as cc  # noqa: E402
from comm_cases import A, D, orc, ROOT  # noqa: E402

import faulthandler  # noqa: E402

world = int(sys.argv[1])
# a rank stuck in a collective would otherwise sit there until the caller's timeout: dump every thread's stack and leave
faulthandler.dump_traceback_later(int(os.environ.get("AH_COMM_WATCHDOG_S", "420")), exit=True)
oracle = orc.load(os.path.join(ROOT, "oracle", "liboracle.so"))
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
        cc.run_rank(ctx, comm, oracle, r, world, heavy=(world == 2))
    except BaseException as ex:  # noqa: BLE001
        import traceback
        # the other ranks are (or soon will be) blocked inside a collective waiting for this one: report and leave NOW
        print(f"FAILED rank {r}: {ex!r}\n{traceback.format_exc()}", flush=True)
        os._exit(1)


ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
[t.start() for t in ths]
[t.join(timeout=800) for t in ths]
if errors or any(t.is_alive() for t in ths):
    print("FAILED", "\n".join(errors), [t.is_alive() for t in ths], flush=True)
    os._exit(1)
print("COMM_RANKS_OK", world, flush=True)
os._exit(0)
