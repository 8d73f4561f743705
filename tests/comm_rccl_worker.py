"""One rank of the REAL-RCCL exchange test: `comm_rccl_worker.py <rank> <world> <rendezvous dir>`, one process per GPU
(rank r on device r), no torch anywhere.  Rank 0's 128-byte unique id travels through a file in the rendezvous
directory (an engine would use its control plane).  Every rank runs tests/comm_cases.py::run_rank: product exchange
code over real xGMI transport, compared with the oracle's un-sharded result.
TEST INFRASTRUCTURE (uses the oracle)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import comm_cases as cc  # noqa: E402
from comm_cases import A, D, orc, ROOT  # noqa: E402

rank, world, rdv = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
shared = os.environ.get("AH_TEST_SHARED_GPU") == "1"  # fake transport: every rank on GPU 0


def share(payload):
    path = os.path.join(rdv, "unique_id")
    if payload is not None:
        with open(path + ".tmp", "wb") as f:
            f.write(payload)
        os.replace(path + ".tmp", path)
        return payload
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 120:
            raise TimeoutError("rank 0 never published the unique id")
        time.sleep(0.01)
    return open(path, "rb").read()


ctx = A.Context(0 if shared else rank)
A.set_default_context(ctx)
oracle = orc.load(os.path.join(ROOT, "oracle", "liboracle.so"))
comm = D.CApiCommunicator(ctx, rank, world, share)
cc.run_rank(ctx, comm, oracle, rank, world, heavy=True)
comm.barrier()
del comm
print(f"COMM_RCCL_RANK_OK {rank}/{world}", flush=True)
