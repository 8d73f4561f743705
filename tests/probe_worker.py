"""Worker of tests/test_bench_probe_cpu.py: one rank of a gloo group that runs bench.probe_transport (the orchestration around
the per-transport probe child of `bench.py --gpus N`) and writes its verdict as JSON into its OWN file
`$PROBE_VERDICT_DIR/rank<r>.json` (written whole, then renamed into place).  Two ranks printing onto the launcher's one
stdout pipe interleaved their lines about every other run — a verdict per file cannot.  """
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    t0 = time.time()
    ok, why = bench.probe_transport(dist, rank, world, sys.argv[1] if len(sys.argv) > 1 else "capi")
    out = os.path.join(os.environ["PROBE_VERDICT_DIR"], f"rank{rank}.json")
    with open(out + ".tmp", "w") as f:
        json.dump({"rank": rank, "ok": ok, "why": why, "seconds": round(time.time() - t0, 1)}, f)
    os.replace(out + ".tmp", out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
