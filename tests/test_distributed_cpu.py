"""N>1 path on CPU: world_size-2 gloo processes exercise the row-sharding + all-gatherv
reassembly logic (arrow-rs_amd/distributed.py) that bench.py --gpus N runs over RCCL.
The per-shard compute here is done by the ORACLE (no GPU on this box); what is under test is
the partitioning, the count exchange, the point-to-point all-gatherv and the reassembly order:
concat(shard results) must equal the un-sharded result (oracle: arrow_select::concat)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import arrow_rs_amd as A
from arrow_rs_amd import distributed as D
import orc
from orc import HostArray
dist.init_process_group("gloo", init_method="file://{rdzv}", rank=int(sys.argv[1]), world_size={world})
rank, world = dist.get_rank(), dist.get_world_size()
oracle = orc.load(os.path.join({root!r}, "oracle", "liboracle.so"))
n = 100_003
vals = oracle.gen_i64(n, 42, -2**63, 2**63 - 1)
valid = oracle.gen_bits(n, 43, 0.9)
mask = oracle.gen_bits(n, 44, 0.1)
s, e = D.shard_range(n, rank, world)
assert s % 64 == 0
loc = oracle.filter(HostArray(A.Int64, vals[s:e], valid[s:e]), HostArray(A.Boolean, mask[s:e]))
# 1. counts
mine = torch.tensor([len(loc), loc.null_count], dtype=torch.int64)
allc = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
dist.all_gather(allc, mine)
lens = [int(c[0]) for c in allc]
offs, total = D.exclusive_offsets(lens)
# 2. values straight to their final offsets
out = torch.zeros(total * 8, dtype=torch.uint8)
local = torch.from_numpy(loc.values.view(np.uint8).copy())
D.all_gatherv_bytes(dist, local, out, [o * 8 for o in offs], [l * 8 for l in lens])
# 3. validity pieces through a staging buffer, merged at bit offsets
pbytes = [((l + 63) // 64) * 8 for l in lens]
poffs, ptotal = D.exclusive_offsets(pbytes)
lv = loc.valid if loc.valid is not None else np.ones(len(loc), dtype=bool)
staging = torch.zeros(ptotal, dtype=torch.uint8)
D.all_gatherv_bytes(dist, torch.from_numpy(A.pack_bits(lv)[:pbytes[rank]].copy()), staging, poffs, pbytes)
merged = np.concatenate([A.unpack_bits(staging.numpy()[poffs[r]:poffs[r] + pbytes[r]], 0, lens[r]) for r in range(world)])
exp = oracle.filter(HostArray(A.Int64, vals, valid), HostArray(A.Boolean, mask))
got = HostArray(A.Int64, out.numpy().view(np.int64), merged)
orc.assert_logical_eq(got, exp, f"rank {{rank}}")
assert sum(int(c[1]) for c in allc) == exp.null_count
dist.barrier()
print("RANK_OK", rank)
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_partitions_rows():
    sys.path.insert(0, ROOT)
    from arrow_rs_amd import distributed as D
    for n in [0, 1, 63, 64, 65, 1000, 10**9, 8 * 10**9 + 17]:
        for world in [1, 2, 4, 8]:
            prev = 0
            for r in range(world):
                s, e = D.shard_range(n, r, world)
                assert s == prev and s <= e and (s % 64 == 0 or s == n)
                prev = e
            assert prev == n
    assert D.exclusive_offsets([3, 0, 5]) == ([0, 3, 3], 8)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_filter_allgatherv_gloo(oracle, world, tmp_path):
    # (a file store: the "free port" idiom can lose its port to another process between close and listen)
    code = WORKER.format(root=ROOT, rdzv=os.path.join(str(tmp_path), "rendezvous"), world=world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", GLOO_SOCKET_IFNAME="lo")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, o[-2000:]


def test_bench_refuses_to_report_n_gpus_from_fewer_devices():
    """`python bench.py --gpus N` without a launcher spawns its own ranks — and when fewer than N GPUs are visible (none, on
    this box) it prints ONE line that says so (value null, n_gpus = what is visible) and exits non-zero: a 1-GPU number is
    never reported as an N-GPU one."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "AH_BENCH_SHARED_GPU")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode != 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    d = json.loads(lines[0])
    assert d["value"] is None and d["requested_gpus"] == 2 and d["n_gpus"] < 2 and "visible" in d["error"]
