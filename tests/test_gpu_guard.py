"""Over-READ detection inside the driver-run suite.  The reference's `unsafe` gathers are Miri-checked
(.github/workflows/miri.sh:12-45; arrow-select/src/filter.rs:736,756-761; arrow-select/src/take.rs:442-454); device ASan does
not exist on gfx950 (no xnack), and the redzone canaries (test_gpu_redzone.py) only see over-WRITES.  AH_DEBUG_GUARD=1
(csrc/context.hip) serves every pool block — the inputs the tests upload, outputs, scratch, slabs — from its own virtual-memory
mapping with the buffer's end flush against an UNMAPPED granule, and unmaps a block when it is released: a kernel that reads or
writes behind a buffer, or touches one after its release, raises "Memory access fault" on every box instead of on the one box
whose neighbouring allocation happens to be missing (GPUTEST_r05).

Two halves:
* exact-fit cases (this file's parity tests): every input ends exactly where its buffer ends — no slice tail, bitmaps of
  ceil(bits / 8) bytes — at the row counts where the kernels' tiles, vectors and bitmap words end ragged (1, 63, 64, 65, 4095,
  4096, 4097, 2^16, 2^20) and leading bit / element offsets 0, 3, 61.  They run in the normal suite (plain parity) AND
* in a child pytest process with AH_DEBUG_GUARD=1 together with the golden-vector tests and the randomized parity subset."""
import os
import subprocess
import sys

import numpy as np
import pytest

import arrow_rs_amd as A
from arrow_rs_amd import compute as K
import orc
from orc import HostArray, assert_logical_eq

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LENGTHS = [1, 63, 64, 65, 4095, 4096, 4097, 1 << 16, 1 << 20]
OFFSETS = [0, 3, 61]
PRIMS = [A.Int8, A.Int16, A.Int32, A.Int64]


def _vals(rng, dt, n):
    if dt == A.Boolean:
        return rng.random(n) < 0.5
    if dt in (A.Float32, A.Float64):
        return rng.normal(size=n).astype(dt.np_dtype)
    if dt == A.Float16:
        return rng.normal(size=n).astype(np.float16)
    info = np.iinfo(dt.np_dtype)
    return rng.integers(info.min, info.max, n, dtype=dt.np_dtype, endpoint=True)


def _arr(rng, dt, n, off, valid_p=0.9):
    """(host, device) pair of n rows whose device buffers END where the data ends: `off` leading rows are sliced away (so the
    values pointer and both bit offsets are advanced by `off`), nothing follows the last row."""
    full = HostArray(dt, _vals(rng, dt, n + off), (rng.random(n + off) < valid_p) if valid_p is not None else None)
    return full.slice(off, n), full.to_device().slice(off, n)


def _eq(got_dev, exp, msg):
    got = HostArray.from_device(got_dev)
    assert_logical_eq(got, exp, msg)


@pytest.mark.parametrize("off", OFFSETS)
@pytest.mark.parametrize("n", LENGTHS)
def test_exact_fit_filter_take(ctx, oracle, n, off):
    rng = np.random.default_rng(n * 7 + off)
    for sel in (0.1, 0.5, 0.95):
        full = HostArray(A.Boolean, rng.random(n + off) < sel, (rng.random(n + off) < 0.95) if sel == 0.5 else None)
        hm, dm = full.slice(off, n), full.to_device().slice(off, n)
        for dt in PRIMS + [A.Float64, A.Boolean]:
            hv, dv = _arr(rng, dt, n, off)
            _eq(K.filter(dv, dm), oracle.filter(hv, hm), f"filter {dt} n={n} off={off} sel={sel}")
    for dt in PRIMS + [A.Boolean]:
        hv, dv = _arr(rng, dt, n, off)
        for idt in (A.UInt32, A.Int64, A.UInt8):
            k = max(1, n // 3)
            hi_ = min(n, 255) if idt == A.UInt8 else n
            hi = HostArray(idt, rng.integers(0, hi_, k).astype(idt.np_dtype), rng.random(k) < 0.9)
            _eq(K.take(dv, hi.to_device()), oracle.take(hv, hi), f"take {dt}[{idt}] n={n} off={off}")


@pytest.mark.parametrize("off", OFFSETS)
@pytest.mark.parametrize("n", LENGTHS)
def test_exact_fit_arith_cmp_cast(ctx, oracle, n, off):
    rng = np.random.default_rng(n * 11 + off)
    for dt in [A.Int8, A.Int16, A.Int32, A.Int64, A.Float32, A.Float64, A.Float16]:
        ha, da = _arr(rng, dt, n, off)
        hb, db = _arr(rng, dt, n, off, None)
        _eq(K.add_wrapping(da, db), oracle.arith(1, ha, hb), f"add_wrapping {dt} n={n} off={off}")
        _eq(K.lt(da, db), oracle.compare(2, ha, hb), f"lt {dt} n={n} off={off}")
        one = HostArray(dt, _vals(rng, dt, 1))
        _eq(K.gt_eq(da, A.Scalar(one.to_device())), oracle.compare(5, ha, one, r_scalar=True), f"gt_eq scalar {dt} n={n} off={off}")
        _eq(K.mul_wrapping(da, A.Scalar(one.to_device())), oracle.arith(5, ha, one, r_scalar=True), f"mul scalar {dt} n={n} off={off}")
    for src, dst in [(A.Int64, A.Float64), (A.Int32, A.Int64), (A.Int64, A.Int8), (A.Float64, A.Int32), (A.Int8, A.Float32),
                     (A.Float32, A.Float64), (A.Int16, A.Boolean), (A.Boolean, A.Int32)]:
        hs, ds = _arr(rng, src, n, off)
        _eq(K.cast(ds, dst), oracle.cast(hs, dst), f"cast {src}->{dst} n={n} off={off}")


@pytest.mark.parametrize("off", OFFSETS)
@pytest.mark.parametrize("n", LENGTHS)
def test_exact_fit_boolean_kernels(ctx, oracle, n, off):
    rng = np.random.default_rng(n * 13 + off)
    ha, da = _arr(rng, A.Boolean, n, off)
    hb, db = _arr(rng, A.Boolean, n, off, 0.8)
    hc, dc = _arr(rng, A.Boolean, n, off, None)
    for op, fn in enumerate([K.and_, K.or_, K.and_not, K.and_kleene, K.or_kleene]):
        _eq(fn(da, db), oracle.boolean_binary(op, ha, hb), f"boolean op {op} n={n} off={off}")
        _eq(fn(da, dc), oracle.boolean_binary(op, ha, hc), f"boolean op {op} (no nulls right) n={n} off={off}")
    for op, fn in enumerate([K.not_, K.is_null, K.is_not_null]):
        _eq(fn(da), oracle.boolean_unary(10 + op, ha), f"boolean unary {op} n={n} off={off}")
    hv, dv = _arr(rng, A.Int64, n, off)
    _eq(K.nullif(dv, db), oracle.nullif(hv, hb), f"nullif n={n} off={off}")
    # the lazy predicate: compare -> and_kleene -> filter with nothing materialised, against the materialised chain
    hw, dw = _arr(rng, A.Int64, n, off)
    lazy = K.filter_expr(dv, [("lt", dv, dw), ("gt_eq", dw, dv)], ["or_kleene"])
    _eq(lazy, oracle.filter(hv, oracle.boolean_binary(4, oracle.compare(2, hv, hw), oracle.compare(5, hw, hv))),
        f"filter_expr n={n} off={off}")


@pytest.mark.parametrize("off", [0, 3])
@pytest.mark.parametrize("n", [1, 63, 65, 4095, 4097, 1 << 16])
def test_exact_fit_strings(ctx, oracle, n, off):
    rng = np.random.default_rng(n * 17 + off)
    words = ["", "a", "héllo", "0123456789abcdef", "x" * 37]
    for dt in (A.Utf8, A.LargeUtf8):
        strs = [words[int(i)] for i in rng.integers(0, len(words), n + off)]
        full = HostArray(dt, strs, rng.random(n + off) < 0.9)
        hv, dv = full.slice(off, n), full.to_device().slice(off, n)
        hm = HostArray(A.Boolean, rng.random(n) < 0.4)
        _eq(K.filter(dv, hm.to_device()), oracle.filter(hv, hm), f"string filter {dt} n={n} off={off}")
        k = max(1, n // 2)
        hi = HostArray(A.UInt32, rng.integers(0, n, k).astype(np.uint32), rng.random(k) < 0.9)
        _eq(K.take(dv, hi.to_device()), oracle.take(hv, hi), f"string take {dt} n={n} off={off}")
    hf, df = _arr(rng, A.Float64, n, off)
    hi64, di64 = _arr(rng, A.Int64, n, off)
    for dst in (A.Utf8, A.LargeUtf8):
        _eq(K.cast(df, dst), oracle.cast(hf, dst), f"cast Float64->{dst} n={n} off={off}")
        _eq(K.cast(di64, dst), oracle.cast(hi64, dst), f"cast Int64->{dst} n={n} off={off}")
    _eq(K.cast_chain(di64, [A.Float64, A.LargeUtf8]), oracle.cast(oracle.cast(hi64, A.Float64), A.LargeUtf8), f"cast chain n={n} off={off}")


@pytest.mark.parametrize("rows", [1, 63, 4097, 8192])
def test_exact_fit_coalescer(ctx, rows):
    rng = np.random.default_rng(rows)
    nb = 70
    vals = rng.integers(-2**62, 2**62, nb * rows, dtype=np.int64)
    valid = rng.random(nb * rows) < 0.9
    keep = rng.random(nb * rows) < 0.3
    co = K.BatchCoalescer.new(["a"], [A.Int64], 8192, ctx)
    pairs = []
    for i in range(nb):
        s = slice(i * rows, (i + 1) * rows)
        col = HostArray(A.Int64, vals[s], valid[s]).to_device()  # every batch its own exact-fit buffers
        pairs.append((A.RecordBatch(["a"], [col], rows), HostArray(A.Boolean, keep[s]).to_device()))
    co.push_batches_with_filters(pairs)
    co.finish_buffered_batch()
    got_v, got_m = [], []
    for b in co.next_completed_batches():
        h = HostArray.from_device(b.columns[0])
        got_v.append(np.asarray(h.values)[:b.num_rows()])
        got_m.append(np.ones(b.num_rows(), bool) if h.valid is None else np.asarray(h.valid)[:b.num_rows()])
    got_v, got_m = np.concatenate(got_v), np.concatenate(got_m)
    assert np.array_equal(got_m, valid[keep])
    assert np.array_equal(got_v[got_m], vals[keep][got_m])


# ------------------------------------------------------------------------------------------------ the guard-page child run
GUARD_FILES = ["test_gpu_guard.py", "test_gpu_parity.py", "test_gpu_filter_sparse.py", "test_gpu_filter_small.py", "test_gpu_filter_expr.py",
               "test_gpu_aggregate.py", "test_gpu_selection.py", "test_gpu_deferred.py", "test_gpu_cdata.py", "test_gpu_ipc.py",
               "test_gpu_interleave.py", "test_gpu_zip.py", "test_gpu_rank_shift.py"]
# not in the guard run: multi-process tests (a mapping made with the virtual-memory API cannot be shared through hipIpc) and the
# bench contract (timing)
GUARD_SUBSET = "not two_ranks and not bench_json"


def guard_child(extra_env=None, files=GUARD_FILES, subset=GUARD_SUBSET, timeout=2400):
    env = dict(os.environ, AH_DEBUG_GUARD="1", AH_GUARD_CHILD="1")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "pytest"] + [os.path.join(ROOT, "tests", f) for f in files] + \
          ["-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k", f"({subset}) and not under_guard_pages and not under_redzones"]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.timeout(2700)
@pytest.mark.skipif(os.environ.get("AH_GUARD_CHILD") == "1", reason="already inside the guard-page child run")
def test_parity_suite_under_guard_pages():
    r = guard_child()
    out = r.stdout + r.stderr
    last = [l for l in out.splitlines() if l.startswith("[ah-test]")]
    tail = (last[-1:] or ["<no test id>"], r.stdout[-2500:], r.stderr[-2500:])
    assert "Memory access fault" not in out, tail
    assert r.returncode == 0, tail
    assert "passed" in r.stdout, tail
    print("guard-page child run:", [l for l in r.stdout.splitlines() if " passed" in l][-1])
