"""ISA-level regression guards (no GPU: hipcc cross-compiles gfx950 here).  Round 4 found three kernels whose SOURCE looked
fine and whose machine code was not (profiles/r04_isa_pass.md): a bounds test that the compiler turned into one load in
flight per wave, and bit-interleaves that ran on the CU's single scalar unit until it was the bottleneck.  These tests pin
the properties the fixes established, with generous margins — they are about the SHAPE of the code, not instruction counts
to the unit.  Method: tools/isa_scan.py (the same scan the profile note describes)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_scan  # noqa: E402

needs_hipcc = pytest.mark.skipif(not os.path.exists(isa_scan.HIPCC) or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")


def _kernels(tmp_path_factory, name):
    out = tmp_path_factory.mktemp("isa")
    ks = isa_scan.kernels(isa_scan.assemble(os.path.join(ROOT, "arrow-rs_amd", "csrc", name), str(out)))
    names = list(ks)
    return {pretty: ks[m] for m, pretty in zip(names, isa_scan.demangle(names))}


@pytest.fixture(scope="module")
def expr_kernels(tmp_path_factory):
    return _kernels(tmp_path_factory, "filter_expr.hip")


@pytest.fixture(scope="module")
def cmp_kernels(tmp_path_factory):
    return _kernels(tmp_path_factory, "cmp.hip")


def _one(kernels, part):
    hits = [(n, l) for n, l in kernels.items() if part in n]
    assert len(hits) == 1, [n for n, _ in hits]
    return hits[0][1]


@needs_hipcc
def test_lazy_predicate_count_pass_keeps_its_loads_in_flight(expr_kernels):
    lines = _one(expr_kernels, "filter_expr_count_kernel<2, 8, true>")
    # the eight 16-byte operand loads of four wave steps go out back to back: no wait on the vector memory counter between them
    run = best = 0
    for l in lines:
        if l.startswith("global_load_dwordx4"):
            run += 1
            best = max(best, run)
        elif l.startswith("s_waitcnt") and "vmcnt" in l:
            run = 0
    assert best >= 8, f"operand loads are separated by waits (longest run {best})"
    s = isa_scan.stats(lines)
    # the interleave of the ballots lives on the vector unit (once per chunk): the scalar unit is shared by four SIMDs.
    # 1 283 scalar instructions before the round-4 rewrite, ~670 after
    assert s["salu"] <= 900, s
    assert s["instructions"] <= 2600, s  # (3 587 before: the whole fast instantiation fits the instruction cache comfortably)


@needs_hipcc
@pytest.mark.parametrize("inst", ["compare_kernel<double, 2>", "compare_kernel<long, 2>", "compare_kernel<float, 4>"])
def test_compare_builds_its_output_words_on_the_vector_unit(cmp_kernels, inst):
    lines = _one(cmp_kernels, inst)
    s = isa_scan.stats(lines)
    # ballots staged in LDS, lanes 0 .. 4V - 1 interleave them: ds traffic present, scalar count far below the per-group
    # scalar spreads of rounds 1-3 (532 / 564 for the 8-byte kernels, ~1 300 for the 4-byte ones)
    assert s["lds"] >= 2, s
    assert s["salu"] <= (450 if "4>" in inst else 350), s


@pytest.fixture(scope="module")
def filter_kernels(tmp_path_factory):
    return _kernels(tmp_path_factory, "filter.hip")


@needs_hipcc
@pytest.mark.parametrize("inst,loads", [("filter_scatter_kernel<1, 16, true, false, 4>", 4), ("filter_scatter_kernel<2, 8, true, false, 2>", 4)])
def test_narrow_scatter_parks_whole_vectors_and_walks_set_bits(filter_kernels, inst, loads):
    """Round 5 (profiles/r05_narrow_filter.md): the 1- and 2-byte scatter was VALU-bound, not HBM-bound — ~14 vector instructions
    per ROW in the per-element form, every 16-byte load unpacked into byte registers behind a vmcnt(0).  The staged form
    loads 16-byte values back to back, parks them in LDS with ds_write_b128 and reads the selected rows back one by one."""
    lines = _one(filter_kernels, inst)
    run = best = 0
    for l in lines:
        if l.startswith("global_load_dwordx4"):
            run += 1
            best = max(best, run)
        elif l.startswith("s_waitcnt") and "vmcnt" in l:
            run = 0
    assert best >= loads, f"value loads separated by waits (longest run {best})"
    assert sum(1 for l in lines if l.startswith("ds_write_b128")) >= loads, "values are parked as whole 16-byte vectors"
    assert any(l.startswith("ds_read_u8" if "<1," in inst else "ds_read_u16") for l in lines), "selected rows are read back by row"
    assert any(l.startswith("v_ffbl_b32") for l in lines), "the compaction walks set bits"
    # round 6: a thread's run of selected rows leaves the walk as whole dwords (unaligned ds_write_b32, 4 / 2 values each) and its
    # validity bits as one shifted register OR-ed into an LDS bitmap laid out like the output words (no byte-per-row flags)
    assert any(l.startswith("ds_write_b32") for l in lines), "the walk stores packed dwords"
    assert any(l.startswith("ds_or_b32") for l in lines), "validity bits of a run are OR-ed into the LDS bitmap"
    s = isa_scan.stats(lines)
    # static size: the walk exists twice (tile staged by one pass / by windows) x two 32-bit halves (1 130 for the per-element Int8 form)
    assert s["instructions"] <= 1000, s


@needs_hipcc
def test_scatter_table_scan_uses_dpp(filter_kernels):
    """wave_scan_incl: six v_add_u32_dpp, not six ds_bpermute round trips (common.hpp)."""
    lines = _one(filter_kernels, "filter_scatter_kernel<8, 2, true, true, 1>")
    assert sum(1 for l in lines if "_dpp" in l and ("row_shr" in l or "row_bcast" in l)) >= 6


def test_window_tile_bound_is_a_superset_on_the_host(tmp_path):
    """csrc/window_tiles.hpp (the host-side bound that lets a cut batch of BatchCoalescer launch only the tiles its window of
    the filtered stream can lie in) against brute force: a bound one tile short would lose rows silently on the GPU.
    tests/cpp/window_tiles_host_test.cpp: 20 000 random selections x 24 windows each, all three tile sizes."""
    import subprocess
    exe = str(tmp_path / "window_tiles_host_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp", "window_tiles_host_test.cpp")],
                   check=True)
    for seed in ("1", "2"):
        r = subprocess.run([exe, "20000", seed], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "WINDOW_TILES_OK" in r.stdout, r.stdout + r.stderr
        checked, tight = [int(x) for x in r.stdout.split() if x.isdigit()][:2]
        assert checked == 480000 and tight > checked // 2, r.stdout  # (the bound really cuts in most cases)
