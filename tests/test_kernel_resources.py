"""Register / LDS budgets of the hot kernels, read from the BUILT library (no GPU, no recompilation).

The occupancy these kernels run at is a design decision with a measured price: filter_scatter
keeps its tile in registers (<= 96 VGPRs = 5 workgroups per CU; forcing more spilled: 1.48 -> 1.85 ms), the two string
passes need <= 80 SGPRs and <= 20 KiB of LDS for 8 workgroups per CU (4.76 -> 4.28 ms).  A compiler or source change
that silently breaks one of these shows up here instead of as a slower bench line.

The code objects are pulled out of libarrow_hip.so's clang offload bundles and their AMDGPU metadata notes are read with
llvm-readelf."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "arrow-rs_amd", "lib", "libarrow_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _kernels():
    data = open(LIB, "rb").read()
    out = {}
    pos = data.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", data, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                with tempfile.NamedTemporaryFile(suffix=".co") as f:
                    f.write(data[pos + off:pos + off + size])
                    f.flush()
                    txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
                for blk in txt.split("- .agpr_count:")[1:] if "- .agpr_count:" in txt else txt.split("  - .args:")[1:]:
                    name = re.search(r"\.name:\s+(\S+)", blk)
                    if not name:
                        continue
                    get = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))  # noqa: E731
                    out[name.group(1)] = {"sgpr": get("sgpr_count"), "vgpr": get("vgpr_count"),
                                          "lds": get("group_segment_fixed_size"), "scratch": get("private_segment_fixed_size")}
        pos = data.find(MAGIC, pos + 1)
    return out


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(READELF):
        pytest.skip("llvm-readelf not in this image")
    k = _kernels()
    assert len(k) > 100, "no kernel metadata found in libarrow_hip.so"
    return k


def _find(kernels, *parts):
    hits = [v for n, v in kernels.items() if all(p in n for p in parts)]
    assert hits, f"no kernel matching {parts}"
    return hits


def test_no_hot_kernel_spills(kernels):
    for parts in (("filter_scatter_kernel",), ("take_kernel",), ("arith_kernel",), ("compare_kernel",), ("cast_stream_kernel",),
                  ("string_len_kernel",), ("string_write_kernel",), ("filter_count",), ("filter_small_kernel",),
                  ("filter_scatter_multi_kernel",), ("filter_expr_count_kernel",), ("string_filter_ranges_kernel",),
                  ("string_filter_gather_kernel",)):
        for r in _find(kernels, *parts):
            assert r["scratch"] == 0, f"{parts}: {r['scratch']} bytes of scratch (register spill)"


def test_filter_scatter_keeps_five_workgroups_per_cu(kernels):
    # Int64 values, two per lane, with validity, predicated loads: the configs[1] kernel.  512 VGPRs / 96 -> 5 waves per SIMD.
    for r in _find(kernels, "filter_scatter_kernelILi8ELi2ELb1ELb1E"):
        assert r["vgpr"] <= 96 and r["lds"] <= 32768, r


def test_string_passes_keep_eight_workgroups_per_cu(kernels):
    for r in _find(kernels, "string_write_kernelId"):
        assert r["sgpr"] <= 80 and r["lds"] <= 20480 and r["vgpr"] <= 64, r
    for r in _find(kernels, "string_len_kernelId"):
        assert r["sgpr"] <= 80 and r["lds"] <= 20480 and r["vgpr"] <= 64, r


def test_round3_kernels_keep_their_occupancy(kernels):
    # the lazy predicate's fast instantiation (2 terms, 8-byte operands, scalar right sides): 5 waves per SIMD, each with
    # eight 16-byte loads in flight (r04)
    for r in _find(kernels, "filter_expr_count_kernelILi2ELi8ELb1E"):
        assert r["vgpr"] <= 84, r
    # the multi-batch scatter shares filter_scatter's body: same budget
    for r in _find(kernels, "filter_scatter_multi_kernelILi8ELi2ELb1E"):
        assert r["vgpr"] <= 96 and r["lds"] <= 32768, r
    # the one-launch filter of small batches, Int64 with validity: a latency-bound kernel (<= 256 tiles); 4 waves per SIMD
    for r in _find(kernels, "filter_small_kernelILi8ELi2ELb1E"):
        assert r["vgpr"] <= 128 and r["lds"] <= 32768, r
    # the wave-per-tile scatter for sparse selections lives on occupancy: 8 waves per SIMD, no scratch, <= 3 KiB of LDS
    for r in _find(kernels, "filter_scatter_sparse_kernelILi8ELb1E"):
        assert r["vgpr"] <= 64 and r["lds"] <= 3072 and r["scratch"] == 0, r
    # string filter ranges: a 16 KiB stage (1024 selected rows per round) -> 7 workgroups per CU (registers)
    # (LargeUtf8: 8-byte pairs "Ill" and the 4-byte pairs "Ilj" that normally run — relative starts computed IN PLACE, or the
    #  compiler keeps 16 more values live: 86 VGPRs, 5 waves per SIMD)
    for r in _find(kernels, "string_filter_ranges_kernelIl", "Lb1ELb"):
        assert r["lds"] <= 20480 and r["vgpr"] <= 72, r
