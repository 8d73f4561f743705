#!/usr/bin/env python3
"""Static scan of the gfx950 ISA of the library's kernels (no GPU needed): for every kernel of the given translation units —
vector loads, how many of them have an `s_waitcnt vmcnt(0)` between them and the previous load (one load in flight per wave
there), the longest run of loads issued back to back, scalar / vector instruction counts, LDS instructions.

    python tools/isa_scan.py arrow-rs_amd/csrc/filter_expr.hip arrow-rs_amd/csrc/cmp.hip [--filter compare_kernel] [--keep /tmp/asm]

This is the method behind profiles/r04_isa_pass.md: the counts are STATIC (tail branches included), so a high "serialised"
figure is a pointer to read that kernel's loop, not a verdict.  tests/test_kernel_isa.py pins the properties the round-4
fixes established (loads of the lazy predicate's count pass in flight together; compare's word assembly off the scalar unit).
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LOADS = ("global_load", "buffer_load", "flat_load", "scratch_load")
NOT_ALU = ("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_setpc", "s_sleep")


def assemble(src, outdir):
    out = os.path.join(outdir, os.path.splitext(os.path.basename(src))[0] + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "arrow-rs_amd", "csrc"), "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else names


def kernels(asm_path):
    """-> {mangled name: [instruction lines]} for every kernel (a function whose body contains s_endpgm); the body runs to the
    function's end label, not to the first s_endpgm (early exits end the program too)"""
    txt = open(asm_path).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M):
        if "s_endpgm" not in m.group(2):
            continue
        out[m.group(1)] = [l.strip() for l in m.group(2).splitlines() if l.strip() and not l.strip().startswith((";", "."))]
    return out


def stats(lines):
    loads = [i for i, l in enumerate(lines) if l.startswith(LOADS)]
    waits0 = [i for i, l in enumerate(lines) if l.startswith("s_waitcnt") and "vmcnt(0)" in l]
    serialised = sum(1 for a, b in zip(loads, loads[1:]) if any(a < w < b for w in waits0))
    run = best = 0
    for l in lines:  # longest run of loads with no wait on the vector memory counter in between
        if l.startswith(LOADS):
            run += 1
            best = max(best, run)
        elif l.startswith("s_waitcnt") and "vmcnt" in l:
            run = 0
    return {"loads": len(loads), "serialised": serialised, "longest_load_run": best,
            "salu": sum(1 for l in lines if l.startswith("s_") and not l.startswith(NOT_ALU)),
            "valu": sum(1 for l in lines if l.startswith("v_")),
            "lds": sum(1 for l in lines if l.startswith("ds_")), "instructions": len(lines)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sources", nargs="+")
    ap.add_argument("--filter", default="", help="only kernels whose demangled name contains this")
    ap.add_argument("--keep", default=None, help="directory to keep the .s files in")
    a = ap.parse_args()
    outdir = a.keep or tempfile.mkdtemp(prefix="ah_isa_")
    os.makedirs(outdir, exist_ok=True)
    rows = []
    for src in a.sources:
        ks = kernels(assemble(src, outdir))
        names = list(ks)
        for mangled, pretty in zip(names, demangle(names)):
            if a.filter in pretty:
                rows.append((pretty, stats(ks[mangled])))
    rows.sort(key=lambda r: -r[1]["serialised"])
    print(f"{'serialised/loads':>16} {'run':>4} {'salu':>6} {'valu':>6} {'lds':>5}  kernel")
    for pretty, s in rows:
        print(f"{s['serialised']:>7}/{s['loads']:<8} {s['longest_load_run']:>4} {s['salu']:>6} {s['valu']:>6} {s['lds']:>5}  {pretty[:140]}")


if __name__ == "__main__":
    sys.exit(main())
