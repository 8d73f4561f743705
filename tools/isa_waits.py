#!/usr/bin/env python
"""Load / wait sequence of every kernel of one .hip file, from the gfx950 assembly (no GPU needed).

  L global or buffer load      S global store      A global atomic      (wN) s_waitcnt vmcnt(N)
  l s_load (scalar)            (k) s_waitcnt lgkmcnt(0)                 | s_barrier      b branch

A run of `L(w0)L(w0)L(w0)` at the head of a kernel is what round 2 went looking for: hipcc places the wait right
behind a load whose value is used at once (e.g. bv_fetch64's shift of the validity word), so loads that could
travel together go out one memory round trip at a time.  Found and fixed this way: string_len / string_write
(validity words through scalar loads), filter_scatter (bv_issue / bv_finish), the two filter_count kernels.

usage: python tools/isa_waits.py arrow-rs_amd/csrc/filter.hip [kernel-name-substring]"""
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.NamedTemporaryFile(suffix=".s") as f:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", f.name, src],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(f.name).read().split("\n")
name, seq, out = None, [], {}
for l in lines:
    m = re.match(r"^(_Z\S+):", l)
    if m:
        name, seq = m.group(1), []
        out[name] = seq
        continue
    if name is None:
        continue
    t = l.strip()
    if t.startswith(("global_load", "buffer_load")):
        seq.append("L")
    elif t.startswith("global_store"):
        seq.append("S")
    elif t.startswith("global_atomic"):
        seq.append("A")
    elif t.startswith("s_load"):
        seq.append("l")
    elif t.startswith("s_waitcnt"):
        v = re.search(r"vmcnt\((\d+)\)", t)
        if v:
            seq.append(f"(w{v.group(1)})")
        elif "lgkmcnt(0)" in t:
            seq.append("(k)")
    elif t.startswith("s_barrier"):
        seq.append("|")
    elif t.startswith("s_cbranch"):
        seq.append("b")
    elif t.startswith("s_endpgm"):
        name = None
for k, v in out.items():
    if len(v) > 3 and want in k:
        dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(dem.replace("(anonymous namespace)::", "")[:110])
        print("   ", "".join(v)[:400])
