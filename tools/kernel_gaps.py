#!/usr/bin/env python
"""Idle time on the GPU timeline of a rocprofv3 --kernel-trace run: for the LAST `--steps` repetitions of a kernel
pattern, how much wall time lies between the end of one kernel and the start of the next, by the pair of kernels around
the gap.  usage: kernel_gaps.py <..._kernel_trace.csv> [--tail-ms 50] [--top 15]
Answers "is the step's non-kernel time idle GPU (host waits) or many small launch boundaries?" (round 4, coalescer)."""
import argparse
import collections
import csv
import re

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--tail-ms", type=float, default=40.0, help="analyse the last N ms of GPU activity (the timed steps)")
ap.add_argument("--top", type=int, default=15)
a = ap.parse_args()


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    return name.split("(")[0].split("<")[0]


rows = []
for r in csv.DictReader(open(a.csv)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
t_end = rows[-1][1]
rows = [r for r in rows if r[0] >= t_end - a.tail_ms * 1e6]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = collections.defaultdict(lambda: [0, 0.0, 0.0])
prev_end, prev_name = rows[0][1], rows[0][2]
for s, e, n in rows[1:]:
    g = max(0, s - prev_end)
    k = (prev_name, n)
    gaps[k][0] += 1
    gaps[k][1] += g
    gaps[k][2] = max(gaps[k][2], g)
    if e > prev_end:
        prev_end, prev_name = e, n
idle = sum(v[1] for v in gaps.values())
print(f"window {span/1e6:.3f} ms: {len(rows)} kernels, busy {busy/1e6:.3f} ms, idle {idle/1e6:.3f} ms ({100*idle/span:.1f} %)")
print("| after kernel | before kernel | gaps | total us | avg us | max us |")
print("|---|---|---|---|---|---|")
for (p, n), (c, tot, mx) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print(f"| {p} | {n} | {c} | {tot/1e3:.1f} | {tot/c/1e3:.2f} | {mx/1e3:.1f} |")
by_kernel = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in rows:
    by_kernel[n][0] += 1
    by_kernel[n][1] += e - s
print("\n| kernel | launches | total us | avg us |")
print("|---|---|---|---|")
for n, (c, tot) in sorted(by_kernel.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print(f"| {n} | {c} | {tot/1e3:.1f} | {tot/c/1e3:.2f} |")
