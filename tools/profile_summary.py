#!/usr/bin/env python
"""Turns a tools/collect_profiles.sh output directory into the markdown + CSV evidence
committed under profiles/.  usage: profile_summary.py gpurun_out/profiles_r01 r01"""
import collections
import csv
import json
import os
import re
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    return m.group(1) if m else name[:60]


def bench_line(path):
    if not os.path.exists(path):
        return None
    for l in open(path):
        if l.startswith("{"):
            return json.loads(l)
    return None


out = [f"# rocprofv3 evidence, round {tag} (MI355X, ROCm 7.2)\n",
       "Collected by `tools/collect_profiles.sh` (kernel-trace + stats; FETCH_SIZE and WRITE_SIZE each in their own",
       "PMC pass).  Times in µs.  FETCH_SIZE/WRITE_SIZE are KB as reported; per MI355X_MICROARCH.md §HBM, FETCH_SIZE of a",
       "wide coalesced stream reads exactly half the bytes (doubled in the `HBM bytes` column for streaming kernels,",
       "NOT for the random gather whose 128-B line fills are tallied in full); WRITE_SIZE is calibrated",
       "(gen_i64 writes 8.0e9 B and reports 7 812 500 KB).\n"]

for sub, title in [("trace", "filter + take step (bench.py default)"), ("trace_arith", "add_wrapping f64"),
                   ("trace_cmp", "lt f64"), ("trace_cast", "cast Int64->Float64"),
                   ("trace_cast_string", "cast Float64->LargeUtf8"),
                   ("trace_coalesce", "BatchCoalescer.push_batch_with_filter (2 columns, 2^24-row batches)"),
                   ("trace_string_filter_take", "filter + take on a LargeUtf8 column (2^27 rows)"),
                   ("trace_aggregate", "sum + min + max of an Int64 column (1e9 rows, 10 % nulls)"),
                   ("trace_sort", "sort_to_indices of a full-range Int64 column (2^29 rows, 10 % nulls)")]:
    p = os.path.join(src, sub, "bench_kernel_stats.csv")
    if not os.path.exists(p):
        continue
    shutil.copy(p, os.path.join(dst, f"{tag}_{sub}_kernel_stats.csv"))
    out.append(f"## {title}\n")
    b = bench_line(os.path.join(src, f"bench_{sub.replace('trace_', '') if sub != 'trace' else 'trace'}.json"))
    if b:
        out.append(f"bench line under the profiler: value {b['value']} {b['unit']}, {b['ms_per_step']} ms/step, "
                   f"roofline {json.dumps(b['roofline'])}\n")
    out.append("| kernel | calls | avg µs | min µs | max µs | % |")
    out.append("|---|---|---|---|---|---|")
    for r in csv.DictReader(open(p)):
        if float(r["Percentage"]) < 0.05:
            continue
        out.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
                   f"{float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    out.append("")

pm = {}
for sub, cname in [("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")]:
    p = os.path.join(src, sub, "bench_counter_collection.csv")
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        if k.startswith("take_kernel"):  # random-index launches (>2.5 ms) vs the sorted-index extra launches
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            k += " [random idx]" if dur > 2.5 else " [sorted idx]"
        agg[k].append(float(r["Counter_Value"]))
    pm[cname] = {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}
    with open(os.path.join(dst, f"{tag}_{cname}_per_kernel.csv"), "w") as f:
        f.write("kernel,dispatches,avg_value_KB\n")
        for k, (n, a) in sorted(pm[cname].items(), key=lambda kv: -kv[1][1]):
            f.write(f"\"{k}\",{n},{a:.1f}\n")
if pm:
    out.append("## HBM traffic per launch (PMC)\n")
    out.append("| kernel | FETCH_SIZE KB | WRITE_SIZE KB | HBM bytes per launch (corrected) | note |")
    out.append("|---|---|---|---|---|")
    traffic = {}
    for k, key in [("filter_scatter_kernel<8, 2, true, true>", "filter_scatter"),
                   ("take_kernel<8, unsigned int, true, 4> [random idx]", "take_gather"),
                   ("take_kernel<8, unsigned int, true, 4> [sorted idx]", "take_gather_sorted"),
                   ("filter_count_kernel", "filter_count")]:
        f = pm.get("FETCH_SIZE", {}).get(k, (0, 0))[1]
        w = pm.get("WRITE_SIZE", {}).get(k, (0, 0))[1]
        if "random" in k:
            total = f * 1024 + w * 1024
            note = "8-byte random gathers: each fills one 128-B line, tallied in full (no x2)"
        else:
            total = 2 * f * 1024 + w * 1024
            note = "wide coalesced reads: FETCH x2 (guide correction)"
        traffic[key] = round(total)
        out.append(f"| {k} | {f:.0f} | {w:.0f} | {total/1e9:.2f} GB | {note} |")
    out.append("")
    json.dump({"source": f"profiles/{tag}_FETCH_SIZE_per_kernel.csv + {tag}_WRITE_SIZE_per_kernel.csv",
               "workload": "bench.py default (1e9 Int64 rows, 10% nulls, 10% selectivity, 1e8 random u32 indices)",
               "hbm_bytes_per_launch": traffic}, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
# streaming kernels of the other workloads: FETCH x2 + WRITE
extra = []
for wl, kern in [("arith", "arith_kernel"), ("cmp", "compare_kernel"), ("aggregate", "agg_kernel")]:
    vals = {}
    for sub, cname in [(f"fetch_{wl}", "FETCH_SIZE"), (f"write_{wl}", "WRITE_SIZE")]:
        p = os.path.join(src, sub, "bench_counter_collection.csv")
        if os.path.exists(p):
            xs = [float(r["Counter_Value"]) for r in csv.DictReader(open(p)) if kern in r["Kernel_Name"]]
            if xs:
                vals[cname] = sum(xs) / len(xs)
    if "FETCH_SIZE" in vals:
        vals.setdefault("WRITE_SIZE", 0.0)
        tot = 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024
        extra.append(f"| {kern} ({wl}, 1e9 rows) | {vals['FETCH_SIZE']:.0f} | {vals['WRITE_SIZE']:.0f} | {tot/1e9:.2f} GB | wide coalesced reads: FETCH x2 |")
if extra:
    out.append("## HBM traffic of the other streaming kernels (PMC)\n")
    out.append("| kernel | FETCH_SIZE KB | WRITE_SIZE KB | HBM bytes per launch (corrected) | note |")
    out.append("|---|---|---|---|---|")
    out += extra
    out.append("")
p = os.path.join(src, "sq", "bench_counter_collection.csv")
if os.path.exists(p):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        if k.startswith(("filter_scatter", "take_kernel", "filter_count")):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.append("## SQ counters, filter + take step (quad-cycles summed over waves)\n")
    names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
             "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"]
    out.append("| kernel | " + " | ".join(n.replace("SQ_", "") for n in names) + " |")
    out.append("|---|" + "---|" * len(names))
    for k, v in agg.items():
        out.append(f"| {k} | " + " | ".join(f"{sum(v[n])/max(len(v[n]),1):.3g}" if n in v else "-" for n in names) + " |")
    out.append("")
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(out) + "\n")
nr = os.path.join(src, "next_rows.json")
if os.path.exists(nr):
    for l in open(nr):
        if l.startswith("{"):
            d = json.loads(l)
            with open(os.path.join(dst, f"{tag}_summary.md"), "a") as fsum:
                fsum.write("\n## SURVEY §8f rows without a bench workload (tools/next_rows_time.py, wall clock of the "
                           "synchronous C-ABI call, median of 5)\n\n| quantity | value |\n|---|---|\n")
                for k, v in d.items():
                    fsum.write(f"| {k} | {v} |\n")
            shutil.copy(nr, os.path.join(dst, f"{tag}_next_rows.json"))
for f in ["bench_plain.json"]:
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
print("\n".join(out))
