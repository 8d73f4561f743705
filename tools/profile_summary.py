#!/usr/bin/env python
"""Turns a tools/collect_profiles.sh output directory into the evidence committed under profiles/ (one parametrised tool
since round 4; the per-round copies of rounds 1-3 are in the git history).
usage: python tools/profile_summary.py <round tag, e.g. r04> gpurun_out/profiles_<tag>"""
import csv
import json
import os
import re
import shutil
import sys

R, src = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)


def line_of(name):
    p = os.path.join(src, name)
    if not os.path.exists(p):
        return None
    for l in open(p):
        if l.startswith("{"):
            return json.loads(l)
    return None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    return m.group(1) if m else name[:60]


out = [f"# rocprofv3 / bench evidence, round {R} (one MI355X, ROCm 7.2; builder-side gpurun box)\n",
       "Produced by `tools/collect_profiles.sh` + `tools/profile_summary.py`.  Kernel times: HIP events recorded by the",
       "library on its launch stream (`kernel_avg_ms`), cross-checked by the rocprofv3 kernel trace below.  `traffic` = HBM",
       "bytes per launch from rocprofv3 PMC passes run INSIDE the bench invocation (FETCH_SIZE and WRITE_SIZE in separate",
       "passes; FETCH_SIZE doubled: this rocprofv3 tallies every 128-byte L2 line fill at 64 bytes — calibrated with",
       "TCC_EA0_RDREQ_128B in `r02_take_ablation.md`).  `frac` = algorithmic bytes (SURVEY §8d) / kernel time / 8 TB/s.\n"]

d = line_of("bench_default.json")
if d:
    shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, f"{R}_bench_default.json"))
    rf, fs = d["roofline"], d.get("roofline_filter_scatter", {})
    out.append("## The default line: `python bench.py --steps 20 --warmup 5` (configs[1] + configs[2] + configs[3] + the SURVEY 8f rows)\n")
    out.append(f"value **{d['value']} Mrows/s**, {d['ms_per_step']} ms per step, host_gap_ms {d['host_gap_ms']} "
               f"(step time minus its profiled kernels), kernels {d['kernel_avg_ms']}.\n")
    out.append("| kernel / config | avg ms | algorithmic GB per launch | GB/s | frac of 8 TB/s | PMC traffic GB per launch | traffic frac |")
    out.append("|---|---|---|---|---|---|---|")

    def row(name, r):
        tr = r.get("traffic")
        out.append(f"| {name} | {r['avg_launch_ms']} | {r['algorithmic_bytes_per_launch'] / 1e9:.3f} | {r['achieved']} | {r['frac']} | "
                   f"{tr / 1e9:.2f} | {r.get('traffic_frac')} |" if tr else
                   f"| {name} | {r['avg_launch_ms']} | {r['algorithmic_bytes_per_launch'] / 1e9:.3f} | {r['achieved']} | {r['frac']} | — | — |")
    if fs:
        row("filter_scatter (configs[1] filter)", fs)
    row(f"{rf['kernel']} (configs[1] take, 1e8 random u32 indices)", rf)
    for k, v in d.get("configs", {}).items():
        if "roofline" in v:
            row(f"{v['roofline']['kernel']} ({k}: {v['rows']} rows, {v['ms']} ms per call)", v["roofline"])
    for k, v in d.get("next_rows", {}).items():
        if "roofline" in v:
            extra_ms = f", {v['ms_without_kernel_events']} ms without per-kernel events" if "ms_without_kernel_events" in v else ""
            row(f"{v['roofline']['kernel']} (next_rows.{k}: {v['rows']} rows, {v['ms']} ms per step{extra_ms}; all launches of a step)"
                if k != "record_batch" else f"{v['roofline']['kernel']} (next_rows.{k}: {v['rows']} rows, {v['ms']} ms per call)", v["roofline"])
    for k, v in d.get("configs_narrow", {}).items():  # round 5: 4-byte and narrower operands
        if "roofline" in v:
            row(f"{v['roofline']['kernel']} (configs_narrow.{k}: {v['rows']} rows, {v['ms']} ms per call)", v["roofline"])
    sf = d.get("next_rows", {}).get("string_filter", {}).get("roofline", {})
    if "frac_survey_rule" in sf:
        out.append(f"\nstring_filter both ways: selected rows' bytes only {sf['algorithmic_bytes_per_launch'] / 1e9:.3f} GB -> frac {sf['frac']}; "
                   f"whole text buffer as an input (SURVEY 8d) {sf['algorithmic_bytes_survey_rule'] / 1e9:.3f} GB -> frac {sf['frac_survey_rule']}.\n")
    cs = d.get("coalesce_by_batch_rows", {})
    if cs.get("points"):
        out.append("\n### BatchCoalescer by input batch size (`coalesce_by_batch_rows`; the reference's operating point is 8192-row batches)\n")
        out.append(cs.get("what", "") + "\n")
        out.append("| batch rows x target | ms per 1e9 rows | Mrows/s | frac | pushes | output batches | launches (count / scatter) | kernel ms (count + scatter) |")
        out.append("|---|---|---|---|---|---|---|---|")
        for k, v in cs["points"].items():
            if "error" in v:
                out.append(f"| {k} | error: {v['error']} |")
                continue
            kl, km = v.get("kernel_launches", {}), v.get("kernel_ms", {})
            out.append(f"| {k} (target {v['target']}) | {v['ms']} | {v['value']} | {v['frac']} | {v['pushes']} | {v['output_batches']} | "
                       f"{kl.get('filter_count')} / {kl.get('filter_scatter')} | {km.get('filter_count')} + {km.get('filter_scatter')} |")
        out.append(f"\none CPU core (oracle filter of both columns, batch by batch): {cs.get('cpu_1core_Mrows_per_s_by_batch_rows')} Mrows/s; cores granted: {cs.get('cpu_cores_granted')}\n")
    if "requests" in rf:
        out.append(f"\ntake_gather in line fills: {rf['requests']}\n")
    if d.get("roofline_take_sorted"):
        out.append(f"take with sorted indices (positions of the predicate): {d['roofline_take_sorted']}\n")
    if d.get("crossover_rows"):
        out.append(f"crossover_rows: {d['crossover_rows']}\n")
    if d.get("filter_by_selectivity"):
        out.append(f"filter_scatter at other selectivities of the same column (the sparse path below 3 %): {d['filter_by_selectivity']}\n")
    out.append(f"take variants: sorted indices {d.get('take_sorted_indices_ms')} ms, 10 % null indices {d.get('take_null_indices_ms')} ms.\n")
    cb = d.get("cpu_baseline", {})
    out.append(f"cpu_baseline (oracle = scalar port of the reference, same box): {cb.get('value')} Mrows/s on 1 core; all cores: "
               f"{cb.get('all_cores')}; Arrow C++ sanity: {cb.get('arrow_cpp_sanity')}.\n")

p = os.path.join(src, "trace", "bench_kernel_stats.csv")
if os.path.exists(p):
    shutil.copy(p, os.path.join(dst, f"{R}_trace_kernel_stats.csv"))
    out.append("## rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 3` (same workloads; times in µs)\n")
    out.append("| kernel | calls | avg µs | min µs | max µs | % |")
    out.append("|---|---|---|---|---|---|")
    for r in csv.DictReader(open(p)):
        if float(r["Percentage"]) < 0.04:
            continue
        out.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
                   f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    out.append("\n(`take_kernel`'s average mixes the random-index launches of the step, 3.7–3.9 ms, with the sorted-index "
               "launches of the extra measurement, 1.5 ms.)\n")

out.append("## Per-workload lines with in-run PMC traffic\n")
out.append("| workload | ms per call (without per-kernel events) | roofline kernel | frac | traffic_frac | PMC traffic per launch / step (GB) | host_gap_ms |")
out.append("|---|---|---|---|---|---|---|")
traffic = {}
for wl in ["arith", "cmp", "cast", "cast_string", "cast_chain", "coalesce", "record_batch", "string_filter", "string_take", "predicate_filter",
           "predicate_filter_fused"]:
    d2 = line_of(f"bench_{wl}.json")
    if not d2:
        continue
    tr = d2.get("pmc_traffic_bytes_per_launch")
    traffic[wl] = tr
    out.append(f"| {wl} | {d2['ms_per_step']} ({d2.get('ms_per_step_without_kernel_events', '—')}) | {d2['roofline']['kernel']} | "
               f"{d2['roofline']['frac']} | {d2['roofline'].get('traffic_frac', '—')} | "
               f"{ {k: round(v / 1e9, 2) for k, v in tr.items()} if isinstance(tr, dict) else '—'} | {d2.get('host_gap_ms')} |")
    shutil.copy(os.path.join(src, f"bench_{wl}.json"), os.path.join(dst, f"{R}_bench_{wl}.json"))
out.append("")

for tag, title in (("trace_coalesce", "coalesce (BatchCoalescer, 60 batches of 2^24 rows per step, pushed 8 at a time)"),
                   ("trace_string_filter", "string_filter (LargeUtf8 column, 2^27 rows, 10 % selected)"),
                   ("trace_predicate_filter_fused", "predicate_filter_fused (ah_filter_expr: WHERE a < 0 AND b >= 0.0, 1e9 rows)")):
    p2 = os.path.join(src, tag, "bench_kernel_stats.csv")
    if os.path.exists(p2):
        shutil.copy(p2, os.path.join(dst, f"{R}_{tag}_kernel_stats.csv"))
        out.append(f"## rocprofv3 --kernel-trace --stats: {title}; times in µs\n")
        out.append("| kernel | calls | avg µs | min µs | % |")
        out.append("|---|---|---|---|---|")
        for r in csv.DictReader(open(p2)):
            if float(r["Percentage"]) < 0.5:
                continue
            out.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
                       f"{float(r['Percentage']):.2f} |")
        out.append("")

for wl, title in (("coalesce", "coalesce"), ("string_filter", "string_filter"), ("predicate_filter_fused", "predicate_filter_fused")):
    p3 = os.path.join(src, f"gaps_{wl}.md")
    if os.path.exists(p3) and os.path.getsize(p3):
        shutil.copy(p3, os.path.join(dst, f"{R}_gaps_{wl}.md"))
        out.append(f"## GPU timeline of the `{title}` step under rocprofv3 (tools/kernel_gaps.py: idle time between kernels, by the pair around the gap)\n")
        out.append(open(p3).read())
if d and isinstance(d.get("reference_bench_shapes"), dict) and "shapes" in d["reference_bench_shapes"]:
    rs = d["reference_bench_shapes"]
    out.append("## The reference's own criterion shapes through the raw C ABI (`reference_bench_shapes`; microseconds per call)\n")
    out.append(rs["what"] + "\n")
    out.append("| shape | rows | synchronous call | per call in a batch of %d | per call in a replayed hipGraph of %d | 1-core oracle | batched / graph beats one core |" % (rs["batch"], rs["batch"]))
    out.append("|---|---|---|---|---|---|---|")
    for k, v in rs["shapes"].items():
        out.append(f"| {k} | {v['rows']} | {v['sync_us']} | {v['batched_us'] if v['batched_us'] is not None else '— (synchronous by contract)'} | "
                   f"{v.get('graph_us') if v.get('graph_us') is not None else '—'} | {v['cpu_1core_us']} | {v['batched_beats_cpu'] if v['batched_beats_cpu'] is not None else v['sync_beats_cpu']} |")
    out.append("")
if d and d.get("hbm_pool"):
    out.append(f"HBM held by the pooled allocator over the default run (`ah_context_stats`): {d['hbm_pool']}\n")

ex = line_of("bench_exchange_world1.json")
if ex:
    out.append("## Exchange step through the C ABI at world 1 (real RCCL: ncclCommInitRank, count all-gather, group, merge)\n")
    out.append(f"`bench.py --reassemble allgatherv`: transport {ex['config'].get('transport')}, step {ex['ms_per_step']} ms with the "
               f"reassembly (ah_all_gather_columns_begin -> take on the second context -> _end) vs {ex.get('local_ms_per_step')} ms "
               f"without; last call {ex.get('reassemble_last_ms')}.  configs[4] (filter_record_batch + ah_all_gather_columns): "
               f"{ex.get('configs', {}).get('record_batch_allgather')}\n")
wb = line_of("bench_wait_block.json")
if wb and d:
    out.append("## Host waits: mailbox spin (default) vs `AH_WAIT=block` (hipStreamSynchronize), same box\n")
    out.append(f"spin: {d['ms_per_step']} ms per step, host_gap_ms {d['host_gap_ms']}; block: {wb['ms_per_step']} ms, host_gap_ms "
               f"{wb['host_gap_ms']}.  (The driver's round-1 box showed 2.39 ms of gap with blocking waits: interrupt wake-up "
               "latency is a property of the host, the spin path does not depend on it.)\n")

with open(os.path.join(dst, f"{R}_summary.md"), "w") as f:
    f.write("\n".join(out) + "\n")
if d and isinstance(d.get("pmc_traffic_bytes_per_launch"), dict):
    tj = {"source": f"bench.py in-run rocprofv3 PMC passes (profiles/{R}_bench_*.json)", "hbm_bytes_per_launch": dict(d["pmc_traffic_bytes_per_launch"])}
    for wl, tr in traffic.items():
        if isinstance(tr, dict):
            tj["hbm_bytes_per_launch"].update(tr)
    json.dump(tj, open(os.path.join(dst, f"{R}_traffic.json"), "w"), indent=1)
print("\n".join(out))
