#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run19b; mkdir -p $O
B="python bench.py --workload cast_string --steps 2 --warmup 1 --no-cpu-baseline --pmc-traffic off"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  AH_BENCH_CAST_PURE=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o b -- $B > $O/p$i.json 2> $O/p$i.err
done
python - <<'PY'
import csv,glob,collections,re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r02_run19b/p*/**/b_counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r"(string_(len|write)_kernel)",r["Kernel_Name"])
        if not m: continue
        agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
n=2**29
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        a=sum(vals)/len(vals)
        print("   ",c, round(a), "per row: %.2f"%(a*64/n) if c.startswith("SQ_INSTS") else "")
PY
