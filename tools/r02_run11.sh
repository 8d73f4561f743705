#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run11; mkdir -p $O
L=$GRAFT_REPO_ROOT/arrow-rs_amd/lib
B="python bench.py --workload cast_string --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off"
for v in base XG_NOLAYOUT XG_NOSTORE; do
  if [ $v = base ]; then E=""; else E="AH_LIB_PATH=$L/libarrow_hip_$v.so"; fi
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$v -o b -- $B > $O/$v.json 2> $O/$v.err
  f=$(find $O/tr_$v -name "b_kernel_stats.csv" | head -1)
  echo "== $v"; head -8 "$f" | cut -c1-160
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o b -- $B > $O/pmc_$c.json 2> $O/pmc_$c.err
done
python - <<'PY'
import csv,glob,collections
for c in ["FETCH_SIZE","WRITE_SIZE"]:
    fs=glob.glob(f"gpurun_out/r02_run11/pmc_{c}/**/b_counter_collection.csv",recursive=True)
    if not fs: print("no",c); continue
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"][:70]; agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
    for k,(n,v) in sorted(agg.items(), key=lambda x:-x[1][1])[:8]:
        print(c, k, n, "per launch KB-units:", round(v/n))
PY
