for rep in 1 2; do
for slab in 0 1; do for mode in "1 0" "1 1" "8 1"; do set -- $mode
  echo "== slab=$slab group=$1 pipeline=$2 rep $rep: $(AH_COALESCE_SLAB=$slab AH_COALESCE_GROUP=$1 AH_COALESCE_PIPELINE=$2 python bench.py --workload coalesce --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_per_step_without_kernel_events'), d['kernel_avg_ms'])")"
done; done; done
