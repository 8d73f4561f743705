#!/bin/bash
# Collects the round-3 rocprofv3 evidence committed under profiles/ (run on the GPU box via gpurun):
# the default bench line (all single-GPU configs + next rows + in-run PMC traffic + crossover + CPU baselines), a kernel
# trace of that same command, per-workload lines with in-run PMC traffic, traces of the coalescer / string filter / lazy
# predicate steps, and the world-1 exchange through the C ABI (filter_take and configs[4]).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_r03
mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/bench_trace.json 2> $O/trace.log
for wl in arith cmp cast cast_string; do
  timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
done
for wl in coalesce string_filter string_take predicate_filter predicate_filter_fused; do  # roofline over ALL launches of a step, PMC traffic summed per step
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
done
timeout 300 python bench.py --workload record_batch --steps 3 --warmup 1 --no-cpu-baseline --pmc-traffic off > $O/bench_record_batch.json 2> $O/bench_record_batch.err
for wl in coalesce string_filter predicate_filter_fused; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$wl -o bench -- python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/bench_${wl}_trace.json 2> $O/trace_$wl.log
done
timeout 300 python bench.py --reassemble allgatherv --steps 5 --warmup 2 --no-cpu-baseline --config-steps 3 --pmc-traffic off > $O/bench_exchange_world1.json 2> $O/bench_exchange_world1.err
ls $O
