#!/bin/bash
# Collects one round's rocprofv3 / bench evidence for profiles/ (run on the GPU box via gpurun):
#   usage: bash tools/collect_profiles.sh <round tag, e.g. r04>      -> gpurun_out/profiles_<tag>/, then
#          python tools/profile_summary.py <tag> gpurun_out/profiles_<tag>   (in the build container) writes profiles/<tag>_*
# the default bench line (all single-GPU configs + next rows + reference bench shapes + in-run PMC traffic + crossover + CPU
# baselines; FULL-DETAIL object via --detail-json, the compact stdout line beside it), a kernel trace of that same command,
# per-workload lines with in-run PMC traffic, traces of the coalescer / string filter / lazy predicate steps with the
# GPU-timeline gap analysis, and the world-1 exchange through the C ABI (filter_take and configs[4]).
# (One parametrised script since round 4; rounds 1-3 had a copy per round — see the git history.)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r05}
O=gpurun_out/profiles_$R
mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 --detail-json $O/bench_default.json > $O/bench_default.compact.json 2> $O/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off --detail-json "" > $O/bench_trace.json 2> $O/trace.log
for wl in arith cmp cast cast_string cast_chain; do
  timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --detail-json $O/bench_$wl.json > /dev/null 2> $O/bench_$wl.err
done
for wl in coalesce string_filter string_take predicate_filter predicate_filter_fused; do  # roofline over ALL launches of a step, PMC traffic summed per step
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --detail-json $O/bench_$wl.json > /dev/null 2> $O/bench_$wl.err
done
timeout 300 python bench.py --workload record_batch --steps 3 --warmup 1 --no-cpu-baseline --pmc-traffic off --detail-json $O/bench_record_batch.json > /dev/null 2> $O/bench_record_batch.err
for wl in coalesce string_filter predicate_filter_fused; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$wl -o bench -- python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off --detail-json "" > /dev/null 2> $O/trace_$wl.log
  python tools/kernel_gaps.py $(find $O/trace_$wl -name "*kernel_trace.csv" | head -1) --tail-ms 18 > $O/gaps_$wl.md 2>/dev/null
  find $O/trace_$wl -name "*kernel_trace.csv" -delete   # (tens of MB; the stats CSV and the gap table are what is kept)
done
find $O/trace -name "*kernel_trace.csv" -delete
timeout 300 python bench.py --reassemble allgatherv --steps 5 --warmup 2 --no-cpu-baseline --config-steps 3 --pmc-traffic off --detail-json $O/bench_exchange_world1.json > /dev/null 2> $O/bench_exchange_world1.err
AH_WAIT=block timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --pmc-traffic off --detail-json $O/bench_wait_block.json > /dev/null 2>&1
ls $O
