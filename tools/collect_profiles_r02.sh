#!/bin/bash
# Collects the round-2 rocprofv3 evidence committed under profiles/ (run on the GPU box via gpurun):
# the default bench line (all single-GPU configs + in-run PMC traffic), a kernel trace of that same command,
# per-workload lines with in-run PMC traffic, the coalescer and the world-1 exchange through the C ABI.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_r02
mkdir -p $O
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/bench_trace.json 2> $O/trace.log
for wl in arith cmp cast cast_string; do
  python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
done
for wl in coalesce string_filter_take predicate_filter; do  # roofline over ALL launches of a step, PMC traffic summed per step
  python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
done
for wl in record_batch aggregate; do
  python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --pmc-traffic off > $O/bench_$wl.json 2> $O/bench_$wl.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cast_string -o bench -- python bench.py --workload cast_string --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/bench_cs_trace.json 2> $O/trace_cs.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_coalesce -o bench -- python bench.py --workload coalesce --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/bench_co_trace.json 2> $O/trace_co.log
python bench.py --reassemble allgatherv --steps 5 --warmup 2 --no-cpu-baseline --no-configs --pmc-traffic off > $O/bench_exchange_world1.json 2> $O/bench_exchange_world1.err
AH_WAIT=block python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --pmc-traffic off > $O/bench_wait_block.json 2> $O/bench_wait_block.err
ls $O
