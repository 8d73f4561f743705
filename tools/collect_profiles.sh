#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box via gpurun):
#   kernel-trace + stats, then FETCH_SIZE and WRITE_SIZE in their own passes (PMC never combined
#   with sys/runtime tracing).  Usage: bash tools/collect_profiles.sh <round-tag>
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/write.log
for wl in arith cmp aggregate; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$wl -o bench -- python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/fetch_$wl.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_$wl -o bench -- python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/write_$wl.log
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/sq.log
for wl in arith cmp cast cast_string coalesce string_filter_take aggregate sort; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl -o bench -- python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_$wl.json 2> $OUT/trace_$wl.log
done
python tools/next_rows_time.py > $OUT/next_rows.json 2> $OUT/next_rows.log
python bench.py --steps 10 --warmup 3 > $OUT/bench_plain.json 2> $OUT/bench_plain.log
find $OUT -name "*.csv" | head -40
