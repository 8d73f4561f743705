#!/bin/bash
# round-2 GPU call 1: regressions, host-gap check (spin vs block waits), gather probe + PMC calibration
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run1
mkdir -p $O
nproc > $O/nproc.txt; free -g >> $O/nproc.txt
timeout 900 python -m pytest tests -m gpu -x -q -n 4 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_spin.json 2> $O/bench_spin.err
AH_WAIT=block python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_block.json 2> $O/bench_block.err
python bench.py --workload cast --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cast.json 2> $O/bench_cast.err
python bench.py --workload arith --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_arith.json 2> $O/bench_arith.err
head -c 1500 $O/bench_spin.json; echo; head -c 600 $O/bench_block.json; echo
tools/gather_probe2 5 > $O/probe.txt 2>&1
cat $O/probe.txt
rocprofv3 -L > $O/counters.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- tools/gather_probe2 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- tools/gather_probe2 1 > $O/pmc_write.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -o p -- tools/gather_probe2 1 > $O/pmc_tcc.log 2>&1
rocprofv3 --pmc TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $O/pmc_tcc2 -o p -- tools/gather_probe2 1 > $O/pmc_tcc2.log 2>&1
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $O/hiptrace -o b -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_hiptrace.json 2> $O/hiptrace.log
find $O -name "*.csv" | head -30
du -sh $O
