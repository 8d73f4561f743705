#!/bin/bash
# round-2 GPU call 2: new tests (full-size parity, comm, ipc hardening), the full default bench line, X2 diagnosis
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run2
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.json; tail -5 $O/bench_default.err
python bench.py --workload cast_string --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/bench_cast_string.json 2> $O/bench_cast_string.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cs -o b -- python bench.py --workload cast_string --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off > /dev/null 2> $O/trace_cs.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $O/sq_cs -o b -- python bench.py --workload cast_string --steps 2 --warmup 1 --no-cpu-baseline --pmc-traffic off > /dev/null 2> $O/sq_cs.log
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $O/sq2_cs -o b -- python bench.py --workload cast_string --steps 2 --warmup 1 --no-cpu-baseline --pmc-traffic off > /dev/null 2> $O/sq2_cs.log
rocprofv3 --pmc TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/pmc_sz -o p -- tools/gather_probe2 1 > $O/pmc_sz.log 2>&1
python bench.py --reassemble allgatherv --steps 3 --warmup 1 --no-cpu-baseline --no-configs --pmc-traffic off > $O/bench_comm1.json 2> $O/bench_comm1.err
tail -c 600 $O/bench_comm1.json; tail -3 $O/bench_comm1.err
du -sh $O
