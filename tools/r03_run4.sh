#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_filter_expr.py tests/test_gpu_filter_small.py -x -q -m gpu > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py -x -q -m gpu -k "filter or arith or cmp or compare or bool or null or count" > $O/pytest2.log 2>&1; tail -5 $O/pytest2.log
hipcc -O2 -std=c++17 -fPIC -shared --offload-arch=gfx950 -Wno-unused-value -o /tmp/libfake_rccl.so tests/cpp/fake_rccl.cpp
AH_COMM_WATCHDOG_S=150 AH_RCCL_LIBRARY=/tmp/libfake_rccl.so timeout 200 python tests/comm_ranks_worker.py 8 > $O/comm8.log 2>&1; tail -60 $O/comm8.log
python tools/size_sweep.py > $O/size_sweep.txt 2>&1; cat $O/size_sweep.txt
python tools/record_batch_latency.py > $O/rbl.txt 2>&1; tail -15 $O/rbl.txt
