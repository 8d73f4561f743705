#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_filter_small.py tests/test_gpu_cdata.py -q -m gpu -k "arith or scalar or golden or bitwise or small or bool or null or deferred or neg" -p no:cacheprovider > $O/pytest.log 2>&1; tail -8 $O/pytest.log | grep -v "^\.\.\.\."
python tools/size_sweep.py 1e4 1e5 1e6 > $O/size_sweep.txt 2>&1; cat $O/size_sweep.txt
python tools/record_batch_latency.py > $O/rbl.txt 2>&1; tail -12 $O/rbl.txt
