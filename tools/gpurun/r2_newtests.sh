#!/bin/bash
# round 2: the new GPU tests (no -x: all failures in one call), then the one-query timing
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 1000 python -m pytest tests/test_gpu_query.py tests/test_gpu_configs.py tests/test_gpu_flows.py tests/test_gpu_multirank.py -m gpu -q -rf --durations=15 > gpurun_out/r2_newtests.log 2>&1
echo "pytest rc=$?"; tail -60 gpurun_out/r2_newtests.log
timeout -k 10 200 python tools/single_query_bench.py 2>&1 | tail -3
