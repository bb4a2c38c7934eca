#!/bin/bash
# round 3: query-path tests
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_query.py tests/test_gpu_flows.py tests/test_abi.py -q -rf -s --durations=5 ${1:+-k "$1"} > gpurun_out/r3_query.log 2>&1
echo "pytest rc=$?"; grep -E "continuations|passed|failed|Error" gpurun_out/r3_query.log | cut -c1-200; tail -5 gpurun_out/r3_query.log | cut -c1-200
