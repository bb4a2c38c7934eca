#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_query.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 300 python tools/single_query_bench.py 2>&1 | tail -1
timeout -k 10 300 python tools/query_bench.py 2>&1 | tail -1 | cut -c1-300
