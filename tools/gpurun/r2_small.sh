#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 400 python -m pytest tests/test_gpu_query.py tests/test_gpu_configs.py -m gpu -q -x -k "query or search or topk or config2 or bm25" 2>&1 | tail -3 || exit 1
timeout -k 10 500 python tools/s1_small.py 2>&1 | grep -v Warning | tee gpurun_out/s1_small.txt
