#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 400 python -m pytest tests/test_gpu_query.py tests/test_gpu_configs.py -m gpu -q -x -k "query or search or topk or config2 or bm25" 2>&1 | tail -3 || exit 1
for f in 1 0; do echo "fused $f"; HIPTS_SEARCH1_FUSED=$f timeout -k 10 120 python tools/s1_sweep.py 2>&1 | grep variant || exit 1; done
HIPTS_SEARCH1_FUSED=1 S1_DIM=768 timeout -k 10 120 python tools/s1_sweep.py 2>&1 | grep variant
timeout -k 10 200 python tools/single_query_bench.py 2>&1 | grep -E "python path|C ABI|k=10"
