#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for dl in 0 1000 2000 3000; do echo "delay=$dl"; HIPTS_DBG_DELAY=$dl timeout -k 10 300 python tools/gemm_bench.py resid,50176,768,768 resid,50176,768,3072 resid,100352,768,768 || exit 1; done
