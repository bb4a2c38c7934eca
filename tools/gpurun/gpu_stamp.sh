#!/bin/bash
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
HIPTS_GEMM=pp2 HIPTS_GEMM_STAMPS=1 timeout -k 10 300 python tools/gemm_bench.py gelu,4096,4096,4096 2>&1
