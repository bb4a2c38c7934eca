#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
HIPTS_DBG_DELAY=-1 HIPTS_GEMM_STAMPS=1 HIPTS_GEMM_TRACE=gpurun_out/dw_trace_qk.txt HIPTS_GEMM=dw timeout -k 10 300 python tools/gemm_bench.py qk,50176,1536,768 2>&1 | tail -3
python tools/dw_trace.py gpurun_out/dw_trace_qk.txt
