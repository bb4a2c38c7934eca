#!/bin/bash
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for d in 0 2 6 14 10 0; do echo "== dbg=$d"; HIPTS_GEMM_DBG=$d timeout -k 10 300 python tools/gemm_bench.py gelu,4096,4096,4096 || exit 1; done
