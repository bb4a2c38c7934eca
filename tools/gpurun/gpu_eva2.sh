#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for m in 8 5 8 5; do echo "minsub=$m"; HIPTS_EVA_MINSUB=$m timeout -k 10 600 python tools/eva_bench.py 2>&1 | tail -2 | head -1; done
