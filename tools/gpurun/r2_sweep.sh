#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for v in 0 1 2 3 4 5 6 7; do HIPTS_S1_VARIANT=$v timeout -k 10 120 python tools/s1_sweep.py 2>&1 | grep variant; done
