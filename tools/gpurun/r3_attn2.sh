#!/bin/bash
# round 3: the new attention kernel alone -- correctness of every variant, then timing against the round-2 kernel
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 500 python tools/attn2_check.py ${1:-both} ${2:-2,3} > gpurun_out/attn2.log 2>&1; echo "rc=$?"; tail -70 gpurun_out/attn2.log
