#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
SH="resid,50176,768,768 resid,25088,768,768 resid,50176,768,3072 resid,25088,768,3072 vt,25088,768,768 qk,25088,1536,768 gelu,25088,3072,768"
for v in pp dw; do echo "--- $v"; HIPTS_GEMM=$v timeout -k 10 300 python tools/gemm_bench.py $SH 2>&1 | grep -v amdgpu.ids; done
