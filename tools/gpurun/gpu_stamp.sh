#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for sh in resid,50176,768,768 resid,50176,768,3072 gelu,50176,3072,768 qk,50176,1536,768; do
echo "=== $sh"; HIPTS_GEMM_STAMPS=1 timeout -k 10 120 python tools/gemm_bench.py $sh 2>&1 | grep -v amdgpu.ids | head -12
done
