#!/bin/bash
# round 4: where a tile's time goes in the persistent GEMM (in-kernel cycle stamps of workgroup 8, tools/gemm_bench.py with HIPTS_GEMM_STAMPS=1):
# the ViT's shapes per 32-image sub-batch
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
HIPTS_GEMM_STAMPS=1 timeout -k 10 300 python tools/gemm_bench.py gelu,25088,3072,768 qk,25088,2304,768 resid,25088,768,768 resid,25088,768,3072 2>&1 | grep -v amdgpu.ids | cut -c1-200
