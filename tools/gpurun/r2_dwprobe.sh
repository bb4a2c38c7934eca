#!/bin/bash
# pp (default) vs dw (two 256x128 workgroups per CU) on the ViT's residual shapes, plain residual epilogue, 32- and 64-image row counts
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for v in pp dw pp dw; do echo "== HIPTS_GEMM=$v"; HIPTS_GEMM=$v timeout -k 10 120 python tools/gemm_bench.py resid,25088,768,768 resid,50176,768,768 resid,25088,768,3072 resid,50176,768,3072 vt,50176,768,768 2>&1 | grep -v amdgpu.ids; done
