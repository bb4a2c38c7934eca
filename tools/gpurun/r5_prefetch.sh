#!/bin/bash
# round 5: HIPTS_EPI_PREFETCH -- the residual epilogue's fp32 tile requested into L2 during the last K-tile: standalone launches with stamps, then the forward A B A B
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export HIPTS_DBG_GEMM_F16=1 HIPTS_DBG_GEMM_SHARED=1
for v in 0 1 0 1; do
  echo "== HIPTS_EPI_PREFETCH=$v"; HIPTS_EPI_PREFETCH=$v timeout -k 10 200 python tools/gemm_bench.py xg,25088,768,768 xg,25088,768,3072 xg,50176,768,768 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r5_prefetch.txt 2>&1
for v in 0 1; do
  echo "== stamps HIPTS_EPI_PREFETCH=$v"; HIPTS_EPI_PREFETCH=$v HIPTS_GEMM_STAMPS=1 timeout -k 10 200 python tools/gemm_bench.py xg,25088,768,768 2>&1 | grep -v amdgpu.ids
done >> gpurun_out/r5_prefetch.txt 2>&1
cat gpurun_out/r5_prefetch.txt
unset HIPTS_DBG_GEMM_F16 HIPTS_DBG_GEMM_SHARED
bash tools/gpurun/r5_ab_env.sh HIPTS_EPI_PREFETCH 0 1 prefetch
