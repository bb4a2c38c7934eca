#!/bin/bash
# round 5: the fp32 residual stream as 16 x 16 blocks (HIPTS_X_BLOCKED, default 1): parity tests, the residual launches alone, the forward A B A B
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_configs.py -x -q -k "vit or config1 or config0" > gpurun_out/r5_xb_tests.txt 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r5_xb_tests.txt
tail -4 gpurun_out/r5_xb_tests.txt
export HIPTS_DBG_GEMM_F16=1 HIPTS_DBG_GEMM_SHARED=1
( for v in 0 1 0 1; do
    echo "== blocked=$v"
    if [ $v = 1 ]; then export HIPTS_DBG_X_BLOCKED=1; else unset HIPTS_DBG_X_BLOCKED; fi
    timeout -k 10 200 python tools/gemm_bench.py xg,25088,768,768 xg,25088,768,3072 xg,50176,768,768 2>&1 | grep -v amdgpu.ids
  done
  for v in 0 1; do
    echo "== stamps blocked=$v"
    if [ $v = 1 ]; then export HIPTS_DBG_X_BLOCKED=1; else unset HIPTS_DBG_X_BLOCKED; fi
    HIPTS_GEMM_STAMPS=1 timeout -k 10 200 python tools/gemm_bench.py xg,25088,768,768 2>&1 | grep -v amdgpu.ids
  done ) > gpurun_out/r5_xblocked.txt 2>&1
cat gpurun_out/r5_xblocked.txt
unset HIPTS_DBG_GEMM_F16 HIPTS_DBG_GEMM_SHARED HIPTS_DBG_X_BLOCKED
bash tools/gpurun/r5_ab_env.sh HIPTS_X_BLOCKED 0 1 xblocked
