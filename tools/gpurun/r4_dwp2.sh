#!/bin/bash
# round 4: persistent two-per-CU GEMM without its epilogue (HIPTS_GEMM_DWP_SLEEP = -100 - n: no epilogue, n sleeps for the second half of the grid)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for sl in -100 -102 -2; do
  echo "== dwp sleep $sl"
  HIPTS_GEMM_DWP_MASK=18 HIPTS_GEMM_DWP_SLEEP=$sl timeout -k 10 200 python tools/gemm_bench.py gelu,25088,3072,768 qk,25088,2304,768 2>&1 | grep -v amdgpu.ids
done
