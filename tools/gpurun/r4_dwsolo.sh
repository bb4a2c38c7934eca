#!/bin/bash
# round 4: what does a 4-wave workgroup (one wave per SIMD) achieve ALONE on a CU?  The two-per-CU GEMM (HIPTS_GEMM=dw) with its in-loop
# stamps, two workgroups per CU and one (HIPTS_DW_SOLO=1), on the fc1 and q|k|v shapes
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for solo in 0 1; do
  echo "== HIPTS_DW_SOLO=$solo"
  HIPTS_GEMM=dw HIPTS_DW_SOLO=$solo HIPTS_GEMM_STAMPS=1 timeout -k 10 200 python tools/gemm_bench.py gelu,25088,3072,768 qk,25088,2304,768 resid,25088,768,3072 2>&1 | grep -v amdgpu.ids | cut -c1-200
done
echo "== pp"
timeout -k 10 200 python tools/gemm_bench.py gelu,25088,3072,768 qk,25088,2304,768 resid,25088,768,3072 2>&1 | grep -v amdgpu.ids
