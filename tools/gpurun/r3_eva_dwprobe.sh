#!/bin/bash
# timing probe: EVA02-L batch 10 with the residual GEMMs (proj, fc2) on the two-workgroups-per-CU kernel (statistics wrong in this build)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
cd anime-illust-image-searcher_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -DHIPTS_X_DW_STAT -c gemm.hip -o gemm.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o || exit 1
cd ../..
for m in 0 0x2000 0x4000 0x6000; do
  echo "dw mask $m"; HIPTS_GEMM_DW_MASK=$m timeout -k 10 300 python tools/eva_bench.py 2>&1 | grep batch
done
cd anime-illust-image-searcher_amd/csrc
/opt/rocm/bin/hipcc $FLAGS -c gemm.hip -o gemm.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o
