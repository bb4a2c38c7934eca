#!/bin/bash
# round 4: phase stamps of one wave of the fused MLP kernel (stamped build of mlp.hip; the object is removed afterwards)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
cd anime-illust-image-searcher_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -DHIPTS_MLP_STAMPS=${1:-100} $2 -c mlp.hip -o mlp.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o || exit 1
cd ../..
timeout -k 10 200 python tools/mlp_stamps.py; rc=$?
rm -f anime-illust-image-searcher_amd/csrc/mlp.o
exit $rc
