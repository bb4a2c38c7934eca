#!/bin/bash
# round 3: wall-clock stamps inside the batched top-k kernel (measurement-only build of query.hip), then the default build again
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
cd anime-illust-image-searcher_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -DHIPTS_X_TOPK_STAMPS=${1:-0} -c query.hip -o query.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o || exit 1
(cd ../.. && timeout -k 10 200 python tools/topk_stamps.py 2>&1 | grep "k="; timeout -k 10 200 python tools/topk1_stamps.py 2>&1 | grep "topk<")
/opt/rocm/bin/hipcc $FLAGS -c query.hip -o query.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o
