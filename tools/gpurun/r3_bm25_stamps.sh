#!/bin/bash
# round 3: wall-clock stamps inside bm25_postings_kernel for a few workgroups (measurement-only builds of query.hip), then the default build again
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
cd anime-illust-image-searcher_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function"
for wg in "$@"; do
/opt/rocm/bin/hipcc $FLAGS -DHIPTS_X_TOPK_STAMPS=$wg -c query.hip -o query.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o || exit 1
(cd ../.. && WG=$wg timeout -k 10 200 python tools/bm25_stamps.py 2>&1 | grep "query ")
done
/opt/rocm/bin/hipcc $FLAGS -c query.hip -o query.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o
