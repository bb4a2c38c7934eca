#!/bin/bash
# round 3: in-kernel cycle stamps of one wave of the attention kernel (HIPTS_X_STAMPS=<workgroup id> build of csrc/attn2.hip)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
V=${1:-5}; shift
cd anime-illust-image-searcher_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function -fno-honor-nans"
for M in "${@:-HIPTS_X_STAMPS=1500}"; do
  D=""; for m in ${M//+/ }; do D="$D -D$m"; done
  /opt/rocm/bin/hipcc $FLAGS $D -c attn2.hip -o attn2.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o || exit 1
  echo "== $M"; (cd ../.. && timeout -k 10 120 python tools/attn2_check.py stamps $V 2>&1 | tail -20)
done
/opt/rocm/bin/hipcc $FLAGS -c attn2.hip -o attn2.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o
