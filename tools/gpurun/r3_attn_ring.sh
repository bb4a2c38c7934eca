#!/bin/bash
# round 3: sequential-halves attention body with a ring of three K / V slots (two tiles in flight) against the two-slot ring
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
cd anime-illust-image-searcher_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function -fno-honor-nans"
for M in "$@"; do
  D=""; [ "$M" != "none" ] && for m in ${M//+/ }; do D="$D -D$m"; done
  /opt/rocm/bin/hipcc $FLAGS $D -c attn2.hip -o attn2.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o || exit 1
  echo "== $M"; (cd ../.. && timeout -k 10 300 python tools/attn2_check.py both 5,4 2>&1 | grep -E "FAIL|CHECK|attn2 variant|attn \(round")
done
/opt/rocm/bin/hipcc $FLAGS -c attn2.hip -o attn2.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o
