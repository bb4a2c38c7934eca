#!/bin/bash
# round 5: error of the half-operand attention against float64 by the reference exponent's head room (HIPTS_ATTN_REF_MARGIN), default kernel
R=$GRAFT_REPO_ROOT
cd $R/anime-illust-image-searcher_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function -fno-honor-nans -fno-slp-vectorize"
cp attn2.o /tmp/attn2.o.keep; cp ../libhip_tagsearch.so /tmp/lib.keep
for M in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DHIPTS_ATTN_REF_MARGIN=$M -c attn2.hip -o attn2.o 2> $R/gpurun_out/r5_attn_margin_build.log && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o || { tail -20 $R/gpurun_out/r5_attn_margin_build.log; exit 1; }
  echo "== margin $M"; (cd $R && timeout -k 10 200 python3 tools/attn2_check.py both 5 2>&1 | grep "f16=1\|CHECK")
done > $R/gpurun_out/r5_attn_margin.txt 2>&1
cp /tmp/attn2.o.keep attn2.o; cp /tmp/lib.keep ../libhip_tagsearch.so
cat $R/gpurun_out/r5_attn_margin.txt
