#!/bin/bash
# round 5: the forward by picture content (tools/vit_content_bench.py) for attention builds. usage: r5_attn_content.sh <def-set> [<def-set> ...]
R=$GRAFT_REPO_ROOT
cd $R/anime-illust-image-searcher_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function -fno-honor-nans -fno-slp-vectorize"
cp attn2.o /tmp/attn2.o.keep; cp ../libhip_tagsearch.so /tmp/lib.keep
for M in "$@"; do
  D=""; [ "$M" != "none" ] && D="${M//+/ }"
  /opt/rocm/bin/hipcc $FLAGS $D -c attn2.hip -o attn2.o 2> $R/gpurun_out/r5_attn_content_build.log && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o || { tail -20 $R/gpurun_out/r5_attn_content_build.log; exit 1; }
  echo "== $M"; (cd $R && timeout -k 10 300 python3 tools/vit_content_bench.py 2>&1 | grep "images/s")
done > $R/gpurun_out/r5_attn_content.txt 2>&1
cp /tmp/attn2.o.keep attn2.o; cp /tmp/lib.keep ../libhip_tagsearch.so
cat $R/gpurun_out/r5_attn_content.txt
