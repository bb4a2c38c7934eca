#!/bin/bash
# round 5: A/B of attention builds. usage: r5_attn_defs.sh <variants> <def-set> [<def-set> ...]   (def-set: "none" or -DX=1+-DY=2)
R=$GRAFT_REPO_ROOT
V=$1; shift
cd $R/anime-illust-image-searcher_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function -fno-honor-nans -fno-slp-vectorize"
cp attn2.o /tmp/attn2.o.keep; cp ../libhip_tagsearch.so /tmp/lib.keep
for M in "$@"; do
  D=""; [ "$M" != "none" ] && D="${M//+/ }"
  /opt/rocm/bin/hipcc $FLAGS $D -c attn2.hip -o attn2.o 2> $R/gpurun_out/r5_attn_defs_build.log && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o || { tail -20 $R/gpurun_out/r5_attn_defs_build.log; exit 1; }
  echo "== $M"; (cd $R && timeout -k 10 200 python3 tools/attn2_check.py both $V 2>&1 | grep "attn2 variant\|CHECK\|FAIL")
done > $R/gpurun_out/r5_attn_defs.txt 2>&1
cp /tmp/attn2.o.keep attn2.o; cp /tmp/lib.keep ../libhip_tagsearch.so
cat $R/gpurun_out/r5_attn_defs.txt
