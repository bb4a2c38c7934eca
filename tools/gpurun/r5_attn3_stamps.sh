#!/bin/bash
# round 5: cycle stamps of one wave of the one-wave-per-SIMD attention (csrc/attn3.h). usage: r5_attn3_stamps.sh <workgroup> [extra -D flags]
R=$GRAFT_REPO_ROOT
WG=${1:-0}; shift
cd $R/anime-illust-image-searcher_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function -fno-honor-nans -fno-slp-vectorize"
cp attn2.o /tmp/attn2.o.keep; cp ../libhip_tagsearch.so /tmp/lib.keep
/opt/rocm/bin/hipcc $FLAGS -DHIPTS_A3_STAMPS=$WG "$@" -c attn2.hip -o attn2.o 2> $R/gpurun_out/r5_attn3_build.log && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o || { tail -20 $R/gpurun_out/r5_attn3_build.log; exit 1; }
(cd $R && timeout -k 10 200 python3 tools/attn2_check.py stamps3 > gpurun_out/r5_attn3_stamps.txt 2>&1; echo "rc $?" >> gpurun_out/r5_attn3_stamps.txt)
cp /tmp/attn2.o.keep attn2.o; cp /tmp/lib.keep ../libhip_tagsearch.so
cat $R/gpurun_out/r5_attn3_stamps.txt
