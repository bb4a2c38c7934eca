#!/bin/bash
# round 5: CCIP encoder A/B of alternative prebuilt objects (csrc/ab_<name>_<tag>.o.keep replacing <name>.o): tools/ccip_bench.py base alt base alt.
# usage: r5_ccip_ab_objs.sh <tag> <name> [<name> ...]
set -o pipefail
cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
C=anime-illust-image-searcher_amd/csrc
mkdir -p /tmp/base; for n in "$@"; do cp $C/$n.o /tmp/base/$n.o; done; cp $C/../libhip_tagsearch.so /tmp/lib.keep
link() { (cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o) || exit 1; }
for v in base alt base alt; do
  for n in "$@"; do if [ $v = alt ]; then cp $C/ab_${n}_$TAG.o.keep $C/$n.o; else cp /tmp/base/$n.o $C/$n.o; fi; done
  link
  echo "== $v ($*)"; timeout -k 10 300 python tools/ccip_bench.py 2>&1 | grep -v amdgpu.ids | tail -2
done | tee gpurun_out/r5_ccip_abo_$TAG.txt
for n in "$@"; do cp /tmp/base/$n.o $C/$n.o; done; cp /tmp/lib.keep $C/../libhip_tagsearch.so
