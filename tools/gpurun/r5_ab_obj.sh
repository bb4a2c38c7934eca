#!/bin/bash
# round 5: A/B of one prebuilt object through the bench line's forward (A B A B on one box).
# usage: r5_ab_obj.sh <name>.o <alternative object file> <tag>     (both under csrc/; the default object is put back at the end)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OBJ=$1; ALT=$2; TAG=$3
C=anime-illust-image-searcher_amd/csrc
cp $C/$OBJ /tmp/base.o; cp $C/../libhip_tagsearch.so /tmp/lib.keep
link() { (cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so $(ls *.o | grep -v "^ab_") ) || exit 1; }
for v in base alt base alt; do
  if [ $v = alt ]; then cp $C/$ALT $C/$OBJ; else cp /tmp/base.o $C/$OBJ; fi
  link
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-query --no-cpu-baseline --no-exclusive 2> gpurun_out/r5_abo_$TAG.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', round(d['value'],1), 'img/s  sustained', round(d['sustained'].get('images_per_s',0),1), [ (k['kernel'][:24], round(k['avg_us'],1)) for k in d['kernels'][:8]])" | tee -a gpurun_out/r5_abo_$TAG.txt
done
cp /tmp/base.o $C/$OBJ; cp /tmp/lib.keep $C/../libhip_tagsearch.so
