#!/bin/bash
# round 5: MFMA issue order inside a phase of gemm_pp_kernel (prebuilt objects csrc/ab_gemm_order{1,2}.o.keep): standalone GEMMs on random data, then the forward
set -o pipefail
cd "$GRAFT_REPO_ROOT"
C=anime-illust-image-searcher_amd/csrc
cp $C/gemm.o /tmp/base.o; cp $C/../libhip_tagsearch.so /tmp/lib.keep
link() { (cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o) || exit 1; }
export HIPTS_DBG_GEMM_F16=1
for v in 0 1 2 0 1 2; do
  if [ $v = 0 ]; then cp /tmp/base.o $C/gemm.o; else cp $C/ab_gemm_order$v.o.keep $C/gemm.o; fi
  link
  echo "== order $v"; timeout -k 10 120 python tools/gemm_bench.py gelu,8192,8192,8192 gelu,4096,4096,4096 gelu,25088,3072,768 2>&1 | grep -v amdgpu.ids | tail -3
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-query --no-cpu-baseline --no-exclusive 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('forward', round(d['value'],1), 'img/s  sustained', round(d['sustained'].get('images_per_s',0),1))"
done 2>&1 | tee gpurun_out/r5_gemm_order.txt
cp /tmp/base.o $C/gemm.o; cp /tmp/lib.keep $C/../libhip_tagsearch.so
