#!/bin/bash
# round 4: head room of the attention fast path's reference exponent (csrc/variants/attn2_m{4,8}.o built beforehand with
# -DHIPTS_ATTN_REF_MARGIN=4 / 8): forward rate by image content, then the attention tests per variant; the default object is restored
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
cd anime-illust-image-searcher_amd/csrc
OTHERS=$(ls *.o | grep -v "^attn2.o$" | tr '\n' ' ')
for v in default m4 m8; do
  if [ $v = default ]; then G=attn2.o; else G=variants/attn2_$v.o; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so $OTHERS $G || exit 1
  echo "== $v"
  (cd ../.. && timeout -k 10 200 python tools/vit_content_bench.py 2>&1 | grep -E "uniform|halfflat|lineart|pipeline_e2e")
  (cd ../.. && timeout -k 10 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_vit.py -m gpu -q -x 2>&1 | tail -2)
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so $OTHERS attn2.o
