#!/bin/bash
# round 4: head room of the attention fast path's reference exponent -- csrc/variants/attn2_m<N>.o built beforehand with
# -DHIPTS_ATTN_REF_MARGIN=N; per variant: forward rate by picture content (ViT-B/16, EVA02-L), the parity tests, and the bench's oracle
# check of the trained-like checkpoint on structured images.  usage: r4_margin.sh "default m10 m12"
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
mkdir -p gpurun_out/r04
cd anime-illust-image-searcher_amd/csrc
OTHERS=$(ls *.o | grep -v "^attn2.o$" | tr '\n' ' ')
for v in ${1:-default}; do
  if [ $v = default ]; then G=attn2.o; else G=variants/attn2_$v.o; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so $OTHERS $G || exit 1
  echo "== $v"
  (cd ../.. && timeout -k 10 200 python tools/vit_content_bench.py 2>&1 | grep -E "uniform|pipeline_e2e|different")
  (cd ../.. && MODEL=eva timeout -k 10 300 python tools/vit_content_bench.py 2>&1 | grep -E "uniform|pipeline_e2e|different")
  (cd ../.. && timeout -k 10 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_vit.py tests/test_gpu_eva.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5)
  (cd ../.. && HIPTS_BENCH_NO_SUSTAINED=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-query --no-exclusive > gpurun_out/r04/margin_$v.json 2> gpurun_out/margin.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r04/margin_$v.json').read().strip().splitlines()[-1]); o=d['output_check']['oracle']; print('oracle check max abs logit error per image:', ['%.2e' % v for v in o['max_abs_logit_error']]); print('rms relative:', ['%.2e' % v for v in o['rms_relative_logit_error']])")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so $OTHERS attn2.o
