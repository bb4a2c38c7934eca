#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_vit.py tests/test_gpu_eva.py tests/test_gpu_ccip.py -m gpu -q -x -s 2>&1 | grep -E "passed|failed|error|attention|forced|max \|" | tail -30 || exit 1
for c in 0 1 0 1; do HIPTS_ATTN_CLASSIC=$c timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-query --no-exclusive 2> gpurun_out/b.err | python -c "
import json,sys; d=json.load(sys.stdin); a=[k for k in d['kernels'] if k['kernel']=='attn_kernel'][0]; print('classic=$c', round(d['value'],1), 'img/s; attn avg', round(a['avg_us'],1), 'us', round(a['tflops'],1), 'TF; sustained', round(d['sustained']['images_per_s'],1))"; done
