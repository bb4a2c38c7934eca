#!/bin/bash
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_vit.py -m gpu -x -q 2>&1 | tail -4 || exit 1
for rep in 1 2; do for bm in 224 256; do echo "== BM=$bm"; HIPTS_GEMM_BM=$bm timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query 2>gpurun_out/b.err >gpurun_out/b.json; grep -E "gemm|patchify" gpurun_out/b.err | cut -c1-75; python -c "import json; d=json.loads(open('gpurun_out/b.json').read()); print('img/s', d['value'], 'ms/step', d['ms_per_step'])"; done; done
