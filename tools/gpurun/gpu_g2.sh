#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_vit.py tests/test_gpu_gemm.py tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -4 || exit 1
for at in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-query 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [ (k['kernel'],round(k['avg_us'],1)) for k in d['roofline']['exclusive']['kernels'] if 'attn' in k['kernel']])"; done
