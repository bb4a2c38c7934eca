#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_vit.py -m gpu -x -q 2>&1 | tail -8 || exit 1
for v in ${VARIANTS:-pp}; do
echo "=== $v"; HIPTS_GEMM=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query 2>gpurun_out/bench_$v.err >gpurun_out/bench_$v.json; grep -E "kernel|gemm|attn|layernorm" gpurun_out/bench_$v.err
python -c "import json,sys; d=json.loads(open('gpurun_out/bench_$v.json').read()); print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'mfma frac', d['model_mfma_frac'])"
done
