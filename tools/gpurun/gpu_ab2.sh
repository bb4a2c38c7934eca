#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for rep in 1 2; do for t in 1 0; do
echo "=== tail=$t rep=$rep"; HIPTS_GEMM_TAIL=$t timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query 2>gpurun_out/b.err >gpurun_out/b.json; grep -E "gemm|attn" gpurun_out/b.err | cut -c1-80
python -c "import json,sys; d=json.loads(open('gpurun_out/b.json').read()); print('img/s', d['value'], 'ms/step', d['ms_per_step'])"
done; done
