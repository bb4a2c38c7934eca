#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_eva.py -m gpu -x -q 2>&1 | tail -2
for f in 0 1 0 1; do
echo "=== HIPTS_LN_FOLD=$f"; HIPTS_LN_FOLD=$f timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query --no-exclusive 2>gpurun_out/bench_fold$f.err >gpurun_out/bench_fold$f.json; grep -E "RESID" gpurun_out/bench_fold$f.err
python -c "import json,sys; d=json.loads(open('gpurun_out/bench_fold$f.json').read()); print('img/s', d['value'], 'ms/step', d['ms_per_step'], 'mfma frac', d['model_mfma_frac'])"
done
timeout -k 10 300 python tools/eva_bench.py 2>&1 | tail -2
