#!/bin/bash
# round 4: GEMM / ViT / CCIP tests, then the bench line (ViT forward and the model sections) -- after a change of a GEMM epilogue
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x 2>&1 | tail -2 || exit 1
export HIPTS_BENCH_NO_SUSTAINED=1
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/epi.err | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline()); print('images/s %.0f  frac %.3f  ccip %.0f / %.0f  eva %.0f / %.0f' % (d['value'], d['model_mfma_frac'], d['ccip']['images_per_s_batch20'], d['ccip']['images_per_s_batch64'], d['eva02_large']['images_per_s_batch10'], d['eva02_large']['images_per_s_batch32']))" || exit 1; done
