#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_d2v_tags.py tests/test_gpu_d2v_train.py tests/test_gpu_query.py -m gpu -q -x 2>&1 | tail -4 || exit 1
for p in 0 1; do HIPTS_D2V_PLAN=$p timeout -k 10 300 python tools/d2v_latency.py 2>&1 | grep plan=; done
