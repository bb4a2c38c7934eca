#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 300 python -m pytest tests/test_gpu_d2v_train.py tests/test_gpu_query.py -m gpu -v -x --durations=5 2>&1 | grep -v Warning | tee gpurun_out/d2v_pytest.log | tail -40
for p in 0 1; do HIPTS_D2V_PLAN=$p timeout -k 10 200 python tools/d2v_latency.py 2>&1 | grep plan= | tee -a gpurun_out/d2v_pytest.log; done
