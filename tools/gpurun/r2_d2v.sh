#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_d2v_train.py tests/test_gpu_e2e.py tests/test_gpu_d2v_tags.py -m gpu -q -x -s 2>&1 | grep -vE "^$|Warning|diff_arr|^  " | tail -25
