#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_resize.py tests/test_abi.py -q -rf --durations=5 > gpurun_out/r3_resize.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/r3_resize.log | cut -c1-250
