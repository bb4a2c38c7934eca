#!/bin/bash
# round 4: the tagging loop with predict() split in two pipelined halves -- CLI tests, then the end-to-end rate again
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_flows.py tests/test_gpu_jpeg.py tests/test_pipeline.py -m gpu -q -rf -x > gpurun_out/r4_jpeg4_tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -6 gpurun_out/r4_jpeg4_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
E2E_MODES=${1:-3,4} timeout -k 10 400 python tools/pipeline_e2e.py 10240 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/pipeline_e2e_b.txt
