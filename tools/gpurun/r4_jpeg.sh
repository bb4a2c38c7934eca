#!/bin/bash
# round 4: hybrid JPEG decode -- parity tests, then the end-to-end tagging rate by input pipeline (tools/pipeline_e2e.py)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_jpeg.py tests/test_oracle_jpeg.py -q -rf -x > gpurun_out/r4_jpeg_tests.log 2>&1; rc=$?
echo "jpeg pytest rc=$rc"; tail -15 gpurun_out/r4_jpeg_tests.log | cut -c1-400
[ $rc -ne 0 ] && exit 1
mkdir -p gpurun_out/r04
timeout -k 10 600 python tools/pipeline_e2e.py ${1:-10240} 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/pipeline_e2e.txt
