#!/bin/bash
# round 4: decode + resize kernels batched over the images of a call -- parity tests, the device-only ceiling, the end-to-end table rows 3-4
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_resize.py tests/test_pipeline.py tests/test_gpu_e2e.py tests/test_gpu_configs.py -m gpu -q -x > gpurun_out/r4_jpeg9_tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/r4_jpeg9_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python tools/jpeg_bench.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/jpeg_overlap_bench.py 2>&1 | grep -v amdgpu.ids
E2E_MODES=3,4 timeout -k 10 400 python tools/pipeline_e2e.py 10240 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/pipeline_e2e_d.txt
