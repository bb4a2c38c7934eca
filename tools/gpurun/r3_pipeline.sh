#!/bin/bash
# round 3: decode-only worker processes + device pad / resize (pipeline.DecodePool(device_resize=True)): parity tests, then end-to-end images/s
mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_resize.py tests/test_pipeline.py tests/test_gpu_e2e.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -5 || exit 1
timeout -k 10 900 python tools/pipeline_e2e.py ${1:-4096} ${2:-16} 2>&1 | grep -v Warning | tee gpurun_out/r03/pipeline_e2e.txt
