#!/bin/bash
# round 4: where the decode pipeline's threads wait (HIPTS_PIPELINE_TIMING=1), Pillow workers and entropy-decode workers, 16 and 12 workers
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for w in 16; do
  timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_flows.py tests/test_gpu_jpeg.py tests/test_pipeline.py tests/test_gpu_e2e.py -m gpu -q -x 2>&1 | tail -3
  HIPTS_PIPELINE_TIMING=1 E2E_MODES=3,4 timeout -k 10 400 python tools/pipeline_e2e.py 8192 $w 2>&1 | grep -v amdgpu.ids
done
