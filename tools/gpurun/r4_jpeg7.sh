#!/bin/bash
# round 4: where the decode pipeline's threads wait in steady state (HIPTS_PIPELINE_TIMING=1), Pillow workers and entropy-decode workers
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for w in ${1:-16}; do
  HIPTS_PIPELINE_TIMING=1 E2E_MODES=3,4 timeout -k 10 400 python tools/pipeline_e2e.py 8192 $w 2>&1 | grep -v amdgpu.ids
done
