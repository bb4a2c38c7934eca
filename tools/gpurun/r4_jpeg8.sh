#!/bin/bash
# round 4: the full end-to-end table after the three-part ring and the asynchronous consumer, then the progressive corpus
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
mkdir -p gpurun_out/r04
timeout -k 10 600 python tools/pipeline_e2e.py 10240 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/pipeline_e2e_c.txt
E2E_PROGRESSIVE=1 E2E_MODES=3,4 timeout -k 10 400 python tools/pipeline_e2e.py 8192 16 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/pipeline_e2e_c.txt
E2E_MODES=3,4 timeout -k 10 400 python tools/pipeline_e2e.py 8192 8 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/pipeline_e2e_c.txt
