#!/bin/bash
# round 4: device-only ceiling of the hybrid-decode pipeline (decode of batch k + 1 beside the forward of batch k, no worker processes)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 300 python tools/jpeg_overlap_bench.py 2>&1 | grep -v amdgpu.ids
