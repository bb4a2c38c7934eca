#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 300 python tests/diag/ccip_check.py tiny 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python tests/diag/ccip_check.py b36 2>&1 | grep -v amdgpu.ids
