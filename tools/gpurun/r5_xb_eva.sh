#!/bin/bash
# round 5: the blocked residual stream in the EVA02 forward: parity tests, then images/s A B A B
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_eva.py -x -q > gpurun_out/r5_xb_eva_tests.txt 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r5_xb_eva_tests.txt
tail -3 gpurun_out/r5_xb_eva_tests.txt
for v in 0 1 0 1; do
  echo "== HIPTS_X_BLOCKED=$v"; HIPTS_X_BLOCKED=$v timeout -k 10 300 python tools/eva_bench.py 2>&1 | grep -v amdgpu.ids | tail -3
done | tee gpurun_out/r5_xb_eva_ab.txt
