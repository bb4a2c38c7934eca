#!/bin/bash
# round 5: CCIP wide-stage LayerNorm fold (HIPTS_CCIP_LN_FOLD, default 1): tests, then the encoder's rate A B A B
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_ccip.py -x -q -s > gpurun_out/r5_ccip_fold_tests.txt 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r5_ccip_fold_tests.txt
tail -15 gpurun_out/r5_ccip_fold_tests.txt
for v in 0 1 0 1; do
  echo "== HIPTS_CCIP_LN_FOLD=$v"; HIPTS_CCIP_LN_FOLD=$v timeout -k 10 300 python tools/ccip_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
done | tee gpurun_out/r5_ccip_fold_ab.txt
