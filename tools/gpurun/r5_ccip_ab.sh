#!/bin/bash
# round 5: CCIP encoder A/B of an environment switch: the CCIP tests (default setting), then tools/ccip_bench.py A B A B.  usage: r5_ccip_ab.sh VAR A B
set -o pipefail
cd "$GRAFT_REPO_ROOT"
VAR=$1; A=$2; B=$3
timeout -k 10 900 python -m pytest tests/test_gpu_ccip.py tests/test_gpu_e2e.py -x -q > gpurun_out/r5_ccip_${VAR}_tests.txt 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r5_ccip_${VAR}_tests.txt
tail -3 gpurun_out/r5_ccip_${VAR}_tests.txt
for v in $A $B $A $B; do
  echo "== $VAR=$v"; env $VAR=$v timeout -k 10 300 python tools/ccip_bench.py 2>&1 | grep -v amdgpu.ids | tail -2
done | tee gpurun_out/r5_ccip_${VAR}_ab.txt
