#!/bin/bash
# round 4: CCIP tests, then images/s with and without the fused MLP and the matrix-core depthwise 7x7
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_ccip.py -m gpu -q -x -s 2>&1 | grep -E "CCIP|passed|failed|rror" | tail -12 || exit 1
for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg; echo "HIPTS_CCIP_FUSED_MLP=$1 HIPTS_CCIP_DW_MFMA=$2"; HIPTS_CCIP_FUSED_MLP=$1 HIPTS_CCIP_DW_MFMA=$2 timeout -k 10 300 python tools/ccip_bench.py 2>&1 | tail -2 || exit 1; done
