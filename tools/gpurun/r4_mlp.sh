#!/bin/bash
# round 4: the fused MLP kernel (csrc/mlp.hip) -- unit test, time per launch at the encoder's shapes [, CCIP tests and images/s with and without]
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 300 python -m pytest tests/test_gpu_ccip.py -m gpu -q -x -s -k mlp_fused 2>&1 | grep -E "passed|failed|rror|assert" | tail -5 || exit 1
timeout -k 10 200 python tools/mlp_bench.py || exit 1
[ "$1" = "unit" ] && exit 0
timeout -k 10 600 python -m pytest tests/test_gpu_ccip.py -m gpu -q -x -s 2>&1 | grep -E "CCIP|passed|failed|rror" | tail -20 || exit 1
for m in 0 1; do echo "HIPTS_CCIP_FUSED_MLP=$m"; HIPTS_CCIP_FUSED_MLP=$m timeout -k 10 300 python tools/ccip_bench.py 2>&1 | tail -2 || exit 1; done
