#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_ccip.py -m gpu -x -q 2>&1 | tail -3 || exit 1
for f in 0 1; do if [ $f = 1 ]; then export HIPTS_CCIP_NO_LN_FUSION=1; else unset HIPTS_CCIP_NO_LN_FUSION; fi; echo "no_fusion=$f"; timeout -k 10 300 python tools/ccip_bench.py 2>&1 | tail -2; done
