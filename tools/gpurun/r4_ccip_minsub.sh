#!/bin/bash
# round 4: CCIP at the reference's batch of 20 as one stream (default: sub-batches of >= 16 images) against two streams of 10
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for m in 16 10 16 10; do echo "HIPTS_CCIP_MINSUB=$m"; HIPTS_CCIP_MINSUB=$m timeout -k 10 300 python tools/ccip_bench.py 2>&1 | tail -2 || exit 1; done
