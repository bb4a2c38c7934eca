#!/bin/bash
# round 4: EVA02-L batch 10 / 32 as one stream against two sub-batch streams, under the launcher's late-round-4 rules
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for s in 2 1 2 1; do echo "HIPTS_EVA_STREAMS=$s"; HIPTS_EVA_STREAMS=$s timeout -k 10 300 python tools/eva_bench.py 2>&1 | tail -2 || exit 1; done
