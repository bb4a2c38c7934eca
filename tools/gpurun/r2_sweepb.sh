#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for s in 2 1; do
  HIPTS_VIT_STREAMS=$s timeout -k 10 200 python tools/vit_batch_sweep.py 64 55 110 128 2>&1 | grep -v Warning | tee -a gpurun_out/sweepb.txt || exit 1
done
