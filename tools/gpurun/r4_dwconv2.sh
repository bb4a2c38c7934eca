#!/bin/bash
# round 4: matrix-core depthwise 7x7 -- is it bound by memory?  batch 8 (inputs + outputs inside the Infinity Cache) vs 32; tiles per workgroup
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for b in 8 32; do echo "batch $b"; B=$b timeout -k 10 200 python tools/dwconv_bench.py || exit 1; done
for t in 1 2 12; do echo "tiles per workgroup $t"; HIPTS_CCIP_DW_TPW=$t timeout -k 10 200 python tools/dwconv_bench.py || exit 1; done
