#!/bin/bash
# round 4: fused MLP at C = 128, four-wave workgroups: four chunk buffers (two workgroups per CU) against two (three per CU: the default; HIPTS_MLP_NBUF128=4 = four)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 300 python -m pytest tests/test_gpu_ccip.py -m gpu -q -x 2>&1 | tail -2 || exit 1
for n in 4 2; do echo "HIPTS_MLP_NBUF128=$n"; HIPTS_MLP_NBUF128=$n timeout -k 10 200 python tools/mlp_bench.py 2>&1 | grep "C 128" || exit 1; done
for n in 4 2 4 2; do echo "HIPTS_MLP_NBUF128=$n"; HIPTS_MLP_NBUF128=$n timeout -k 10 300 python tools/ccip_bench.py 2>&1 | tail -2 || exit 1; done
