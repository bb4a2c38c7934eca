#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
HIPTS_GEMM=pp2 timeout -k 10 600 python -m pytest tests/test_gpu_vit.py -m gpu -x -q 2>&1 | tail -4 || exit 1
for v in ${VARIANTS:-pp2 pp pp2 pp}; do echo "== $v"; HIPTS_GEMM=$v timeout -k 10 300 python tools/gemm_bench.py "$@" || exit 1; done
