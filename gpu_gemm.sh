#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for v in ${VARIANTS:-s3 pp v1}; do echo "== $v"; HIPTS_GEMM=$v timeout -k 10 300 python tools/gemm_bench.py "$@" || exit 1; done
