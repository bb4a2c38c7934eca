#!/bin/bash
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
run() { env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-query 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run HIPTS_VIT_STREAMS=2
run HIPTS_VIT_STREAMS=2 HIPTS_GEMM_BM=256
run HIPTS_VIT_STREAMS=3
run HIPTS_VIT_STREAMS=3 HIPTS_GEMM_BM=256
run HIPTS_VIT_STREAMS=1
done
