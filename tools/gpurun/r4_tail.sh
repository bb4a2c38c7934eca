#!/bin/bash
# round 4: the ViT forward's tail -- coalesced patchify, pool kernel with 28 splits + prefetch (HIPTS_POOL_SPLITS=8: the old split count),
# split-K of the tag head (HIPTS_GEMM_SPLITK_HEAD=4) -- tests, then images/s per setting; then the query section with two batches in flight
mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_gemm.py -m gpu -q -rf -x > gpurun_out/r4_tail_tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -6 gpurun_out/r4_tail_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
export HIPTS_BENCH_NO_SUSTAINED=1
for cfgs in "28 0" "8 0" "28 4" "28 0" "8 0" "28 4"; do
  set -- $cfgs
  HIPTS_POOL_SPLITS=$1 HIPTS_GEMM_SPLITK_HEAD=$2 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-query --no-exclusive > gpurun_out/r04/tail.json 2> gpurun_out/tail.err || { tail -5 gpurun_out/tail.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04/tail.json').read().strip().splitlines()[-1]); print('pool splits $1 head splitk $2: images/s', round(d['value'],1))"
done
timeout -k 10 300 python tools/query_bench.py 2>&1 | tail -1 | cut -c1-400
