#!/bin/bash
# round 4: batched query path -- BM25 on a side stream beside the index product (HIPTS_SEARCH_OVERLAP=0/1), the direct oracle test of the
# benched shape, the query tests
mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_query.py tests/test_gpu_configs.py -m gpu -q -rf -x > gpurun_out/r4_query_tests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r4_query_tests.log | cut -c1-200
for ov in 0 1 1 0; do
  echo "== HIPTS_SEARCH_OVERLAP=$ov"
  HIPTS_SEARCH_OVERLAP=$ov timeout -k 10 300 python tools/query_bench.py 2>&1 | tail -2 | cut -c1-1500
done
