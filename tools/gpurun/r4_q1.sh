#!/bin/bash
# round 4: one-query path with combine + collect in one launch (search1_finish_kernel, threshold from the score kernel's witness records):
# query tests, then one-query timing with HIPTS_SEARCH1_FINISH=0 / 1 (one process each)
mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_query.py tests/test_gpu_configs.py tests/test_gpu_flows.py -m gpu -q -rf -x > gpurun_out/r4_q1_tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/r4_q1_tests.log | cut -c1-250
[ $rc -ne 0 ] && exit 1
for f in 0 1 0 1; do
  echo "== HIPTS_SEARCH1_FINISH=$f"
  HIPTS_SEARCH1_FINISH=$f timeout -k 10 300 python tools/single_query_bench.py 2>&1 | tail -6 | cut -c1-300
done
