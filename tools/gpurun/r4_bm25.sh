#!/bin/bash
# round 4: BM25 postings walk cut into document slices (HIPTS_BM25_PARTS workgroups per query; 1 = one workgroup per query as before):
# query tests under the default, then the batched rate per setting
mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_query.py tests/test_gpu_configs.py tests/test_gpu_flows.py -m gpu -q -rf -x > gpurun_out/r4_bm25_tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r4_bm25_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
for p in 1 4 2 8 1 4; do
  echo "== HIPTS_BM25_PARTS=$p"
  HIPTS_BM25_PARTS=$p timeout -k 10 300 python tools/query_bench.py 2>&1 | tail -1 | sed -E "s/.*'batched_qps': ([0-9.]+), 'batched_one_at_a_time_qps': ([0-9.]+).*bm25_postings_kernel', 'launches': [0-9]+, 'avg_us': ([0-9.]+).*/batched \1 one-at-a-time \2 bm25_us \3/"
done
