#!/bin/bash
# round 5: the one-query path with the BM25 chain dealt between the rounds of the index stream (HIPTS_S1_PIPE, default 1): tests, then A B A B
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_query.py -x -q -k "one_query or search_one or config2 or single" > gpurun_out/r5_s1_tests.txt 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r5_s1_tests.txt
tail -3 gpurun_out/r5_s1_tests.txt
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -k config2 >> gpurun_out/r5_s1_tests.txt 2>&1; echo "config2 rc=$?" | tee -a gpurun_out/r5_s1_tests.txt
for v in 0 1 0 1; do
  echo "== HIPTS_S1_PIPE=$v"; HIPTS_S1_PIPE=$v timeout -k 10 300 python tools/single_query_bench.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r5_s1_ab.txt 2>&1
grep -n "HIPTS_S1_PIPE\|single query\|search1_score\|finish\|topk_kernel\|C ABI" gpurun_out/r5_s1_ab.txt
