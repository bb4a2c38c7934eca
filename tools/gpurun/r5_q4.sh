#!/bin/bash
# round 5: the 4-wave GEMM loop (csrc/gemm4.hip) against the 8-wave loop: bits and launch times, then the forward either way
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python tools/gemm_q4_check.py > gpurun_out/r5_q4_check.txt 2>&1
echo "check rc=$?" >> gpurun_out/r5_q4_check.txt
tail -25 gpurun_out/r5_q4_check.txt
grep -q "check rc=0" gpurun_out/r5_q4_check.txt || exit 1
for m in 0 8210 0 8210; do
  HIPTS_GEMM_Q4=$m timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-query --no-cpu-baseline --no-exclusive 2> gpurun_out/r5_q4_bench_$m.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('Q4=$m', round(d['value'],1), 'img/s', [ (k['kernel'][:28], round(k['avg_us'],1)) for k in d['kernels'][:6]])" | tee -a gpurun_out/r5_q4_bench.txt
done
