#!/bin/bash
# EVA02-L at the reference's batch of 10: one stream + 192-row tiles against two sub-batch streams
mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
OUT=gpurun_out/r03/eva_mr.txt
rm -f $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_eva.py tests/test_gpu_vit.py -x -q -m gpu 2>&1 | tail -3 | tee -a $OUT || exit 1
for rep in 1 2; do
for cfg in "5 192" "6 192" "6 224" "6 256" "5 224"; do
  set -- $cfg
  echo "minsub $1 bm $2" | tee -a $OUT
  HIPTS_EVA_MINSUB=$1 HIPTS_GEMM_BM=$2 timeout -k 10 300 python tools/eva_bench.py 2>&1 | grep batch | tee -a $OUT || exit 1
done
done
