#!/bin/bash
# round 4: split-K tail of the residual GEMMs -- unit tests, the forwards' tests, then A/B through the ViT bench and the EVA02 bench
# (HIPTS_GEMM_SPLITK=0 off, default 4; HIPTS_GEMM_SPLITK_MINKT K-tiles per slice at least), one process per setting
mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
export HIPTS_BENCH_NO_SUSTAINED=1
for cfgs in "0 5" "4 5" "4 3" "2 5" "0 5" "4 5"; do
  set -- $cfgs
  HIPTS_GEMM_SPLITK=$1 HIPTS_GEMM_SPLITK_MINKT=$2 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-query --no-exclusive > gpurun_out/r04/sk_vit.json 2> gpurun_out/sk.err || { tail -5 gpurun_out/sk.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04/sk_vit.json').read().strip().splitlines()[-1]); print('ViT splitk $1 minkt $2: images/s', round(d['value'],1))"
done
for cfgs in "0 5" "4 5" "3 4" "0 5" "4 5"; do
  set -- $cfgs
  echo "== EVA splitk $1 minkt $2"
  HIPTS_GEMM_SPLITK=$1 HIPTS_GEMM_SPLITK_MINKT=$2 timeout -k 10 300 python tools/eva_bench.py 2>&1 | tail -4 | cut -c1-300
done
