#!/bin/bash
# A/B of the column-group tile raster (HIPTS_GEMM_RASTER_GN): ViT forward images/s and the fc1 shape alone
mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
OUT=gpurun_out/r03/raster.txt
rm -f $OUT
for rep in 1 2; do
for gn in 0 6 4 3 5; do
  echo "raster_gn $gn" | tee -a $OUT
  HIPTS_GEMM_RASTER_GN=$gn timeout -k 10 200 python tools/vit_batch_sweep.py 64 2>&1 | grep -v Warning | grep batch | tee -a $OUT || exit 1
done
done
for gn in 0 6 4 3; do
  echo "raster_gn $gn (alone)" | tee -a $OUT
  HIPTS_GEMM_RASTER_GN=$gn timeout -k 10 200 python tools/gemm_bench.py gelu,25088,3072,768 gelu,50176,3072,768 2>&1 | grep -v Warning | tee -a $OUT || exit 1
done
