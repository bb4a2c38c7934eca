#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
rm -f gpurun_out/raster.txt
for rep in 1 2; do
for r in 0 8 4 16; do
  echo "raster $r" | tee -a gpurun_out/raster.txt
  HIPTS_GEMM_RASTER=$r timeout -k 10 200 python tools/vit_batch_sweep.py 64 2>&1 | grep -v Warning | grep batch | tee -a gpurun_out/raster.txt || exit 1
done
done
HIPTS_GEMM_RASTER=8 timeout -k 10 600 python -m pytest tests/test_gpu_vit.py -x -q -m gpu 2>&1 | tail -3
