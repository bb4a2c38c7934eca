#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
rm -f gpurun_out/trim.txt
timeout -k 10 500 python -m pytest tests/test_gpu_attention.py tests/test_gpu_vit.py tests/test_gpu_eva.py tests/test_gpu_ccip.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for rep in 1 2 3; do
for r in 0 1; do
  echo "trim $r" | tee -a gpurun_out/trim.txt
  HIPTS_ATTN_TRIM=$r timeout -k 10 200 python tools/vit_batch_sweep.py 64 2>&1 | grep -v Warning | grep batch | tee -a gpurun_out/trim.txt || exit 1
done
done
