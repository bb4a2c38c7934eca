#!/bin/bash
# round 4: row maxima of the index products from the product kernel's accumulators (HIPTS_SIM_PARTS=1, default) against rowmax_kernel
# over the stored products (=0): query tests under both, then the batched rate
mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for p in 1; do
  HIPTS_SIM_PARTS=$p timeout -k 10 900 python -m pytest tests/test_gpu_query.py tests/test_gpu_configs.py tests/test_gpu_flows.py -m gpu -q -rf -x > gpurun_out/r4_parts_tests$p.log 2>&1; rc=$?
  echo "HIPTS_SIM_PARTS=$p pytest rc=$rc"; tail -4 gpurun_out/r4_parts_tests$p.log | cut -c1-300
  [ $rc -ne 0 ] && exit 1
done
for p in 0 1 0 1; do
  echo "== HIPTS_SIM_PARTS=$p"
  HIPTS_SIM_PARTS=$p timeout -k 10 300 python tools/query_bench.py 2>&1 | tail -2 | cut -c1-1800
done
