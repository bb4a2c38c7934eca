#!/bin/bash
# round 4: the persistent two-per-CU GEMM (gemm_dwp_kernel, HIPTS_GEMM_DWP_MASK) -- correctness through the GEMM test harness (EPI_RESID = bit 3),
# then timing of the fc1 / q|k|v shapes per start delay, then the ViT tests and bench with fc1 + q|k|v on it
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
HIPTS_GEMM_DWP_MASK=8 timeout -k 10 600 python -m pytest "tests/test_gpu_gemm.py::test_gemm_default_dispatch_persistent_and_underfilled" "tests/test_gpu_gemm.py::test_gemm_half_operands_and_tile_heights" -m gpu -q -rf -x > gpurun_out/r4_dwp_gemm.log 2>&1; rc=$?
echo "gemm pytest (dwp for EPI_RESID) rc=$rc"; tail -5 gpurun_out/r4_dwp_gemm.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
echo "== pp"; timeout -k 10 200 python tools/gemm_bench.py gelu,25088,3072,768 qk,25088,2304,768 gelu,50176,3072,768 2>&1 | grep -v amdgpu.ids
for sl in 0 1 2 3 4; do
  echo "== dwp sleep $sl"
  HIPTS_GEMM_DWP_MASK=18 HIPTS_GEMM_DWP_SLEEP=$sl timeout -k 10 200 python tools/gemm_bench.py gelu,25088,3072,768 qk,25088,2304,768 gelu,50176,3072,768 2>&1 | grep -v amdgpu.ids
done
HIPTS_GEMM_DWP_MASK=18 timeout -k 10 600 python -m pytest tests/test_gpu_vit.py -m gpu -q -rf -x > gpurun_out/r4_dwp_vit.log 2>&1; rc=$?
echo "vit pytest (dwp for fc1 + qkv) rc=$rc"; tail -5 gpurun_out/r4_dwp_vit.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
export HIPTS_BENCH_NO_SUSTAINED=1
for cfgs in "0 2" "18 2" "18 1" "18 3" "0 2" "18 2"; do
  set -- $cfgs
  HIPTS_GEMM_DWP_MASK=$1 HIPTS_GEMM_DWP_SLEEP=$2 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-query --no-exclusive > gpurun_out/r04/dwp.json 2> gpurun_out/dwp.err || { tail -5 gpurun_out/dwp.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04/dwp.json').read().strip().splitlines()[-1]); print('ViT dwp mask $1 sleep $2: images/s', round(d['value'],1))"
done
