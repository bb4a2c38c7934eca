#!/bin/bash
# round 4: gemm_dw2_kernel (two tiles in flight inside one workgroup, HIPTS_GEMM_DW2_MASK: 16 = fc1 + GELU, 2 = q | k | v) -- one ViT-B/16
# parity test under a short timeout first (a barrier mismatch would hang), then timing against the persistent 256 x 256 loop, then the
# ViT tests and the bench
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
HIPTS_GEMM_DW2_MASK=18 timeout -k 10 150 python -m pytest "tests/test_gpu_vit.py::test_vit_b16_448_matches_oracle" -m gpu -q -x -s 2>&1 | grep -v "^  File" | tail -6 | cut -c1-250
rc=${PIPESTATUS[0]}; echo "first test rc=$rc"; [ $rc -ne 0 ] && exit 1
echo "== pp"; timeout -k 10 120 python tools/gemm_bench.py gelu,25088,3072,768 qk,25088,2304,768 gelu,50176,3072,768 qk,50176,2304,768 2>&1 | grep -v amdgpu.ids
echo "== dw2"; HIPTS_GEMM_DW2_MASK=18 timeout -k 10 120 python tools/gemm_bench.py gelu,25088,3072,768 qk,25088,2304,768 gelu,50176,3072,768 qk,50176,2304,768 2>&1 | grep -v amdgpu.ids
HIPTS_GEMM_DW2_MASK=18 timeout -k 10 600 python -m pytest tests/test_gpu_vit.py -m gpu -q -rf -x > gpurun_out/r4_dw2_vit.log 2>&1; rc=$?
echo "vit pytest (dw2 for fc1 + qkv) rc=$rc"; tail -5 gpurun_out/r4_dw2_vit.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
export HIPTS_BENCH_NO_SUSTAINED=1
for m in 0 18 16 2 0 18; do
  HIPTS_GEMM_DW2_MASK=$m timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-query --no-exclusive > gpurun_out/r04/dw2.json 2> gpurun_out/dw2.err || { tail -5 gpurun_out/dw2.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04/dw2.json').read().strip().splitlines()[-1]); print('ViT dw2 mask $m: images/s', round(d['value'],1))"
done
