#!/bin/bash
# round 4: the residual epilogue's interior instantiation (gemm_pp_kernel<EPI_RESID_XG, 8, ., ., ., INT = true>) and the rational erf GELU.
# HIPTS_RESID_GENERAL=1 runs the predicated epilogue everywhere (A/B).  Parity first, then launch times, then the forward.
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_vit.py tests/test_gpu_eva.py -m gpu -q -rf -x > gpurun_out/r4_resid_tests.log 2>&1; rc=$?
echo "gemm + vit + eva pytest rc=$rc"; tail -5 gpurun_out/r4_resid_tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
echo "== interior epilogue"; timeout -k 10 120 python tools/gemm_bench.py xg,25088,768,768 xg,25088,768,3072 resid,25088,768,768 resid,25088,768,3072 2>&1 | grep -v amdgpu.ids
echo "== general epilogue"; HIPTS_RESID_GENERAL=1 timeout -k 10 120 python tools/gemm_bench.py xg,25088,768,768 xg,25088,768,3072 2>&1 | grep -v amdgpu.ids
export HIPTS_BENCH_NO_SUSTAINED=1
mkdir -p gpurun_out/r04
for g in 1 0 1 0; do
  HIPTS_RESID_GENERAL=$g timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-query --no-exclusive > gpurun_out/r04/resid.json 2> gpurun_out/resid.err || { tail -5 gpurun_out/resid.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04/resid.json').read().strip().splitlines()[-1]); print('ViT HIPTS_RESID_GENERAL=$g: images/s', round(d['value'],1), 'max logit err', d.get('output_check',{}).get('oracle',{}).get('max_abs_logit_error'))"
done
