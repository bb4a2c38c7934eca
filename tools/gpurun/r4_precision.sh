#!/bin/bash
# round 4: the operand-precision cost curve -- ViT / attention / benched-query tests, tools/precision_curve.py (ViT-B/16 and EVA02-L:
# logit error per image kind and images/s per operand mode, one process), GEMM timings of the doubled-K shapes the other splits would run
mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_vit.py tests/test_gpu_eva.py "tests/test_gpu_configs.py::test_config2_benched_shape_256_queries_fused_topk" -m gpu -q -rf -s > gpurun_out/r4_parity.log 2>&1
echo "parity pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r4_parity.log | cut -c1-200 | tail -20
timeout -k 10 600 python tools/precision_curve.py > gpurun_out/r04/r04_precision_vit.json 2> gpurun_out/r04/r04_precision_vit.log; echo "vit rc=$?"; cat gpurun_out/r04/r04_precision_vit.log | cut -c1-250
timeout -k 10 600 python tools/precision_curve.py --eva --modes 1,17 > gpurun_out/r04/r04_precision_eva.json 2> gpurun_out/r04/r04_precision_eva.log; echo "eva rc=$?"; cat gpurun_out/r04/r04_precision_eva.log | cut -c1-250
timeout -k 10 300 python tools/gemm_bench.py qk,25088,2304,768 qk,25088,2304,1536 gelu,25088,3072,768 gelu,25088,3072,1536 resid,25088,768,768 resid,25088,768,1536 resid,25088,768,3072 resid,25088,768,6144 gelu,8192,8192,8192 2>&1 | tee gpurun_out/r04/r04_gemm_doubled_k.txt
