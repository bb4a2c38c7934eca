#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_vit.py tests/test_gpu_gemm.py::test_forward_is_deterministic_and_batch_invariant -m gpu -x -q 2>&1 | tail -3 || exit 1
for st in 1 2; do HIPTS_VIT_STREAMS=$st timeout -k 10 300 python bench.py --no-cpu-baseline --no-query 2> gpurun_out/att.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams=$st', d['value'], d['ms_per_step'])"; grep -E "attn|layernorm" gpurun_out/att.err; done
