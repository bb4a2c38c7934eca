#!/bin/bash
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_vit.py tests/test_gpu_gemm.py::test_forward_is_deterministic_and_batch_invariant -m gpu -x -q 2>&1 | tail -3 || exit 1
for rep in 1 2 3; do for m in 1 0; do HIPTS_QKV_MERGE=$m timeout -k 10 300 python bench.py --no-cpu-baseline --no-query --no-exclusive 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('merge=$m', d['value'], d['ms_per_step'])"; done; done
