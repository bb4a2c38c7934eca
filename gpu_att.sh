#!/bin/bash
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_vit.py tests/test_gpu_gemm.py::test_forward_is_deterministic_and_batch_invariant -m gpu -x -q 2>&1 | tail -4 || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query 2>&1 >/dev/null | grep -E "attn|EPI_GELU"
