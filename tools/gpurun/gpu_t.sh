#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_gemm.py -m gpu -x -q -s 2>&1 | grep -E "max|passed|failed|Error" | head
for i in 1 2; do timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query --no-exclusive 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('img/s', round(d['value'],1))"; done
