#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
run() { timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query --no-exclusive 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 img/s', round(d['value'],1))"; }
run s2; HIPTS_VIT_STREAMS=3 run s3; HIPTS_VIT_STREAMS=4 run s4; HIPTS_VIT_STREAMS=1 run s1; run s2
