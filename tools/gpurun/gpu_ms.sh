#!/bin/bash
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for rep in 1 2; do for m in 2 3 4; do HIPTS_VIT_STREAMS=$m timeout -k 10 300 python bench.py --no-cpu-baseline --no-query --no-exclusive 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams=$m', d['value'], d['ms_per_step'])"; done; done
