#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
S=$(date +%s); python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "default bench wall: $(( $(date +%s) - S )) s"; python -c "
import json; d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['steps'], d['warmup'], list(d.keys()))"
