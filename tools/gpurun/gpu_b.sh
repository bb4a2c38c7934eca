#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err; tail -3 gpurun_out/bench_q.err
python -c "import json; d=json.loads(open('gpurun_out/bench_q.json').read()); print(d['value'], d['roofline'].get('clock'))"
