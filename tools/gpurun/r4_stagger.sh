#!/bin/bash
# round 4: sub-batch streams out of phase (HIPTS_VIT_STAGGER = position in stream 0's first layer at which the later streams start):
# images/s of the bench per value, two runs each, one box
mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
export HIPTS_BENCH_NO_SUSTAINED=1
for rep in 1 2; do
for st in 0 1 2 3 4 5; do
  HIPTS_VIT_STAGGER=$st timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-query --no-exclusive > gpurun_out/r04/stagger_$st.json 2> gpurun_out/stagger.err || { tail -5 gpurun_out/stagger.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04/stagger_$st.json').read().strip().splitlines()[-1]); print('stagger $st rep $rep: images/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3))"
done
done
