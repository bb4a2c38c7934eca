#!/bin/bash
# round 3: A/B of environment switches through the quick bench.  usage: r3_ab.sh "VAR=val VAR2=val" "..." (each argument one run)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for E in "$@"; do
  echo "== [$E]"
  env $E HIPTS_BENCH_SUSTAINED_STEPS=100 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-query --no-cpu-baseline --no-exclusive 2> gpurun_out/ab.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   value %.0f img/s  %.3f ms/step  sustained %.0f' % (d['value'], d['ms_per_step'], d['sustained']['images_per_s']))
for k in d['kernels']: print('      %-24s %4d  %7.1f us' % (k['kernel'], k['launches'], k['avg_us']))
" || tail -5 gpurun_out/ab.err
done
