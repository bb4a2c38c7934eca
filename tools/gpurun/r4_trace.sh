#!/bin/bash
# round 4: per-kernel times of the ViT forward at batch 64 with ONE stream and with the default two sub-batch streams (rocprofv3 kernel stats of
# the same bench command), to see what each launch costs with the chip to itself inside a sustained forward
mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
R=$GRAFT_REPO_ROOT
export HIPTS_BENCH_NO_SUSTAINED=1
cd /tmp && export TMPDIR=/tmp
for ns in 1 2; do
  export HIPTS_VIT_STREAMS=$ns
  rm -rf $R/gpurun_out/prof_s$ns
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s$ns -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query --no-exclusive > $R/gpurun_out/r04/r04_trace_s$ns.out 2> $R/gpurun_out/prof_s$ns.err || { tail -5 $R/gpurun_out/prof_s$ns.err; exit 1; }
  f=$(find $R/gpurun_out/prof_s$ns -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $R/gpurun_out/r04/r04_trace_s${ns}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    for r in rows:
        w.writerow([r[0][:160]] + r[1:])
print("".join(",".join(r[:5])[:170] + "\n" for r in rows[:12]))
PY
  python3 -c "
import json; d=json.loads(open('$R/gpurun_out/r04/r04_trace_s$ns.out').read().strip().splitlines()[-1]); print('streams $ns (under rocprof): images/s', round(d['value'],1), 'ms', d['ms_per_step'])"
done
unset HIPTS_VIT_STREAMS
cd $R
for ns in 1 2; do
  HIPTS_VIT_STREAMS=$ns timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-query --no-exclusive > gpurun_out/r04/r04_bench_s$ns.json 2> gpurun_out/bench_s$ns.err
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04/r04_bench_s$ns.json').read().strip().splitlines()[-1]); print('streams $ns: images/s', round(d['value'],1), 'ms', d['ms_per_step'])"
done
