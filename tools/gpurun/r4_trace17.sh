#!/bin/bash
# round 4: why does the hi | lo attention output cost 21 % with half operands and 6 % with bf16?  rocprofv3 kernel stats of
# tools/precision_curve.py per operand mode (one mode per process)
mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in 1 17 16; do
  rm -rf $R/gpurun_out/prof_m$m
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_m$m -- python3 $R/tools/precision_curve.py --modes $m --rounds 2 --steps 5 > $R/gpurun_out/r04/r04_mode$m.json 2> $R/gpurun_out/prof_m$m.err || { tail -5 $R/gpurun_out/prof_m$m.err; exit 1; }
  f=$(find $R/gpurun_out/prof_m$m -name "*kernel_stats.csv" | head -1)
  echo "== mode $m"; grep "images/s" $R/gpurun_out/prof_m$m.err
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
print("".join(",".join([r[0][:90]] + r[1:5]) + "\n" for r in rows[:10]))
PY
done
