#!/bin/bash
# round 4: how much of a CCIP forward's wall time is inside kernels (launch gaps), batch 20 (one stream) and 64 (two streams)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for b in 20 64; do
  rm -rf $R/gpurun_out/gaps_$b
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps_$b -- python3 $R/tools/ccip_gaps.py $b > $R/gpurun_out/gaps_$b.log 2>&1 || { tail -5 $R/gpurun_out/gaps_$b.log; exit 1; }
  grep "per forward" $R/gpurun_out/gaps_$b.log
  python3 - $R/gpurun_out/gaps_$b <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Stream_Id", "")))
rows.sort()
# the last 10 forwards: from the last 10 stem kernels on
stems = [i for i, r in enumerate(rows) if "stem_im2col" in r[2]]
per = len(stems) // 13 if len(stems) >= 13 else 1          # kernels named stem per forward (1 or 2 streams)
first = stems[-10 * per]
sel = rows[first:]
wall = sel[-1][1] - sel[0][0]
busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]
for s, e, _, _ in sel[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("   last 10 forwards: %d kernels, span %.2f ms, some kernel running %.2f ms (%.1f %%), sum of kernel times %.2f ms" % (len(sel), wall / 1e6, busy / 1e6, 100.0 * busy / wall, sum(e - s for s, e, _, _ in sel) / 1e6))
PY
done
