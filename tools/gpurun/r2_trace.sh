#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_q1 && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_q1 -- python3 $GRAFT_REPO_ROOT/tools/s1_sweep.py > $GRAFT_REPO_ROOT/gpurun_out/prof_q1.out 2>&1
cd $GRAFT_REPO_ROOT; tail -2 gpurun_out/prof_q1.out
f=$(find gpurun_out/prof_q1 -name "*kernel_trace.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
mid=len(rows)//3
prev_end=None
for r in rows[mid:mid+13]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap = (s-prev_end)/1000 if prev_end else 0
    print("%-34s dur %7.1f us  gap %7.1f us" % (r['Kernel_Name'].replace('(anonymous namespace)::','')[:34], (e-s)/1000, gap))
    prev_end=e
PY
