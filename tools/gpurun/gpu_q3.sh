#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_query.py -m gpu -x -q 2>&1 | tail -2 || exit 1
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/qprof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/qprof -- python3 $R/tools/query_bench.py 2>&1 | grep -v "^[WE]2026" | tail -1
cd $R; f=$(find gpurun_out/qprof -name "*kernel_stats.csv" | head -1); python - <<PY
import csv
import collections
for r in list(csv.reader(open("$f")))[:7]:
    print(r[0][:50].ljust(50), r[1:5])
PY
python - <<'PY'
import csv,glob,collections
f=sorted(glob.glob('gpurun_out/qprof/runc/*_kernel_trace.csv'))[-1]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    for k in ('topk_kernel','sim_mfma','bm25_postings','rowmax','combine'):
        if k in r['Kernel_Name']: agg[(k,int(r['Grid_Size_X']))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items()): print(k, len(v), round(sum(v)/len(v),1))
PY
