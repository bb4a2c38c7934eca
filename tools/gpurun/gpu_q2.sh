#!/bin/bash
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/qprof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/qprof -- python3 $R/tools/query_bench.py 2>&1 | grep -v "^[WE]2026" | tail -1
cd $R; f=$(find gpurun_out/qprof -name "*kernel_stats.csv" | head -1); python - <<PY
import csv
for r in csv.reader(open("$f")):
    print(r[0][:50].ljust(50), r[1:7])
PY
