#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 300 python tools/single_query_bench.py 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_q && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_q -- python3 $GRAFT_REPO_ROOT/tools/single_query_bench.py > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof_q.err; cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_q -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
