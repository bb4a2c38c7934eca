#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_eva.py -m gpu -x -q -s 2>&1 | grep -E "EVA|passed|failed|Error" || exit 1
timeout -k 10 600 python tools/eva_bench.py 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_eva && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_eva -- python3 $GRAFT_REPO_ROOT/tools/eva_bench.py > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof_eva.err; cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_eva -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-150
