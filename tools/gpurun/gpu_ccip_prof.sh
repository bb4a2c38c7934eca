#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_ccip.py -m gpu -x -q 2>&1 | tail -2 || exit 1
timeout -k 10 300 python tools/ccip_bench.py 0 2 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ccip && HIPTS_CCIP_STREAMS=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ccip -- python3 $GRAFT_REPO_ROOT/tools/ccip_bench.py 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_ccip.out 2> $GRAFT_REPO_ROOT/gpurun_out/prof_ccip.err; cd $GRAFT_REPO_ROOT; cat gpurun_out/prof_ccip.out; f=$(find gpurun_out/prof_ccip -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-170
