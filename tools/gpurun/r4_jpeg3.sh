#!/bin/bash
# round 4: device time of the hybrid decode per batch (tools/jpeg_bench.py) and its kernels (rocprofv3 kernel trace)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
mkdir -p gpurun_out/r04
timeout -k 10 200 python tools/jpeg_bench.py 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/jpeg_prof -o jpeg -- python3 $GRAFT_REPO_ROOT/tools/jpeg_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r04/jpeg_prof.out 2>&1
cd $GRAFT_REPO_ROOT && f=$(find gpurun_out/r04/jpeg_prof -name "*kernel_stats.csv" | head -1) && cp $f gpurun_out/r04/jpeg_kernel_stats.csv && head -12 $f | cut -c1-220
