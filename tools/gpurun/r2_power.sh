#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 200 python tools/vit_power.py 6 2>&1 | grep -v Warning > gpurun_out/power.txt; echo rc=$?
(rocm-smi --showpower --showmaxpower --showclocks 2>&1 | head -40) >> gpurun_out/power.txt
tail -60 gpurun_out/power.txt
