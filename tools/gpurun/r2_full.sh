#!/bin/bash
# round 2: whole GPU test suite, smoke, the default bench command
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -22 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 || exit 1
timeout -k 10 700 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -40 gpurun_out/bench.err | cut -c1-220; cut -c1-1500 gpurun_out/bench.json
