#!/bin/bash
# first GPU contact: build check + parity tests, output under gpurun_out/
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu.log | tail -40
