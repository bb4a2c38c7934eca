#!/bin/bash
# round 4: what bounds the hybrid-decode pipeline -- worker count sweep (rows 3 = Pillow workers, 4 = entropy-decode workers)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
mkdir -p gpurun_out/r04
for w in 16 14 12 20; do
  echo "== $w workers"; E2E_MODES=3,4 timeout -k 10 300 python tools/pipeline_e2e.py 8192 $w 2>&1 | grep -v amdgpu.ids | grep "processes"
done
