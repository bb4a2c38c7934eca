#!/bin/bash
# round 4: read the message of an abort inside the ViT forward (pytest's capture swallows it): the failing test without capture, with the
# pool kernel's split count at its old and its new value
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for ps in 8 28; do
  echo "== HIPTS_POOL_SPLITS=$ps"
  HIPTS_POOL_SPLITS=$ps timeout -k 10 300 python -m pytest "tests/test_gpu_vit.py::test_vit_b16_448_matches_oracle" -m gpu -q -x -s 2>&1 | grep -v "^  File" | tail -12 | cut -c1-300
done
