#!/bin/bash
# round 4: the hi | lo split of the attention output -- its test, the scale of the low halves (HIPTS_SPLIT_LO_SCALE: one process per value,
# the scale is read once) through tools/precision_curve.py, and the PV-DM inference tests
mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_d2v_tags.py "tests/test_gpu_vit.py::test_vit_split_attention_output" -m gpu -q -rf -s > gpurun_out/r4_split_tests.log 2>&1
echo "pytest rc=$?"; grep -E "max \|dlogit\||passed|failed|Error|error" gpurun_out/r4_split_tests.log | cut -c1-200 | tail -20
for sc in 1 16 64 256; do
  HIPTS_SPLIT_LO_SCALE=$sc timeout -k 10 300 python tools/precision_curve.py --modes 1,17 --rounds 3 > gpurun_out/r04/r04_precision_vit_scale$sc.json 2> gpurun_out/r04/r04_precision_vit_scale$sc.log || exit 1
  echo "== lo scale $sc"; grep "^mode" gpurun_out/r04/r04_precision_vit_scale$sc.log | cut -c1-220
done
timeout -k 10 400 python tools/precision_curve.py --eva --modes 1,17 --rounds 3 > gpurun_out/r04/r04_precision_eva.json 2> gpurun_out/r04/r04_precision_eva.log; echo "eva rc=$?"; grep "^mode" gpurun_out/r04/r04_precision_eva.log | cut -c1-250
