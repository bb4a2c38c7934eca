#!/bin/bash
# round 4: build, the attention / ViT parity tests with their printed error tables, the whole GPU suite, smoke(), the driver-shaped bench
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_vit.py -m gpu -q -rf -s --durations=8 > gpurun_out/r4_parity.log 2>&1
echo "parity pytest rc=$?"; grep -E "operands|trained|labels selected|window-edge|forced-fallback|passed|failed|Error|error" gpurun_out/r4_parity.log | cut -c1-200 | tail -80
timeout -k 10 900 python -m pytest tests -m gpu -q -rf --durations=8 --deselect tests/test_gpu_attention.py --deselect tests/test_gpu_vit.py > gpurun_out/pytest_gpu.log 2>&1; rc=$?
echo "suite rc=$rc"; tail -30 gpurun_out/pytest_gpu.log | cut -c1-220
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
[ "$1" = "nobench" ] && exit 0
timeout -k 10 700 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -30 gpurun_out/bench.err | cut -c1-220; cut -c1-1200 gpurun_out/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
print(json.dumps(d.get("output_check"), indent=1)[:3000])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["model_mfma_frac"], "sustained", d.get("sustained", {}).get("images_per_s"))
PY
