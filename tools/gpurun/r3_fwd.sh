#!/bin/bash
# round 3: forward-path tests (attention, GEMM, ViT, EVA, configs) + a quick bench line
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 800 python -m pytest tests/test_gpu_attention.py tests/test_gpu_gemm.py tests/test_gpu_vit.py tests/test_gpu_eva.py tests/test_gpu_configs.py -m gpu -q -rf -x --durations=5 > gpurun_out/r3_fwd.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r3_fwd.log | cut -c1-200
[ "$1" = "nobench" ] && exit 0
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-query --no-cpu-baseline > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err; echo "bench rc=$?"; grep -E "kernel|attn" gpurun_out/bench_q.err | cut -c1-160
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_q.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["model_mfma_frac"], "sustained", d.get("sustained", {}).get("images_per_s"))
for k in d["roofline"].get("exclusive", {}).get("kernels", []): print("  excl", k["kernel"], k["launches"], "%.1f us %.0f TF" % (k["avg_us"], k["tflops"]))
PY
