#!/bin/bash
# round 4: A/B of the predicate-free store path of the staged 16-bit GEMM epilogues (GELU / StarReLU / residual + LayerNorm): gemm.o built
# with -DHIPTS_STAGED_INTERIOR=0, benched, then the default object again, benched, then once more each (A B A B); the default object stays
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_vit.py tests/test_gpu_eva.py tests/test_gpu_ccip.py -m gpu -q -x 2>&1 | tail -2 || exit 1
cd anime-illust-image-searcher_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function"
cp gemm.o /tmp/gemm_default.o
/opt/rocm/bin/hipcc $FLAGS -DHIPTS_STAGED_INTERIOR=0 -c gemm.hip -o /tmp/gemm_pred.o || exit 1
echo "variant built"
cd ../..
export HIPTS_BENCH_NO_SUSTAINED=1
bench() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/epi.err | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline()); print('   images/s %.0f  frac %.3f  ccip %.0f / %.0f  eva %.0f / %.0f' % (d['value'], d['model_mfma_frac'], d['ccip']['images_per_s_batch20'], d['ccip']['images_per_s_batch64'], d['eva02_large']['images_per_s_batch10'], d['eva02_large']['images_per_s_batch32']))"; }
link() { cp $1 anime-illust-image-searcher_amd/csrc/gemm.o && (cd anime-illust-image-searcher_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhip_tagsearch.so *.o); }
for v in pred default pred default; do echo "gemm.o = $v"; link /tmp/gemm_$v.o || exit 1; bench || exit 1; done
