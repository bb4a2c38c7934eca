#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function"
run() { timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query --no-exclusive 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 img/s', round(d['value'],1))"; }
run default; run default; timeout -k 10 300 python tools/eva_bench.py 2>&1 | tail -1
cd anime-illust-image-searcher_amd/csrc && touch gemm.hip && make CXXFLAGS="$F -DHIPTS_STAGE_W3=1" > /dev/null 2>&1; cd ../..
run w3; run w3; timeout -k 10 300 python tools/eva_bench.py 2>&1 | tail -1
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -m gpu -x -q 2>&1 | tail -1
cd anime-illust-image-searcher_amd/csrc && touch gemm.hip && make CXXFLAGS="$F" > /dev/null 2>&1; cd ../..
run default; run default
