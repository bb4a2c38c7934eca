#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
SH="resid,50176,768,768 resid,50176,768,3072 gelu,50176,3072,768 qk,50176,1536,768 vt,50176,768,768 gelu,16384,4096,4096 gelu,8192,8192,8192"
echo "--- two tiles ahead"; timeout -k 10 300 python tools/gemm_bench.py $SH 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('img/s', d['value'], 'excl', d['roofline']['exclusive']['frac'])"
cd anime-illust-image-searcher_amd/csrc && touch gemm.hip && make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function -DHIPTS_STAGE_AHEAD=0" > /dev/null 2>&1; cd ../..
echo "--- one tile ahead (previous)"; timeout -k 10 300 python tools/gemm_bench.py $SH 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-query 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('img/s', d['value'], 'excl', d['roofline']['exclusive']['frac'])"
