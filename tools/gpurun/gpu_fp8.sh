#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -m gpu -x -q -k e4m3 2>&1 | tail -15 || exit 1
SH="star,36864,2048,512 resid,36864,512,2048 star,147456,1024,256 resid,147456,256,1024 star,589824,512,128 resid,589824,128,512 resid,50176,768,3072 star,16384,4096,4096"
echo "--- 16-bit"; timeout -k 10 300 python tools/gemm_bench.py $SH 2>&1 | grep -v amdgpu.ids
echo "--- e4m3"; HIPTS_GEMM_OP8=1 timeout -k 10 300 python tools/gemm_bench.py $SH 2>&1 | grep -v amdgpu.ids
echo "--- e4m3 out8"; HIPTS_GEMM_OP8=1 HIPTS_GEMM_OUT8=1 timeout -k 10 300 python tools/gemm_bench.py star,36864,2048,512 star,147456,1024,256 star,589824,512,128 star,16384,4096,4096 2>&1 | grep -v amdgpu.ids
