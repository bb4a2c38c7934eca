#!/bin/bash
# round 5: in-kernel stamps of the 4-wave loop (csrc/gemm4.hip), the ViT's shapes per 32-image sub-batch, half operands; then bits + times
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export HIPTS_DBG_GEMM_F16=1 HIPTS_DBG_GEMM_SHARED=1
SH="gelu,25088,3072,768 xg,25088,768,768 xg,25088,768,3072"
( echo "== 4-wave, stamps"; HIPTS_GEMM_Q4=8210 HIPTS_GEMM_STAMPS=1 timeout -k 10 200 python tools/gemm_bench.py $SH 2>&1 | grep -v "^wave [0-9] tile [0-9]: *-\?[0-9]* *[0-9]* *[0-9]* *[0-9]* -" | grep -v "stamps of workgroup" 
  echo "== 4-wave, no stamps"; HIPTS_GEMM_Q4=8210 timeout -k 10 200 python tools/gemm_bench.py $SH qk,25088,2304,768 gelu,8192,8192,8192 gelu,4096,4096,4096
  echo "== 8-wave, no stamps"; timeout -k 10 200 python tools/gemm_bench.py $SH qk,25088,2304,768 gelu,8192,8192,8192 gelu,4096,4096,4096 ) > gpurun_out/r5_q4_stamps.txt 2>&1
tail -70 gpurun_out/r5_q4_stamps.txt
timeout -k 10 300 python tools/gemm_q4_check.py > gpurun_out/r5_q4_check.txt 2>&1; echo "check rc=$?" >> gpurun_out/r5_q4_check.txt
grep -v "did not take" gpurun_out/r5_q4_check.txt | tail -20
