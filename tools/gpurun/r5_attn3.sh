#!/bin/bash
# round 5: the one-wave-per-SIMD attention (attn3.h, variant 6) against float64 and timed next to the default (variant 5)
R=$GRAFT_REPO_ROOT
timeout -k 10 500 python3 $R/tools/attn2_check.py both 6,5 > $R/gpurun_out/r5_attn3.txt 2>&1
echo "rc $?" >> $R/gpurun_out/r5_attn3.txt
tail -60 $R/gpurun_out/r5_attn3.txt
