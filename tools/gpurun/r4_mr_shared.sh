#!/bin/bash
# round 4: launches smaller than the chip -- two-per-CU kernel up to DW_LIMIT/4 of the CUs in 256 x 256 tiles, above that the persistent kernel
# with the tile height by cost; CCIP / EVA02 / ViT numbers per setting (default: limit 2, by cost when the launch is smaller than the chip)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
export HIPTS_BENCH_NO_SUSTAINED=1
for cfg in "2 -1" "3 0" "2 -1" "3 0"; do set -- $cfg; echo "HIPTS_GEMM_DW_LIMIT=$1 HIPTS_GEMM_MR_SHARED=$2"
  HIPTS_GEMM_DW_LIMIT=$1 HIPTS_GEMM_MR_SHARED=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/epi.err | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline()); print('   images/s %.0f  frac %.3f  ccip %.0f / %.0f  eva %.0f / %.0f' % (d['value'], d['model_mfma_frac'], d['ccip']['images_per_s_batch20'], d['ccip']['images_per_s_batch64'], d['eva02_large']['images_per_s_batch10'], d['eva02_large']['images_per_s_batch32']))" || exit 1
done
