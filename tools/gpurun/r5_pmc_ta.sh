#!/bin/bash
# round 5: address-path counters per kernel of the forward (two separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export HIPTS_BENCH_NO_SUSTAINED=1
i=0
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmc_ta$i
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ta$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-query --no-exclusive > $R/gpurun_out/pmc_ta$i.json 2> $R/gpurun_out/pmc_ta$i.err || { tail -5 $R/gpurun_out/pmc_ta$i.err; }
  echo "=== $set"; python3 $R/tools/pmc_generic.py $R/gpurun_out/pmc_ta$i 6
done > $R/gpurun_out/r5_pmc_ta.txt 2>&1
tail -120 $R/gpurun_out/r5_pmc_ta.txt
