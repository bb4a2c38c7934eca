#!/bin/bash
# round 5: LDS bank conflicts and LDS activity per kernel of the forward (one --pmc pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export HIPTS_BENCH_NO_SUSTAINED=1
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmc_lds$i
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_lds$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-query --no-exclusive > $R/gpurun_out/pmc_lds$i.json 2> $R/gpurun_out/pmc_lds$i.err || { tail -5 $R/gpurun_out/pmc_lds$i.err; }
  echo "=== $set"; python3 $R/tools/pmc_generic.py $R/gpurun_out/pmc_lds$i 5
done > $R/gpurun_out/r5_pmc_lds.txt 2>&1
tail -90 $R/gpurun_out/r5_pmc_lds.txt
