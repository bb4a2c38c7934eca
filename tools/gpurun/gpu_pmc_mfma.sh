#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_mfma
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-query --no-exclusive > $R/gpurun_out/pmc_mfma.json 2> $R/gpurun_out/pmc_mfma.err || { tail -5 $R/gpurun_out/pmc_mfma.err; exit 1; }
cd $R && head -2 $(find gpurun_out/pmc_mfma -name "*counter_collection.csv" | head -1) | cut -c1-400; python tools/pmc_mfma.py gpurun_out/pmc_mfma gpurun_out/pmc_mfma_util.json
