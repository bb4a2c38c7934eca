#!/bin/bash
# rocprofv3 kernel stats of the bench command, then the two PMC passes (separate runs, no tracing domains mixed in)
./tools/gpurun/gpu_run.sh prof > gpurun_out/profall_prof.log 2>&1 || { tail -20 gpurun_out/profall_prof.log; exit 1; }
tail -30 gpurun_out/profall_prof.log | cut -c1-200
./tools/gpurun/gpu_pmc.sh 2>&1 | tail -15
