#!/bin/bash
# FETCH_SIZE of the bench command with the column-group raster (HIPTS_GEMM_RASTER_GN) against the default
mkdir -p gpurun_out/r03
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
R=$GRAFT_REPO_ROOT
export HIPTS_BENCH_NO_SUSTAINED=1
cd /tmp && export TMPDIR=/tmp
for gn in "$@"; do
  export HIPTS_GEMM_RASTER_GN=$gn
  rm -rf $R/gpurun_out/pmc_gn$gn
  for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gn$gn/$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-query --no-exclusive > $R/gpurun_out/pmc_gn.json 2> $R/gpurun_out/pmc_gn.err || { tail -5 $R/gpurun_out/pmc_gn.err; exit 1; }
  done
  echo "raster_gn $gn"
  python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc_gn$gn $R/gpurun_out/r03/raster_gn${gn}_pmc_traffic.json | head -6
done
